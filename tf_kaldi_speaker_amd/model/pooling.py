"""Pooling layers (reference model/pooling.py).  Only statistics_pooling is on the hot path;
self_attention / ghost_vlad raise NotImplementedError (SURVEY.md section 8f-4 / out of scope)."""
try:
    from .. import ops
    from .common import to_device
except (ImportError, ValueError):
    import ops
    from model.common import to_device

VAR2STD_EPSILON = 1e-12


def statistics_pooling(features, aux_features, endpoints, params, is_training):
    """[batch, length, dim] -> [batch, 2*dim] = concat(mean, stddev) with the variance floor of
    reference pooling.py:9-34 (wave-shuffle Welford kernel, csrc/xv_elementwise.hip)."""
    x = to_device(features)
    assert x.dim() == 3
    if x.shape[2] % 4 != 0:
        raise ValueError("statistics_pooling: the channel count must be a multiple of 4 (got %d)" % x.shape[2])
    return ops.stat_pool_forward(x)


def self_attention(features, aux_features, endpoints, params, is_training=None):
    raise NotImplementedError("Not implement self_attention pooling")


def ghost_vlad(features, aux_features, endpoints, params, is_training=None):
    raise NotImplementedError("Not implement ghost_vlad pooling")
