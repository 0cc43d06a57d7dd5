"""Layer helpers with the reference's model/common.py names, evaluated on the GPU through the
C-ABI (no TensorFlow graph: calls execute eagerly on torch CUDA buffers)."""
import numpy as np
import torch

try:
    from .. import ops
except (ImportError, ValueError):   # drop-in layout (PYTHONPATH=$TF_KALDI_ROOT)
    import ops


def shape_list(x):
    """Static shape as a list (reference common.py:7-24)."""
    return list(x.shape)


def to_device(x, dtype=torch.float32, device="cuda:0"):
    if isinstance(x, torch.Tensor):
        return x.to(device=device, dtype=dtype).contiguous()
    np_dtype = np.float32 if dtype == torch.float32 else np.int32
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np_dtype)).to(device)


def l2_scaling(x, scaling_factor, epsilon=1e-12, name="l2_norm"):
    """x * rsqrt(max(sum x^2, eps)) * scaling_factor along the last axis (reference common.py:45-58)."""
    if epsilon != 1e-12:
        raise NotImplementedError("l2_scaling: the HIP kernel fixes epsilon = 1e-12 (common.py:45)")
    x = to_device(x)
    flat = x.reshape(-1, x.shape[-1])
    return ops.l2_scaling_forward(flat, scaling_factor).reshape(x.shape)


_ALPHAS = {}      # "<name>/alpha" -> device tensor (the tf.variable_scope(name) of common.py:35-39)


def prelu(x, name="prelu", shared=False):
    """relu(x) + alpha * (x - |x|) / 2 with a trainable alpha per channel (last axis), initialised to 0.01; one scalar alpha when
    `shared` (reference common.py:27-42).  The variable lives in a module-level store keyed by "<name>/alpha"."""
    x = to_device(x)
    c = x.shape[-1]
    key = name + "/alpha"
    if key not in _ALPHAS:
        _ALPHAS[key] = torch.full((1 if shared else c,), 0.01, dtype=torch.float32, device=x.device)
    alpha = _ALPHAS[key]
    if alpha.numel() == 1 and c > 1:
        alpha = alpha.expand(c).contiguous()
    return ops.prelu_forward(x.reshape(-1, c), alpha).reshape(x.shape)
