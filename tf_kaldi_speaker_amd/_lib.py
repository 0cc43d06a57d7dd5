"""ctypes binding of the C-ABI declared in include/xvector_hip.h.

The shared library is the product; there is NO CPU fallback.  If it is missing
or an entry point fails, this module raises - it never routes to NumPy/torch.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# $XV_LIB: an alternative build of the same C-ABI (tools/variant.sh builds, tools/ab.sh / tools/probe.sh compare them on the GPU box); never a different backend
LIB_PATH = os.environ.get("XV_LIB") or os.path.join(_HERE, "libxvector_hip.so")

c_float_p = C.POINTER(C.c_float)
c_int32_p = C.POINTER(C.c_int32)


class XvError(RuntimeError):
    pass


# include/xvector_hip.h XV_ABI_VERSION: the layout of XvConfig below and the SIGNATURES table belong to this version
ABI_VERSION = 3


class XvConfig(C.Structure):
    """Mirror of `struct xv_config` (include/xvector_hip.h); struct_bytes is filled in on construction."""

    def __init__(self, *args, **kw):
        if args:      # struct_bytes is the first field: a positional XvConfig(feat_dim, ...) would shift every value by one, silently
            raise TypeError("XvConfig takes keyword arguments only (its first field is struct_bytes, filled in here)")
        super(XvConfig, self).__init__(**kw)
        self.struct_bytes = C.sizeof(XvConfig)

    _fields_ = [
        ("struct_bytes", C.c_int32),
        ("feat_dim", C.c_int32),
        ("num_speakers", C.c_int32),
        ("num_nodes_pooling_layer", C.c_int32),
        ("num_nodes_last_layer", C.c_int32),
        ("last_layer_no_bn", C.c_int32),
        ("last_layer_linear", C.c_int32),
        ("feature_norm", C.c_int32),
        ("feature_scaling_factor", C.c_float),
        ("loss_kind", C.c_int32),
        ("margin_m", C.c_float),
        ("lambda_min", C.c_float),
        ("lambda_base", C.c_float),
        ("lambda_gamma", C.c_float),
        ("lambda_power", C.c_float),
        ("weight_l2_regularizer", C.c_float),
        ("output_weight_l2_regularizer", C.c_float),
        ("batchnorm_momentum", C.c_float),
        ("bn_epsilon", C.c_float),
        ("fused_bn_unbiased_moving_var", C.c_int32),
        ("optimizer", C.c_int32),
        ("momentum", C.c_float),
        ("use_nesterov", C.c_int32),
        ("clip_gradient_norm", C.c_float),
        ("max_batch", C.c_int32),
        ("max_frames", C.c_int32),
        ("precision", C.c_int32),
        ("pooling", C.c_int32),
        ("att_key0_nodes", C.c_int32),
        ("att_key1_nodes", C.c_int32),
        ("att_key_type", C.c_int32),
        ("att_use_scale", C.c_int32),
        ("aux_ring", C.c_int32),
        ("ring_loss_init", C.c_float),
        ("ring_loss_lambda", C.c_float),
        ("aux_mhe", C.c_int32),
        ("mhe_lambda", C.c_float),
        ("num_frame_layers", C.c_int32),
        ("frame_context", C.c_int32 * 12),
        ("frame_width", C.c_int32 * 12),
        ("relu_type", C.c_int32),
        ("max_rows", C.c_int32),
    ]


LOSS_KINDS = {
    "softmax": 0,
    "asoftmax": 1,
    "additive_margin_softmax": 2,
    "additive_angular_margin_softmax": 3,
}
OPTIMIZERS = {"sgd": 0, "momentum": 1, "adam": 2}
PRECISIONS = {"f32": 0, "f16x3": 1}
POOLINGS = {"statistics_pooling": 0, "self_attention": 1}
RELU_TYPES = {"relu": 0, "prelu": 1, "lrelu": 2}

_VP = C.c_void_p
_SZ = C.c_size_t
_I = C.c_int
_F = C.c_float

# name -> (restype, argtypes).  Kept in header order; tests/test_abi.py checks that every
# prototype of include/xvector_hip.h is listed here and exported by the .so.
SIGNATURES = {
    "xv_last_error": (C.c_char_p, []),
    "xv_abi_version": (_I, []),
    "xv_device_count": (_I, []),
    "xv_profile_begin": (_I, [_I]),
    "xv_profile_begin_kinds": (_I, [_I, C.c_uint32]),
    "xv_profile_end": (_I, [C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "xv_debug_nt_schedule": (_I, [_I, _I, _I, _I, _I]),
    "xv_copy_2d": (_I, [_VP, _VP, _SZ, _VP, _SZ, _I, _I]),
    "xv_op_workspace_bytes": (_SZ, [_I, _I, _I]),
    "xv_pad_channels": (_I, [_VP, _VP, _I, _I, _VP, _I]),
    "xv_prep_weight_fwd": (_I, [_VP, _VP, _I, _I, _I, _VP, _I]),
    "xv_prep_weight_dgrad": (_I, [_VP, _VP, _I, _I, _I, _VP]),
    "xv_affine_forward": (_I, [_VP, _VP, _I, _I, _I, _I, _VP, _VP, _VP, _I, _I, _VP, _VP, _SZ]),
    "xv_affine_dgrad": (_I, [_VP, _VP, _I, _I, _I, _I, _VP, _VP, _I, _VP, _SZ]),
    "xv_affine_wgrad": (_I, [_VP, _VP, _I, _I, _I, _I, _I, _VP, _I, _I, _I, _VP, _F, _VP, _VP, _SZ]),
    "xv_colsum": (_I, [_VP, _VP, _I, _I, _I, _VP, _VP, _SZ]),
    "xv_col_stats": (_I, [_VP, _VP, _I, _I, _I, _VP]),
    "xv_bn_finalize": (_I, [_VP, _VP, _I, _I, _VP, _VP, _F, _F, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I]),
    "xv_bn_inference_scale": (_I, [_VP, _I, _VP, _VP, _VP, _VP, _F, _VP, _VP]),
    "xv_bn_apply": (_I, [_VP, _VP, _I, _I, _I, _VP, _VP, _I, _VP, _I]),
    "xv_bn_relu_backward": (_I, [_VP, _VP, _VP, _I, _I, _I, _VP, _VP, _VP, _VP, _VP, _I, _I, _VP, _VP, _VP, _VP, _VP, _SZ]),
    "xv_prelu_forward": (_I, [_VP, _VP, _I, _I, _VP, _VP]),
    "xv_set_activation": (_I, [_VP, _VP]),
    "xv_relu_backward": (_I, [_VP, _VP, _VP, _SZ, _VP]),
    "xv_amax": (_I, [_VP, _VP, _SZ, _VP]),
    "xv_split_planes": (_I, [_VP, _VP, _I, _I, _I, _VP, _I, _SZ, _VP]),
    "xv_bn_apply_split": (_I, [_VP, _VP, _I, _I, _I, _VP, _VP, _I, _VP, _VP, _I, _SZ]),
    "xv_bn_output_range": (_I, [_VP, _VP, _I, _I, _VP, _VP, _I, _VP, _VP, _VP]),
    "xv_bn_relu_backward_split": (_I, [_VP, _VP, _VP, _I, _I, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I, _I, _VP, _I, _SZ, _VP, _VP, _VP,
                                       _VP, _VP, _SZ]),
    "xv_affine_forward_f16x3": (_I, [_VP, _VP, _SZ, _VP, _I, _I, _I, _I, _VP, _SZ, _VP, _VP, _VP, _I, _I, _VP]),
    "xv_affine_dgrad_f16x3": (_I, [_VP, _VP, _SZ, _VP, _I, _I, _I, _I, _VP, _SZ, _VP, _VP, _I]),
    "xv_affine_dgrad_bnstats_f16x3": (_I, [_VP, _VP, _SZ, _VP, _I, _I, _I, _I, _VP, _SZ, _VP, _VP, _I, _VP, _VP, _VP, _VP, _VP, _VP]),
    "xv_bn_relu_backward_split_from_part": (_I, [_VP, _VP, _I, _VP, _VP, _I, _I, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I, _VP, _I, _SZ,
                                                 _VP, _VP, _VP, _VP, _VP, _SZ]),
    "xv_affine_wgrad_f16x3": (_I, [_VP, _VP, _SZ, _VP, _I, _I, _I, _I, _I, _VP, _SZ, _VP, _I, _I, _I, _I, _VP, _F, _VP, _VP, _SZ]),
    "xv_stat_pool_forward": (_I, [_VP, _VP, _I, _I, _I, _VP]),
    "xv_stat_pool_backward": (_I, [_VP, _VP, _VP, _VP, _I, _I, _I, _VP]),
    "xv_stat_pool_forward_bn": (_I, [_VP, _VP, _I, _I, _I, _VP, _VP, _I, _VP, _VP]),
    "xv_bn_relu_backward_pooled": (_I, [_VP, _VP, _VP, _VP, _I, _I, _VP, _I, _VP, _VP, _VP, _VP, _VP, _I, _VP, _VP, _VP, _VP, _VP, _SZ]),
    "xv_stat_pool_forward_bn_aux": (_I, [_VP, _VP, _I, _I, _I, _VP, _VP, _I, _VP, _VP, _VP, _VP]),
    "xv_bn_relu_backward_pooled_aux": (_I, [_VP, _VP, _VP, _VP, _VP, _I, _I, _VP, _I, _VP, _VP, _VP, _VP, _VP, _I, _VP, _VP, _VP, _VP, _VP, _SZ]),
    "xv_bn_relu_backward_pooled_split": (_I, [_VP, _VP, _VP, _VP, _I, _I, _VP, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I, _VP, _I, _SZ, _VP,
                                              _VP, _VP, _VP, _VP, _SZ]),
    "xv_ring_loss": (_I, [_VP, _VP, _I, _I, _I, _VP, _F, _VP, _VP, _VP]),
    "xv_mhe_loss": (_I, [_VP, _VP, _I, _I, _I, _VP, _I, _F, _VP, _VP, _VP]),
    "xv_mhe_add_grad": (_I, [_VP, _VP, _I, _I, _I, _VP, _VP]),
    "xv_att_score": (_I, [_VP, _VP, _I, _I, _I, _I, _VP, _F, _VP]),
    "xv_softmax_segments": (_I, [_VP, _VP, _I, _I, _VP]),
    "xv_softmax_segments_backward": (_I, [_VP, _VP, _VP, _I, _I, _VP]),
    "xv_att_pool_backward_weights": (_I, [_VP, _VP, _I, _I, _I, _VP, _VP, _I, _VP, _VP, _VP]),
    "xv_att_key_backward": (_I, [_VP, _VP, _I, _I, _I, _VP, _F, _VP, _VP, _VP, _VP, _VP, _SZ]),
    "xv_key_activation": (_I, [_VP, _VP, _SZ, _I, _VP]),
    "xv_add_inplace": (_I, [_VP, _VP, _VP, _SZ]),
    "xv_l2_scaling_forward": (_I, [_VP, _VP, _I, _I, _F, _VP]),
    "xv_l2_scaling_backward": (_I, [_VP, _VP, _VP, _I, _I, _F, _VP]),
    "xv_loss_prep_weight": (_I, [_VP, _VP, _I, _I, _I, _VP, _VP, _I, _VP]),
    "xv_margin_softmax_rows": (_I, [_VP, _I, _VP, _I, _I, _I, _VP, _I, _VP, _F, _F, _VP, _VP, _VP, _VP]),
    "xv_cm_decode": (_I, [_VP, _VP, _I, _I, _I, _SZ, _VP]),
    "xv_cm_decode_ragged": (_I, [_VP, _VP, _VP, _VP, _I, _I, _I, _VP]),
    "xv_add_norm_grad": (_I, [_VP, _VP, _VP, _I, _I, _VP]),
    "xv_segment_gemm": (_I, [_VP, _VP, C.c_long, _VP, C.c_long, _I, _I, _I, _VP, _VP, _VP, _VP, C.c_long, _VP, C.c_long, _VP, _SZ, _VP]),
    "xv_segment_affine_bn_forward": (_I, [_VP, _VP, C.c_long, _VP, C.c_long, _I, _I, _I, _VP, _VP, _VP, _F, _F, _I, _VP, _VP, _VP, _VP, _VP,
                                          _VP, _VP, _I, _VP, _VP, _SZ, _VP]),
    "xv_segment_dgrad_bn_backward": (_I, [_VP, _VP, C.c_long, _VP, C.c_long, _I, _I, _I, _VP, _VP, _VP, C.c_long, _VP, _VP, _VP, _VP, _VP,
                                          _VP, _I, _VP, _VP, _VP, _VP, _VP, _SZ, _VP]),
    "xv_loss_weight_backward": (_I, [_VP, _VP, _I, _VP, _I, _VP, _VP, _I, _I, _I, _F, _VP, _VP, _SZ]),
    "xv_l2_reg_loss": (_I, [_VP, _VP, _SZ, _F, _VP]),
    "xv_sgd_update": (_I, [_VP, _VP, _VP, _SZ, _F, _F]),
    "xv_momentum_update": (_I, [_VP, _VP, _VP, _VP, _SZ, _F, _F, _I, _F]),
    "xv_adam_update": (_I, [_VP, _VP, _VP, _VP, _VP, _SZ, _F, _F, _F, _F, _I, _F]),
    "xv_sumsq": (_I, [_VP, _VP, _SZ, _VP]),
    "xv_engine_create": (_I, [C.POINTER(XvConfig), C.POINTER(_VP)]),
    "xv_engine_destroy": (None, [_VP]),
    "xv_engine_num_variables": (_I, [_VP]),
    "xv_engine_variable_info": (_I, [_VP, _I, C.POINTER(C.c_char_p), c_int32_p, c_int32_p, C.POINTER(_SZ), c_int32_p]),
    "xv_engine_variables_count": (_SZ, [_VP]),
    "xv_engine_trainable_count": (_SZ, [_VP]),
    "xv_engine_optimizer_state_count": (_SZ, [_VP]),
    "xv_engine_bind": (_I, [_VP, _VP, _VP, _VP]),
    "xv_engine_forward": (_I, [_VP, _VP, _VP, _I, _I, _I]),
    "xv_engine_forward_lengths": (_I, [_VP, _VP, _VP, _I, _I, _VP]),
    "xv_engine_loss_forward": (_I, [_VP, _VP, _VP, _I, _I]),
    "xv_engine_backward": (_I, [_VP, _VP, _I]),
    "xv_engine_backward_async": (_I, [_VP, _VP, _I]),
    "xv_engine_stage_wait": (_I, [_VP, _VP, _I]),
    "xv_engine_allreduce": (_I, [_VP, _VP, _I, _VP]),
    "xv_engine_allreduce_wait": (_I, [_VP, _VP]),
    "xv_engine_stage_grad_range": (_I, [_VP, _I, C.POINTER(_SZ), C.POINTER(_SZ)]),
    "xv_engine_apply": (_I, [_VP, _VP, _F, _F, _I]),
    "xv_engine_arena_bytes": (_SZ, [_VP]),
    "xv_engine_loss_ptrs": (_I, [_VP, C.POINTER(_VP), C.POINTER(_VP)]),
    "xv_engine_endpoint": (_I, [_VP, C.c_char_p, C.POINTER(_VP), c_int32_p, c_int32_p, c_int32_p]),
    "xv_engine_set_concurrency": (_I, [_VP, _I]),
    "xv_engine_invalidate_weights": (_I, [_VP]),
}

XV_BWD_STAGES = 4
XV_MAX_FRAME_LAYERS = 12
_lib = None


def load():
    """Load libxvector_hip.so (built in-tree by `make -C tf_kaldi_speaker_amd/csrc`,
    or `python -c 'import __graft_entry__ as g; g.build()'`).  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise XvError(
            "HIP extension %s is missing - build it with `make -C %s` (hipcc, gfx950). "
            "There is no CPU fallback." % (LIB_PATH, os.path.join(_HERE, "csrc")))
    # torch ships its own libamdhip64 (same SONAME as /opt/rocm's).  Whichever HIP runtime is mapped first serves the whole
    # process: if this library came first it would bind the system runtime and torch's CUDA initialisation would then find
    # "no ROCm-capable device".  Importing torch first makes both share torch's runtime (device buffers are torch's anyway).
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError here == ABI mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    if lib.xv_abi_version() != ABI_VERSION:
        raise XvError("ABI version mismatch: %s reports %d, this package binds version %d (stale build? run `make -C %s`)"
                      % (LIB_PATH, lib.xv_abi_version(), ABI_VERSION, os.path.join(_HERE, "csrc")))
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().xv_last_error()
        raise XvError("%s failed (rc=%d): %s" % (what or "xv call", rc, msg.decode() if msg else "?"))


def call(name, *args):
    """Call an int-returning entry point and raise XvError on failure."""
    check(getattr(load(), name)(*args), name)


def share_gpu():
    """XV_SHARE_GPU=1 (tests of the multi-rank paths on a 1-GPU box): ranks are spread round-robin over the visible devices and
    talk over gloo, which carries device tensors through the host - RCCL refuses two ranks on one device."""
    return os.environ.get("XV_SHARE_GPU") == "1"


def local_device_index():
    """HIP device of this rank: LOCAL_RANK, modulo the device count in XV_SHARE_GPU mode."""
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if share_gpu():
        import torch
        return local_rank % max(torch.cuda.device_count(), 1)
    return local_rank


def dist_backend():
    return "gloo" if share_gpu() else "nccl"

