"""Host-side handle on the native engine (csrc/xv_engine.hip).

torch is used here for exactly three things: owning device memory (the flat
variables / gradient / optimiser-state buffers), naming the HIP stream, and
(in `parallel`) the RCCL all-reduce.  All arithmetic happens in the C-ABI.
"""
from collections import OrderedDict
import ctypes as C
import os

import numpy as np
import torch

try:
    from . import _lib
    from ._lib import XvConfig, XvError, LOSS_KINDS, OPTIMIZERS, PRECISIONS, POOLINGS, RELU_TYPES, XV_BWD_STAGES
except ImportError:      # drop-in layout: PYTHONPATH=$TF_KALDI_ROOT makes these top-level modules
    import _lib
    from _lib import XvConfig, XvError, LOSS_KINDS, OPTIMIZERS, PRECISIONS, POOLINGS, RELU_TYPES, XV_BWD_STAGES


# How tdnn1-5's contractions are evaluated unless make_config(precision=...) / the config key "precision" / $XV_PRECISION
# says otherwise.
#   "f32":   (default) fp32 operands on the fp32-input MFMA (v_mfma_f32_32x32x2_f32): the reference's dtype, exact fp32
#            products, fp32 accumulation.
#   "f16x3": opt-in.  Every fp32 operand is carried as two fp16 planes (hi + lo, power-of-two scaled: 22-bit significands)
#            and each product is three fp16 MFMA products accumulated in fp32 - measured 5e-7..8e-7 of the output scale
#            against float64 (the fp32 path measures 1.7e-7; both far inside the 1e-4 embedding tolerance) at ~2x the
#            speed; passes the same parity tests at the same tolerances, but its operands are narrower than fp32, so it is
#            never selected silently and bench.py reports it separately.
DEFAULT_PRECISION = "f32"


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def make_config(feat_dim, num_speakers=0, loss_func="softmax", margin_m=0.0, lambda_min=0.0, lambda_base=1000.0,
                lambda_gamma=1e-4, lambda_power=5.0, num_nodes_pooling_layer=1500, num_nodes_last_layer=512,
                last_layer_no_bn=False, last_layer_linear=False, feature_norm=False, feature_scaling_factor=1.0,
                weight_l2_regularizer=1e-2, output_weight_l2_regularizer=None, batchnorm_momentum=0.99,
                bn_epsilon=1e-3, fused_bn_unbiased_moving_var=True, optimizer="sgd", momentum=0.9, use_nesterov=False,
                clip_gradient_norm=0.0, max_batch=128, max_frames=400, precision=None, pooling_type="statistics_pooling",
                att_key_num_nodes=(1500, 1500), att_key_network_type=3, att_use_scale=True, aux_loss_func=(), ring_loss_init=20.0,
                ring_loss_lambda=0.01, mhe_lambda=0.01, frame_layers=None, network_relu_type="relu", max_rows=0):
    if pooling_type not in POOLINGS:
        raise NotImplementedError("Not implement %s pooling" % pooling_type)
    if loss_func not in LOSS_KINDS:
        raise NotImplementedError("Not implement %s loss" % loss_func)
    if optimizer not in OPTIMIZERS:
        raise SystemExit("Optimizer %s is not supported." % optimizer)
    c = XvConfig()
    c.feat_dim = int(feat_dim)
    c.num_speakers = int(num_speakers or 0)
    c.num_nodes_pooling_layer = int(num_nodes_pooling_layer)
    c.num_nodes_last_layer = int(num_nodes_last_layer)
    c.last_layer_no_bn = int(bool(last_layer_no_bn))
    c.last_layer_linear = int(bool(last_layer_linear))
    c.feature_norm = int(bool(feature_norm))
    c.feature_scaling_factor = float(feature_scaling_factor)
    c.loss_kind = LOSS_KINDS[loss_func]
    c.margin_m = float(margin_m)
    c.lambda_min, c.lambda_base = float(lambda_min), float(lambda_base)
    c.lambda_gamma, c.lambda_power = float(lambda_gamma), float(lambda_power)
    c.weight_l2_regularizer = float(weight_l2_regularizer)
    c.output_weight_l2_regularizer = -1.0 if output_weight_l2_regularizer is None else float(output_weight_l2_regularizer)
    c.batchnorm_momentum = float(batchnorm_momentum)
    c.bn_epsilon = float(bn_epsilon)
    c.fused_bn_unbiased_moving_var = int(bool(fused_bn_unbiased_moving_var))
    c.optimizer = OPTIMIZERS[optimizer]
    c.momentum = float(momentum)
    c.use_nesterov = int(bool(use_nesterov))
    c.clip_gradient_norm = float(clip_gradient_norm)
    c.max_batch = int(max_batch)
    c.max_frames = int(max_frames)
    c.max_rows = int(max_rows or 0)
    if precision is None:
        precision = os.environ.get("XV_PRECISION", DEFAULT_PRECISION)
    if precision not in PRECISIONS:
        raise ValueError("precision must be one of %s" % sorted(PRECISIONS))
    c.precision = PRECISIONS[precision]
    for aux in aux_loss_func or ():
        if aux not in ("ring_loss", "mhe_loss"):
            raise NotImplementedError("Unsupported loss function %s" % aux)
    c.aux_ring = int("ring_loss" in (aux_loss_func or ()))
    c.ring_loss_init, c.ring_loss_lambda = float(ring_loss_init), float(ring_loss_lambda)
    c.aux_mhe = int("mhe_loss" in (aux_loss_func or ()))
    c.mhe_lambda = float(mhe_lambda)
    if network_relu_type not in RELU_TYPES:
        raise NotImplementedError("network_relu_type %r (relu, prelu or lrelu: tdnn.py:24-30)" % network_relu_type)
    c.relu_type = RELU_TYPES[network_relu_type]
    if frame_layers:
        # extended frame-layer table ((context, width), ...): no reference counterpart (model/tdnn.py hard-codes its five layers;
        # BASELINE configs[4] "extended context, 10 layers").  A width of None / 0 in the last entry = num_nodes_pooling_layer.
        table = [(int(k), int(w) if w else int(num_nodes_pooling_layer)) for k, w in frame_layers]
        if not 3 <= len(table) <= _lib.XV_MAX_FRAME_LAYERS:
            raise ValueError("frame_layers: 3..%d layers (got %d)" % (_lib.XV_MAX_FRAME_LAYERS, len(table)))
        if table[-1][1] != int(num_nodes_pooling_layer):
            raise ValueError("frame_layers: the last frame layer is the pooling layer, its width must equal num_nodes_pooling_layer")
        c.num_frame_layers = len(table)
        for i, (k, w) in enumerate(table):
            c.frame_context[i], c.frame_width[i] = k, w
    c.pooling = POOLINGS[pooling_type]
    if pooling_type == "self_attention":
        # the shipped single-head form (nnet_conf/*_tdnn4_att.json): two key layers on tdnn4_relu, value = tdnn5_relu
        if len(att_key_num_nodes) != 2:
            raise NotImplementedError("self_attention: att_key_num_nodes must have two entries (dense+bn+relu, then the key layer)")
        if int(att_key_network_type) not in (0, 1, 2, 3):
            raise NotImplementedError("self_attention: att_key_network_type %r is not one of 0..3 (pooling.py:84-96)" % att_key_network_type)
        c.att_key0_nodes, c.att_key1_nodes = int(att_key_num_nodes[0]), int(att_key_num_nodes[1])
        c.att_key_type = int(att_key_network_type)
        c.att_use_scale = int(bool(att_use_scale))
    return c


class Engine(object):
    """One TDNN x-vector graph resident on one GPU."""

    def __init__(self, config, device="cuda:0"):
        if not torch.cuda.is_available():
            raise XvError("No HIP device visible: the x-vector engine only runs on MI355X (no CPU fallback).")
        self.lib = _lib.load()
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self.config = config
        h = C.c_void_p()
        _lib.check(self.lib.xv_engine_create(C.byref(config), C.byref(h)), "xv_engine_create")
        self.h = h
        self.n_all = self.lib.xv_engine_variables_count(h)
        self.n_train = self.lib.xv_engine_trainable_count(h)
        self.n_opt = self.lib.xv_engine_optimizer_state_count(h)
        self.variables = torch.zeros(self.n_all, dtype=torch.float32, device=self.device)
        self.grads = torch.zeros(self.n_train, dtype=torch.float32, device=self.device)
        self.opt_state = torch.zeros(max(self.n_opt, 4), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.xv_engine_bind(h, _ptr(self.variables), _ptr(self.grads), _ptr(self.opt_state)), "xv_engine_bind")
        self.table = OrderedDict()   # name -> (shape, offset, trainable)
        for i in range(self.lib.xv_engine_num_variables(h)):
            name = C.c_char_p()
            shape = (C.c_int32 * 4)()
            rank, trainable = C.c_int32(), C.c_int32()
            off = C.c_size_t()
            _lib.check(self.lib.xv_engine_variable_info(h, i, C.byref(name), shape, C.byref(rank), C.byref(off),
                                                        C.byref(trainable)), "xv_engine_variable_info")
            self.table[name.value.decode()] = (tuple(shape[:rank.value]), off.value, bool(trainable.value))
        self.update_count = 0
        self._keep = []

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            torch.cuda.synchronize(self.device)
            self.lib.xv_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- variables ----------------------------------------------------------------
    def set_variables(self, values):
        """values: dict name -> array in the TF shape.  Missing names keep their value."""
        host = self.variables.cpu().numpy()
        for name, v in values.items():
            if name not in self.table:
                raise KeyError("unknown variable %s" % name)
            shape, off, _ = self.table[name]
            a = np.asarray(v, np.float32)
            if a.shape != shape:
                raise ValueError("variable %s: shape %s, expected %s" % (name, a.shape, shape))
            host[off:off + a.size] = a.reshape(-1)
        self.variables.copy_(torch.from_numpy(host))
        _lib.check(self.lib.xv_engine_invalidate_weights(self.h))

    def get_variables(self):
        host = self.variables.cpu().numpy()
        out = OrderedDict()
        for name, (shape, off, _) in self.table.items():
            n = int(np.prod(shape))
            out[name] = host[off:off + n].reshape(shape).copy()
        return out

    def get_gradients(self):
        host = self.grads.cpu().numpy()
        out = OrderedDict()
        for name, (shape, off, trainable) in self.table.items():
            if trainable:
                n = int(np.prod(shape))
                out[name] = host[off:off + n].reshape(shape).copy()
        return out

    def init_variables(self, seed=0):
        """Glorot-uniform kernels, zero biases, gamma=1, beta=0, moving_mean=0, moving_variance=1
        ([TF] defaults of tf.layers.conv2d/dense/batch_normalization and xavier_initializer, loss.py:100-102)."""
        rs = np.random.RandomState(seed)
        vals = {}
        for name, (shape, _, _) in self.table.items():
            leaf = name.rsplit("/", 1)[1]
            if leaf == "kernel":
                if len(shape) == 4:
                    fan_in, fan_out = shape[1] * shape[2], shape[1] * shape[3]
                else:
                    fan_in, fan_out = shape
                lim = np.sqrt(6.0 / (fan_in + fan_out))
                vals[name] = rs.uniform(-lim, lim, size=shape).astype(np.float32)
            elif name == "softmax_ringloss/r":
                vals[name] = np.full(shape, self.config.ring_loss_init, np.float32)
            elif leaf == "query":      # truncated_normal(stddev=0.1), pooling.py:131-132
                vals[name] = np.clip(rs.randn(*shape) * 0.1, -0.2, 0.2).astype(np.float32)
            elif leaf in ("gamma", "moving_variance"):
                vals[name] = np.ones(shape, np.float32)
            elif leaf == "alpha":      # tf.constant_initializer(0.01), common.py:37
                vals[name] = np.full(shape, 0.01, np.float32)
            else:
                vals[name] = np.zeros(shape, np.float32)
        self.set_variables(vals)

    # ---- graph execution ------------------------------------------------------------
    def _dev(self, a, dtype):
        if isinstance(a, torch.Tensor):
            t = a.to(device=self.device, dtype=dtype).contiguous()
        else:
            t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32 if dtype == torch.float32 else np.int32)).to(self.device)
        return t

    def forward(self, features, training):
        x = self._dev(features, torch.float32)
        if x.dim() == 2:
            x = x.unsqueeze(0)
        b, t, d = x.shape
        if d != self.config.feat_dim:
            raise ValueError("feature dim %d != %d" % (d, self.config.feat_dim))
        self._keep = [x]
        _lib.check(self.lib.xv_engine_forward(self.h, _stream(), _ptr(x), int(b), int(t), int(bool(training))),
                   "xv_engine_forward")

    @property
    def min_frames(self):
        """Receptive field of the frame-level stack: the fewest frames a chunk needs for one valid output frame (15 for the
        reference's 5 / 5 / 7 / 1 / 1 contexts, model/tdnn.py:39-127)."""
        c = self.config
        ctx = [c.frame_context[i] for i in range(c.num_frame_layers)] if c.num_frame_layers else [5, 5, 7, 1, 1]
        return 1 + sum(int(k) - 1 for k in ctx)

    def forward_lengths(self, features, frames):
        """Inference forward over utterances of different lengths (xv_engine_forward_lengths): features [b, t, d] with chunk i
        holding frames[i] valid frames followed by padding; pooling uses the valid part only."""
        x = self._dev(features, torch.float32)
        n = frames if (isinstance(frames, torch.Tensor) and frames.is_cuda and frames.dtype == torch.int32 and frames.is_contiguous()) else self._dev(frames, torch.int32)
        b, t, d = x.shape
        if d != self.config.feat_dim:
            raise ValueError("feature dim %d != %d" % (d, self.config.feat_dim))
        if n.numel() != b:
            raise ValueError("%d frame counts for %d chunks" % (n.numel(), b))
        if not isinstance(frames, torch.Tensor) or not frames.is_cuda:
            # host-side lengths are checked here (device-resident ones are the caller's: Trainer.predict_batch validates before the upload);
            # the kernels clamp, so a bad length would give an embedding pooled over padding instead of an error
            fh = np.asarray(frames.cpu() if isinstance(frames, torch.Tensor) else frames).reshape(-1)
            if fh.size and (int(fh.min()) < self.min_frames or int(fh.max()) > t):
                raise ValueError("forward_lengths: frame counts must lie in [%d, %d] (got %d..%d)" % (self.min_frames, t, int(fh.min()), int(fh.max())))
        self._keep = [x, n]
        _lib.check(self.lib.xv_engine_forward_lengths(self.h, _stream(), _ptr(x), int(b), int(t), _ptr(n)), "xv_engine_forward_lengths")

    def decode_packed(self, payload, offsets, rows, t):
        """'CM ' matrices of different lengths, packed back to back in `payload` (uint8; kaldi_io.PackedMatrix images at byte
        `offsets`, `rows[i]` frames each) -> device tensor [b, t, feat_dim] with zero padding (xv_cm_decode_ragged)."""
        dev = self.device
        pk = payload if isinstance(payload, torch.Tensor) else torch.from_numpy(np.array(payload, np.uint8))
        pk = pk.to(dev, non_blocking=True)
        def dev_ints(a, dtype):      # device tensors pass through (torch.as_tensor on them cost ~1 ms a call in the extraction driver's profile)
            if isinstance(a, torch.Tensor) and a.is_cuda and a.dtype == dtype:
                return a if a.is_contiguous() else a.contiguous()
            return torch.as_tensor(a, dtype=dtype).to(dev, non_blocking=True).contiguous()
        off = dev_ints(offsets, torch.int64)
        rw = dev_ints(rows, torch.int32)
        b = int(rw.numel())
        out = torch.empty((b, int(t), self.config.feat_dim), dtype=torch.float32, device=dev)
        _lib.check(self.lib.xv_cm_decode_ragged(_stream(), _ptr(pk), _ptr(off), _ptr(rw), b, int(t), int(self.config.feat_dim), _ptr(out)),
                   "xv_cm_decode_ragged")
        self._keep_decode = [pk, off]
        return out, rw

    def loss(self, labels, global_step, with_margin=True):
        y = self._dev(labels, torch.int32)
        self._keep.append(y)
        _lib.check(self.lib.xv_engine_loss_forward(self.h, _stream(), _ptr(y), int(global_step), int(bool(with_margin))),
                   "xv_engine_loss_forward")

    def backward(self, stage=-1):
        _lib.check(self.lib.xv_engine_backward(self.h, _stream(), int(stage)), "xv_engine_backward")

    def backward_async(self, stage):
        """Stage `stage` of the backward pass without the end-of-stage join: the current stream keeps overlapping the weight
        gradients; stage_wait(stage, stream) orders a consumer stream behind the finished slice."""
        _lib.check(self.lib.xv_engine_backward_async(self.h, _stream(), int(stage)), "xv_engine_backward_async")

    def stage_wait(self, stage, stream_ptr):
        _lib.check(self.lib.xv_engine_stage_wait(self.h, C.c_void_p(int(stream_ptr)), int(stage)), "xv_engine_stage_wait")

    def stage_grad_range(self, stage):
        b, e = C.c_size_t(), C.c_size_t()
        _lib.check(self.lib.xv_engine_stage_grad_range(self.h, int(stage), C.byref(b), C.byref(e)))
        return b.value, e.value

    def apply(self, lr, grad_scale=1.0):
        self.update_count += 1
        _lib.check(self.lib.xv_engine_apply(self.h, _stream(), float(lr), float(grad_scale), int(self.update_count)),
                   "xv_engine_apply")

    @property
    def arena_bytes(self):
        return int(self.lib.xv_engine_arena_bytes(self.h))

    def losses(self):
        """(raw_loss, regularization_loss) of the last loss() call - synchronises."""
        raw, reg = C.c_void_p(), C.c_void_p()
        _lib.check(self.lib.xv_engine_loss_ptrs(self.h, C.byref(raw), C.byref(reg)))
        out = torch.empty(2, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.xv_copy_2d(_stream(), _ptr(out), 2, raw, 2, 1, 2), "xv_copy_2d")
        v = out.cpu().numpy()
        return float(v[0]), float(v[1])

    def raw_loss(self):
        """Mean loss of the last loss() call (synchronises); does not evaluate the regulariser."""
        raw = C.c_void_p()
        _lib.check(self.lib.xv_engine_loss_ptrs(self.h, C.byref(raw), None))
        out = torch.empty(1, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.xv_copy_2d(_stream(), _ptr(out), 1, raw, 1, 1, 1), "xv_copy_2d")
        return float(out.cpu().numpy()[0])

    def endpoint(self, name):
        """Copy of an endpoint of the most recent forward as a torch tensor [rows, cols]."""
        p = C.c_void_p()
        rows, cols, ld = C.c_int32(), C.c_int32(), C.c_int32()
        _lib.check(self.lib.xv_engine_endpoint(self.h, name.encode(), C.byref(p), C.byref(rows), C.byref(cols), C.byref(ld)),
                   "xv_engine_endpoint(%s)" % name)
        out = torch.empty((rows.value, cols.value), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.xv_copy_2d(_stream(), _ptr(out), cols.value, p, ld.value, rows.value, cols.value), "xv_copy_2d")
        return out

    # ---- fine-tuning: variables the optimiser must not touch (trainer.py:379-403 noupdate_var_list, :728-773 set_trainable_variables)
    def set_update_filter(self, frozen_names=()):
        """frozen_names: variable names (trainable ones: never updated; BN moving statistics: their UPDATE_OPS are skipped).
        Implemented on the flat buffers: frozen trainables (and their momentum / Adam slots - TF does not create the update op at
        all) are saved before the optimiser step and put back after it, their gradient slices are zeroed first so that a global-norm
        clip sees what TF's restricted var_list sees; frozen moving statistics are restored after the training forward pass."""
        frozen = [n for n in frozen_names if n in self.table]
        grad_ranges, stat_ranges = [], []
        for n in frozen:
            shape, off, trainable = self.table[n]
            cnt = int(np.prod(shape)) if len(shape) else 1
            (grad_ranges if trainable else stat_ranges).append((off, off + cnt))

        def merge(r):
            out = []
            for b, e in sorted(r):
                if out and b <= out[-1][1] + 3:          # variables start on 4-float boundaries
                    out[-1] = (out[-1][0], max(out[-1][1], e))
                else:
                    out.append((b, e))
            return out
        self._frozen_grads, self._frozen_stats = merge(grad_ranges), merge(stat_ranges)
        self.frozen_names = tuple(frozen)

    def train_step(self, features, labels, lr, global_step, allreduce=None, fetch_losses=False):
        """One sess.run(train_op) (trainer.py:505-508).  `allreduce(tensor_slice)` - if given - is
        called after each backward stage on the finished slice of the flat gradient buffer.
        fetch_losses: also return (raw_loss, regularization_loss) evaluated on the PRE-update
        weights, as the reference's logging fetch does (trainer.py:485-499); this synchronises."""
        frozen_stats = getattr(self, "_frozen_stats", None)
        if frozen_stats:
            keep = [self.variables[b:e].clone() for b, e in frozen_stats]
        self.forward(features, True)
        if frozen_stats:
            for (b, e), v in zip(frozen_stats, keep):
                self.variables[b:e].copy_(v)
        self.loss(labels, global_step, True)
        if allreduce is None:
            self.backward(-1)
            grad_scale = 1.0
        else:
            for st in range(XV_BWD_STAGES):
                self.backward_async(st)
                b, e = self.stage_grad_range(st)
                allreduce(self.grads[b:e], ready=lambda stream_ptr, st=st: self.stage_wait(st, stream_ptr))
            if hasattr(allreduce, "mark_compute_done"):
                allreduce.mark_compute_done()
            allreduce.wait()
            grad_scale = allreduce.grad_scale
        frozen = getattr(self, "_frozen_grads", None) or ()
        saved = []
        for b, e in frozen:
            self.grads[b:e].zero_()
            slots = [self.opt_state[k * self.n_train + b:k * self.n_train + e] for k in range(self.n_opt // max(self.n_train, 1))]
            saved.append((self.variables[b:e].clone(), [t.clone() for t in slots]))
        out = self.losses() if fetch_losses else None
        self.apply(lr, grad_scale)
        for (b, e), (v, slots) in zip(frozen, saved):
            self.variables[b:e].copy_(v)
            for k, t in enumerate(slots):
                self.opt_state[k * self.n_train + b:k * self.n_train + e].copy_(t)
        if frozen:
            _lib.check(self.lib.xv_engine_invalidate_weights(self.h), "xv_engine_invalidate_weights")
        return out
