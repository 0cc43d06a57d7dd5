"""Data parallelism for the x-vector engine: one process per GPU, replicas of all variables,
per-GPU minibatches, sum-all-reduce of the flat fp32 gradient buffer over RCCL/xGMI.

The reference has no multi-GPU path (the tower code was withheld,
model/trainer.py:209,349); this is a new design.  Semantics:
  * every rank draws its own minibatch (the reference loaders are independently seeded,
    dataset/data_loader.py:261-262);
  * BatchNorm uses LOCAL batch statistics (the analogue of per-tower statistics);
  * the loss is a batch mean (tf.losses.sparse_softmax_cross_entropy), so averaging the
    per-rank gradients equals the gradient of the global-batch mean loss;
  * the gradient buffer is reduced in XV_BWD_STAGES slices, each launched as soon as its
    backward stage has been enqueued, so the collective of the (large) speaker-matrix slice
    overlaps the TDNN backward.  The slices are enqueued on a communication stream that waits for
    the engine's end-of-stage events (xv_engine_stage_wait): the compute stream itself never
    waits for the weight-gradient stream in mid-pass, which would cost the overlap of the two
    (0.2 ms of a 2.8 ms step at S1).  The 1/world factor is folded into the optimiser step.
  * BN moving averages stay per-rank during an epoch and are averaged by
    `average_bn_statistics` before a checkpoint is written.
"""
import torch


class GradAllReduce(object):
    """Callable handed to Engine.train_step: all-reduces finished gradient slices asynchronously.

    timing=True (bench.py, N > 1): every slice's collective is bracketed by events on the communication stream - the start event
    after the stream has waited for the slice, the end event after the stream has waited for the collective - and
    mark_compute_done() (called by Engine.train_step when the whole backward pass has been enqueued) drops an event on the
    compute stream, so timing_report() can say how long each slice's all-reduce took and how much of the communication was
    still running when the compute stream had nothing left to do (`exposed_ms`; overlap_frac = 1 - exposed / total)."""

    def __init__(self, dist, world_size, always=False, timing=False):
        self.dist = dist
        self.world_size = int(world_size)
        self.grad_scale = 1.0 / float(self.world_size)
        self.always = bool(always)       # issue the collective even on one rank (tests of the RCCL wiring on a 1-GPU box)
        self.timing = bool(timing)
        self._pending = []
        self._comm = None                # communication stream (GPU tensors only)
        self._steps = []                 # timing: per step [(numel, start_event, end_event), ...] + the compute-done event
        self._cur = None

    def __call__(self, flat_slice, ready=None):
        """ready(stream_ptr), if given, makes that HIP stream wait until the slice is complete (Engine.stage_wait)."""
        if (self.world_size == 1 and not self.always) or flat_slice.numel() == 0:
            return
        if ready is not None and flat_slice.is_cuda:
            if self._comm is None:
                self._comm = torch.cuda.Stream(device=flat_slice.device)
            with torch.cuda.stream(self._comm):
                ready(self._comm.cuda_stream)
                if self.timing:
                    if self._cur is None:
                        self._cur = {"slices": [], "done": None}
                    start = torch.cuda.Event(enable_timing=True)
                    start.record()
                work = self.dist.all_reduce(flat_slice, op=self.dist.ReduceOp.SUM, async_op=True)
                if self.timing:
                    work.wait()          # the communication stream (not the host, with RCCL) waits for the collective
                    end = torch.cuda.Event(enable_timing=True)
                    end.record()
                    self._cur["slices"].append((int(flat_slice.numel()), start, end))
                self._pending.append(work)
            return
        self._pending.append(self.dist.all_reduce(flat_slice, op=self.dist.ReduceOp.SUM, async_op=True))

    def mark_compute_done(self):
        """Everything the step computes before the update has been enqueued on the current stream."""
        if self.timing and self._cur is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self._cur["done"] = ev

    def wait(self):
        """Make the current stream wait for every outstanding slice (called before the update)."""
        for w in self._pending:
            w.wait()
        self._pending = []
        if self._cur is not None:
            self._steps.append(self._cur)
            self._cur = None

    def reset_timing(self):
        self._steps, self._cur = [], None

    def timing_report(self):
        """Per-slice mean all-reduce time over the recorded steps (call after a device synchronize)."""
        if not self.timing or not self._steps:
            return None
        n = len(self._steps[0]["slices"])
        per = [[] for _ in range(n)]
        exposed, total = [], []
        for st in self._steps:
            if len(st["slices"]) != n:
                continue
            ms = [a.elapsed_time(b) for _, a, b in st["slices"]]
            for i, v in enumerate(ms):
                per[i].append(v)
            total.append(sum(ms))
            if st["done"] is not None:
                exposed.append(max(0.0, st["done"].elapsed_time(st["slices"][-1][2])))
        slices = [{"mbytes": round(self._steps[0]["slices"][i][0] * 4 / 1e6, 2), "allreduce_ms": round(sum(per[i]) / max(len(per[i]), 1), 4),
                   "bus_gbs": round(2.0 * (self.world_size - 1) / self.world_size * self._steps[0]["slices"][i][0] * 4 / 1e9
                                    / max(sum(per[i]) / max(len(per[i]), 1) * 1e-3, 1e-9), 1)} for i in range(n)]
        tot = sum(total) / max(len(total), 1)
        exp = sum(exposed) / max(len(exposed), 1) if exposed else None
        return {"steps": len(total), "slices": slices, "allreduce_ms_per_step": round(tot, 4),
                "exposed_ms_per_step": None if exp is None else round(exp, 4),
                "overlap_frac": None if exp is None or tot <= 0 else round(max(0.0, 1.0 - exp / tot), 4)}


def average_bn_statistics(dist, variables, n_trainable, world_size):
    """Mean of the non-trainable tail (BN moving mean / variance) of the flat variables buffer."""
    if world_size == 1:
        return
    tail = variables[n_trainable:]
    dist.all_reduce(tail, op=dist.ReduceOp.SUM)
    tail.mul_(1.0 / world_size)


def broadcast_variables(dist, variables, src=0):
    dist.broadcast(variables, src=src)
