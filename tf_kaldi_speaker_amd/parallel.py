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
    """Callable handed to Engine.train_step: all-reduces finished gradient slices asynchronously."""

    def __init__(self, dist, world_size, always=False):
        self.dist = dist
        self.world_size = int(world_size)
        self.grad_scale = 1.0 / float(self.world_size)
        self.always = bool(always)       # issue the collective even on one rank (tests of the RCCL wiring on a 1-GPU box)
        self._pending = []
        self._comm = None                # communication stream (GPU tensors only)

    def __call__(self, flat_slice, ready=None):
        """ready(stream_ptr), if given, makes that HIP stream wait until the slice is complete (Engine.stage_wait)."""
        if (self.world_size == 1 and not self.always) or flat_slice.numel() == 0:
            return
        if ready is not None and flat_slice.is_cuda:
            if self._comm is None:
                self._comm = torch.cuda.Stream(device=flat_slice.device)
            with torch.cuda.stream(self._comm):
                ready(self._comm.cuda_stream)
                self._pending.append(self.dist.all_reduce(flat_slice, op=self.dist.ReduceOp.SUM, async_op=True))
            return
        self._pending.append(self.dist.all_reduce(flat_slice, op=self.dist.ReduceOp.SUM, async_op=True))

    def wait(self):
        """Make the current stream wait for every outstanding slice (called before the update)."""
        for w in self._pending:
            w.wait()
        self._pending = []


def average_bn_statistics(dist, variables, n_trainable, world_size):
    """Mean of the non-trainable tail (BN moving mean / variance) of the flat variables buffer."""
    if world_size == 1:
        return
    tail = variables[n_trainable:]
    dist.all_reduce(tail, op=dist.ReduceOp.SUM)
    tail.mul_(1.0 / world_size)


def broadcast_variables(dist, variables, src=0):
    dist.broadcast(variables, src=src)
