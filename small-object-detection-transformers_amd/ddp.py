"""Data-parallel gradient exchange: one process per GPU, RCCL all-reduce over xGMI.

The reference wraps the model in torch DDP (Train.py:264-266: bucketed MEAN all-reduce of
22,007,851 f32 gradients per backward, call site C1 of SURVEY.md section 2.3) and
compensates with ``loss *= world_size`` (Train.py:439-440).  ``attach(model)`` keeps exactly
that contract (``average=True``, torch DDP's mean), so the reference's training loop ports
unchanged; ``average=False`` gives the plain SUM for loops that dropped the loss scaling.

The engine already keeps every gradient in one contiguous f32 buffer laid out in reverse
order of completion, so the exchange is a few large collectives: every bucket whose
gradients are final (head + necks + stage 3, stage 2 + PatchMerging, ...) is all-reduced
asynchronously while the backward of the earlier layers keeps the GPU busy
(``reduce_async``, issued at split points recorded in the backward launch plan); the last
slice follows the last weight gradient (``finish``).  88 MB are ~1 ms ring-bound on
7 x 153 GB/s links against ~30 ms of backward.

Gradient accumulation (Train.py:125,448: nbs 64 / batch 16 -> 4 micro-steps per optimizer
step): the weight-gradient kernels accumulate (``+=``) into the flat buffer, so a second
all-reduce of a buffer that already holds reduced values would count them ``world`` times.
Two ways to stay correct, both covered by tests/test_ddp_*.py:
  * ``with reducer.no_sync():`` around all but the last micro-step (torch DDP's idiom):
    one all-reduce per optimizer step;
  * nothing: ``begin_backward`` rescales an already-reduced buffer by 1/world before the
    next backward adds to it (SUM mode; MEAN mode needs no correction), so reducing on every
    micro-step - what the reference's DDP does - gives the same sums.

torch.distributed's "nccl" backend IS RCCL on ROCm; the tests drive the same code with gloo.
"""
from __future__ import annotations

import contextlib

import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, group=None, average: bool = True):
        if not dist.is_initialized():
            raise RuntimeError("init torch.distributed first (backend 'nccl' = RCCL on ROCm)")
        self.group = group
        self.world = dist.get_world_size(group)
        self.average = average
        self.sync = True       # False inside no_sync(): backward only accumulates locally
        self._dirty = False    # the buffer holds values that have already been all-reduced
        self._pending = []     # (work handle, tensor) of collectives launched by reduce_async

    @contextlib.contextmanager
    def no_sync(self):
        """Skip the all-reduce for the backward passes inside the block (gradient accumulation): the next
        backward outside it reduces the accumulated buffer once."""
        old, self.sync = self.sync, False
        try:
            yield self
        finally:
            self.sync = old

    def begin_backward(self, flat_grad: torch.Tensor, fresh: bool) -> None:
        """Called by the engine before a backward adds into ``flat_grad``.  fresh: the buffer was just zeroed."""
        if fresh:
            self._dirty = False
        elif self._dirty and self.world > 1 and not self.average:
            flat_grad.mul_(1.0 / self.world)        # so that the coming SUM restores the reduced part exactly once
            self._dirty = False

    def _active(self) -> bool:
        return self.world > 1 and self.sync

    def reduce(self, flat_grad: torch.Tensor) -> None:
        if not self._active():
            return
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=self.group)
        if self.average:
            flat_grad.mul_(1.0 / self.world)
        self._dirty = True

    def reduce_async(self, part: torch.Tensor) -> None:
        """Start the all-reduce of a finished slice of the gradient buffer; it runs on the backend's own stream, ordered
        after the work already queued on the current stream, while later kernels keep the GPU busy.  Nothing may write
        the slice until finish()."""
        if not self._active():
            return
        self._pending.append((dist.all_reduce(part, op=dist.ReduceOp.SUM, group=self.group, async_op=True), part))

    def finish(self, rest: torch.Tensor) -> None:
        """All-reduce the remaining slice, then make the current stream wait for the asynchronous ones."""
        if not self._active():
            return
        dist.all_reduce(rest, op=dist.ReduceOp.SUM, group=self.group)
        if self.average:
            rest.mul_(1.0 / self.world)
        for work, part in self._pending:
            work.wait()
            if self.average:
                part.mul_(1.0 / self.world)
        self._pending = []
        self._dirty = True


def attach(model, group=None, average: bool = True, broadcast: bool = True):
    """Use instead of torch DDP: synchronises parameters from rank 0 once and makes every
    backward all-reduce the engine's flat gradient buffer.  Returns the model; the reducer is
    ``model.grad_reducer`` (``with model.grad_reducer.no_sync(): ...`` for accumulation steps)."""
    red = GradReducer(group, average)
    if broadcast and red.world > 1:
        for t in list(model.parameters()) + list(model.buffers()):
            dist.broadcast(t.data, src=0, group=group)
    if next(model.parameters()).is_cuda:
        model._get_engine().ddp = red
        model._get_engine().invalidate_params()     # the broadcast wrote the masters through p.data
    else:
        model._pending_ddp = red
    model.grad_reducer = red
    return model


def shard_batch(total_batch: int, rank: int, world: int):
    """Train.py:682-683: per-rank batch = total // world."""
    per = total_batch // world
    return rank * per, per
