"""Data-parallel gradient exchange: the ONLY collective on the path.

One process per GPU; every rank holds a full replica and a disjoint batch shard
(finetune.py:115-122 -> torch DDP in the reference).  Here the flat fp32
trainable-gradient buffer is all-reduced (AVG) with RCCL over xGMI in chunks
that follow backward-completion order (LoRA of the upper encoder layers, LoRA
of the lower layers, conv stem) on a side HIP stream, so only the last chunk
can be exposed.  The grad-norm / inf check / AdamW run after the reduction, so
every rank takes identical decisions (skip-on-inf included) with no further
collective.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, grad_flat: torch.Tensor, group=None, force: bool = False):
        self.G = grad_flat
        self.group = group
        self.force = force        # run the collective even at world size 1 (tests of the stream / event plumbing)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.cuda = grad_flat.is_cuda
        self.side = torch.cuda.Stream(device=grad_flat.device) if self.cuda else None
        backend = dist.get_backend(group) if dist.is_initialized() else ""
        self.native_avg = backend == "nccl"   # RCCL implements ReduceOp.AVG; gloo does not
        self.chunks = []

    def on_ready(self, lo: int, hi: int):
        """G[lo:hi] is final on the current stream: start its all-reduce on the side stream."""
        if (self.world == 1 and not self.force) or hi <= lo:
            return
        self.chunks.append((lo, hi))
        view = self.G[lo:hi]
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.side.wait_event(ev)
            with torch.cuda.stream(self.side):
                self._reduce(view)
        else:
            self._reduce(view)

    def _reduce(self, view):
        if self.native_avg:
            dist.all_reduce(view, op=dist.ReduceOp.AVG, group=self.group)
        else:
            dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)
            view.mul_(1.0 / self.world)

    def finish(self):
        """Make the optimizer (current stream) wait for every chunk."""
        if self.cuda and (self.world > 1 or self.force):
            torch.cuda.current_stream().wait_stream(self.side)
        self.chunks.clear()
