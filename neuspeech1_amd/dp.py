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
    def __init__(self, grad_flat: torch.Tensor, group=None, force: bool = False, timing: bool = False):
        self.G = grad_flat
        self.group = group
        self.force = force        # run the collective even at world size 1 (tests of the stream / event plumbing)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.cuda = grad_flat.is_cuda
        self.side = torch.cuda.Stream(device=grad_flat.device) if self.cuda else None
        backend = dist.get_backend(group) if dist.is_initialized() else ""
        self.native_avg = backend == "nccl"   # RCCL implements ReduceOp.AVG; gloo does not
        self.chunks = []
        self.bytes_per_step = 0
        self._waits = []          # (event before, event after) the optimizer stream's wait for the side stream, per step
        # chunk-ready events are allocated ONCE and reused round-robin (more than any step's chunk count): nothing is created
        # on the step path.  `timing` (bench.py, tests; off in production) adds side-stream events around every chunk's
        # collective and around the optimizer stream's wait: total vs exposed all-reduce time.  Each history entry OWNS its
        # events (taken from a free list that retired entries feed), so a step with any number of chunks cannot re-record an
        # event that an entry still in the history holds.
        self._ready_ev = [torch.cuda.Event() for _ in range(16)] if self.cuda else []
        self._ready_i = 0
        self.timing = timing
        self._free_ev = []
        self._chunk_t = []        # per step: [(event before, event after) on the side stream per chunk]
        self._cur_t = []
        self.keep = 256           # steps of timing history kept

    def on_ready(self, lo: int, hi: int):
        """G[lo:hi] is final on the current stream: start its all-reduce on the side stream."""
        if (self.world == 1 and not self.force) or hi <= lo:
            return
        self.chunks.append((lo, hi))
        self.bytes_per_step += 4 * (hi - lo)
        view = self.G[lo:hi]
        if self.cuda:
            ev = self._ready_ev[self._ready_i % len(self._ready_ev)]
            self._ready_i += 1
            ev.record(torch.cuda.current_stream())
            self.side.wait_event(ev)
            with torch.cuda.stream(self.side):
                if self.timing:
                    t0, t1 = self._tev(), self._tev()
                    t0.record(self.side)
                    self._reduce(view)
                    t1.record(self.side)
                    self._cur_t.append((t0, t1))
                else:
                    self._reduce(view)
        else:
            self._reduce(view)

    def _tev(self):
        """a timing event no history entry holds"""
        return self._free_ev.pop() if self._free_ev else torch.cuda.Event(enable_timing=True)

    def _reduce(self, view):
        if self.native_avg:
            dist.all_reduce(view, op=dist.ReduceOp.AVG, group=self.group)
        else:
            dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)
            view.mul_(1.0 / self.world)

    def finish(self):
        """Make the optimizer (current stream) wait for every chunk."""
        if self.cuda and (self.world > 1 or self.force):
            cur = torch.cuda.current_stream()
            if self.timing:
                e0, e1 = self._tev(), self._tev()
                e0.record(cur)
                cur.wait_stream(self.side)
                e1.record(cur)
                self._waits.append((e0, e1, self.bytes_per_step))
                self._chunk_t.append(self._cur_t)
                self._cur_t = []
                if len(self._waits) > self.keep:        # the oldest entry retires: its events go back to the free list
                    w, c = self._waits.pop(0), self._chunk_t.pop(0)
                    self._free_ev += [w[0], w[1]] + [e for pair in c for e in pair]
            else:
                cur.wait_stream(self.side)
        self.chunks.clear()
        self.bytes_per_step = 0

    def exposed_ms(self, last: int | None = None):
        """Per step: how long the optimizer's stream sat waiting for the all-reduce stream AFTER backward had finished,
        i.e. the collective time that backward did not hide (call after a device synchronize).  Returns (mean exposed ms
        over the `last` recorded steps, bytes reduced per step)."""
        w = self._waits[-last:] if last else self._waits
        if not w:
            return 0.0, 0
        return sum(e0.elapsed_time(e1) for e0, e1, _ in w) / len(w), w[-1][2]

    def total_ms(self, last: int | None = None):
        """Per step: the collective time itself (sum over the chunks of the side stream's time inside all_reduce), the
        quantity `exposed_ms` is a part of: exposed < total means backward hid some of it (call after a synchronize)."""
        c = self._chunk_t[-last:] if last else self._chunk_t
        if not c:
            return 0.0
        return sum(sum(t0.elapsed_time(t1) for t0, t1 in step) for step in c) / len(c)
