"""Data parallelism for the launch-plan engine: one process per GPU, batch sharded by rank,
gradients summed with RCCL (torch.distributed backend "nccl" on ROCm) over xGMI.

Because the engine keeps every gradient in ONE flat fp32 buffer, the exchange is a handful of
large contiguous all-reduces instead of 138 small ones.  Buckets are contiguous ranges of that
buffer; a bucket is reduced on a side stream as soon as the backward plan has issued the last
launch that writes into it (Plan.last_write), so the collectives overlap the remaining
dgrad/wgrad launches.  The 1/world averaging is folded into the clip/AdamW launches
(``inv_scale``), so no extra pass touches the gradients.  BatchNorm statistics stay per-GPU
(plain DDP semantics; the single-device reference defines nothing else)."""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

from . import graph

_NO_COMM = os.environ.get("MI355_DP_NOCOMM") == "1"


class DataParallel:
    def __init__(self, net, bucket_mb: float = 32.0, overlap: bool = True, process_group=None, force: bool = False):
        self.net = net
        self.engine = net.engine
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.bucket_elems = int(bucket_mb * (1 << 20) / 4)
        self.overlap = overlap
        self._sched = {}
        self.comm_stream = None
        self.inv_scale = 1.0 / self.world
        self.run_calls = None                 # injectable (CPU tests replace the kernel launcher)
        if self.world > 1 or force:
            self.engine.bwd_runner = self._run_backward
        if self.world > 1:
            self.sync_state()

    @torch.no_grad()
    def sync_state(self, src: int = 0):
        """Replicas start from rank ``src``'s parameters and buffers (what DistributedDataParallel does at construction):
        one broadcast of the flat parameter buffer, one per buffer dtype (BatchNorm running_mean / running_var /
        num_batches_tracked).  Without it a rank-dependent initialisation, or a checkpoint loaded on one rank only, gives
        replicas that sum gradients but apply them to different weights."""
        self.engine._check_storage()
        dist.broadcast(self.engine.flat_p, src=src, group=self.group)
        by_dtype = {}
        for b in self.net.buffers():
            by_dtype.setdefault(b.dtype, []).append(b)
        for dtype, bufs in sorted(by_dtype.items(), key=lambda kv: str(kv[0])):
            flat = torch.cat([b.reshape(-1) for b in bufs])
            dist.broadcast(flat, src=src, group=self.group)
            o = 0
            for b in bufs:
                b.copy_(flat[o:o + b.numel()].view(b.shape))
                o += b.numel()

    # ---- bucket schedule: (launch index after which the bucket is complete, lo, hi) -----------------
    def schedule(self, plan):
        key = id(plan)
        if key in self._sched:
            return self._sched[key]
        eng = self.engine
        spans = []
        for p in plan.grad_params:
            o, n = eng.offsets[id(p)]
            spans.append((o, o + n, plan.last_write[id(p)]))
        spans.sort()
        buckets, cur_lo, cur_hi, cur_ready = [], None, None, -1
        for lo, hi, ready in spans:
            if cur_lo is None:
                cur_lo, cur_hi, cur_ready = lo, hi, ready
            elif hi - cur_lo > self.bucket_elems and cur_hi > cur_lo:
                buckets.append((cur_ready, cur_lo, cur_hi))
                cur_lo, cur_hi, cur_ready = lo, hi, ready
            else:
                cur_hi, cur_ready = hi, max(cur_ready, ready)
        if cur_lo is not None:
            buckets.append((cur_ready, cur_lo, cur_hi))
        buckets.sort()
        self._sched[key] = buckets
        return buckets

    def _allreduce(self, lo, hi):
        if _NO_COMM:        # probe: the cost of the bucket schedule itself (MI355_DP_NOCOMM=1), never set in production
            return
        dist.all_reduce(self.engine.flat_g[lo:hi], op=dist.ReduceOp.SUM, group=self.group)

    def _run_backward(self, plan, stream):
        plan.bind(stream)
        n = len(plan.bwd)
        buckets = self.schedule(plan)
        run = self.run_calls or (lambda first, last: plan.run_backward_range(stream, first, last))
        if not self.overlap or not self.engine.flat_g.is_cuda:
            start = 0
            for ready, lo, hi in buckets:         # same order as the overlapped path, executed serially
                run(start, ready + 1)
                start = ready + 1
                plan.join_side()
                self._allreduce(lo, hi)
            run(start, n)
            plan.join_side()
            return
        if self.comm_stream is None:
            self.comm_stream = torch.cuda.Stream()
        main = torch.cuda.current_stream()
        start = 0
        for ready, lo, hi in buckets:
            run(start, ready + 1)
            start = ready + 1
            ev = torch.cuda.Event()
            ev.record(main)
            ev_side = None
            if plan._side is not None:                      # the bucket's last writer may be a side-stream wgrad reduce
                ev_side = torch.cuda.Event()
                ev_side.record(plan._side)
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                if ev_side is not None:
                    self.comm_stream.wait_event(ev_side)
                self._allreduce(lo, hi)
        run(start, n)
        plan.join_side()
        main.wait_stream(self.comm_stream)

    def __call__(self, x):
        return self.net(x)
