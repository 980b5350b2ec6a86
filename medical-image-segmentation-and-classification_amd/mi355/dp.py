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

import ctypes
import os

import torch
import torch.distributed as dist

from . import graph
from .lib import lib

_NO_COMM = os.environ.get("MI355_DP_NOCOMM") == "1"
# MI355_DP_NATIVE=1: the bucket all-reduces go through the library's own RCCL entry points (mi355_comm_init /
# mi355_allreduce_bucket, csrc/comm.cpp) instead of torch.distributed.all_reduce — same library underneath, no ProcessGroup
# bookkeeping per call.  torch.distributed stays the default: it is the path the multi-GPU bench has been rehearsed with.
_NATIVE = os.environ.get("MI355_DP_NATIVE") == "1"
# MI355_DP_BUCKET_DTYPE=bf16 | fp16: the buckets travel rounded to two bytes (SURVEY.md 8e's "perf" mode: half the xGMI bytes per
# step; each rank's contribution is rounded once, the ring's partial sums at every hop).  Default fp32 = parity mode: what the
# data-parallel parity statement and the tests hold.
_WIRE = {"": None, "fp32": None, "bf16": torch.bfloat16, "fp16": torch.float16}[os.environ.get("MI355_DP_BUCKET_DTYPE", "")]


class DataParallel:
    def __init__(self, net, bucket_mb: float = 32.0, overlap: bool = True, process_group=None, force: bool = False,
                 bucket_dtype=None):
        self.net = net
        self.wire_dtype = bucket_dtype if bucket_dtype not in (None, torch.float32) else (_WIRE if bucket_dtype is None else None)
        self._wire = None                     # 2-byte staging buffer, one element per gradient (allocated on first use)
        self.engine = net.engine
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.bucket_elems = int(bucket_mb * (1 << 20) / 4)
        self.overlap = overlap
        self._sched = {}
        self.comm_stream = None
        self.inv_scale = 1.0 / self.world
        self.run_calls = None                 # injectable (CPU tests replace the kernel launcher)
        self._events = {}                     # plan -> per-bucket (main, side) event pairs, created once
        self.native = False
        if self.world > 1 or force:
            self.engine.bwd_runner = self._run_backward
            self._check_hw_queues()
            if _NATIVE and torch.cuda.is_available():
                self._init_native()
        if self.world > 1:
            self.sync_state()

    @staticmethod
    def _check_hw_queues():
        """The step runs three streams (main, weight-gradient side stream, comm stream) next to RCCL's own; on the runtime's
        default of four hardware queues two of them share one and serialise (measured +0.7 ms on a 19 ms step).
        mi355/__init__.py asks for eight — which only takes effect if the runtime had not initialised yet."""
        import mi355
        if torch.cuda.is_available() and not mi355.HW_QUEUES_IN_TIME:
            msg = ("mi355.dp: the HIP runtime was initialised before `mi355` was imported, so GPU_MAX_HW_QUEUES=8 could not be "
                   "applied: the main, weight-gradient and all-reduce streams may share a hardware queue (+0.7 ms on a 19 ms step); "
                   "import the package, or export GPU_MAX_HW_QUEUES=8, before the first CUDA call (INTEGRATION.md)")
            if os.environ.get("MI355_DP_STRICT_QUEUES") == "1":
                raise RuntimeError(msg)
            import warnings
            warnings.warn(msg, RuntimeWarning, stacklevel=3)

    def _init_native(self):
        """One RCCL communicator per process through the C ABI; rank 0's 128-byte id travels over the torch.distributed group
        (any backend) that is already up."""
        idt = torch.zeros(128, dtype=torch.uint8)
        rank = dist.get_rank(self.group) if dist.is_initialized() else 0
        if rank == 0:
            buf = (ctypes.c_char * 128)()
            if lib.mi355_comm_unique_id(ctypes.addressof(buf)) != 0:
                raise RuntimeError("mi355_comm_unique_id failed: " + lib.raw("mi355_last_error")().decode())
            idt = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
        if self.world > 1:
            dev = self.engine.flat_g.device if dist.get_backend(self.group) == "nccl" else torch.device("cpu")
            idt = idt.to(dev)
            dist.broadcast(idt, src=0, group=self.group)
            idt = idt.cpu()
        raw = bytes(idt.tolist())
        if lib.mi355_comm_world() == 0:
            if lib.mi355_comm_init(rank, self.world, ctypes.c_char_p(raw)) != 0:
                raise RuntimeError("mi355_comm_init failed: " + lib.raw("mi355_last_error")().decode())
        self.native = True

    @torch.no_grad()
    def sync_state(self, src: int = 0):
        """Replicas start from rank ``src``'s parameters and buffers (what DistributedDataParallel does at construction):
        one broadcast of the flat parameter buffer, one per buffer dtype (BatchNorm running_mean / running_var /
        num_batches_tracked).  Without it a rank-dependent initialisation, or a checkpoint loaded on one rank only, gives
        replicas that sum gradients but apply them to different weights."""
        self.engine._check_storage()
        dist.broadcast(self.engine.flat_p, src=src, group=self.group)
        self.engine.invalidate_packs()         # (a broadcast does not bump version counters: frozen-weight packs would go stale)
        self.sync_buffers(src)

    @torch.no_grad()
    def sync_buffers(self, src: int = 0):
        """Rank ``src``'s buffers (BatchNorm running statistics) on every rank: what DistributedDataParallel's per-forward buffer
        broadcast leaves in the model.  helpers.train() calls it once per epoch, in front of the validation pass, so every rank
        validates — and rank 0 checkpoints — the same model."""
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
        if self.wire_dtype is not None:
            return self._allreduce_on_wire(lo, hi)
        if self.native:     # enqueued on the CURRENT stream (the comm stream in the overlapped path)
            lib.mi355_allreduce_bucket(self.engine.flat_g[lo:hi], hi - lo, 0, torch.cuda.current_stream().cuda_stream)
            return
        dist.all_reduce(self.engine.flat_g[lo:hi], op=dist.ReduceOp.SUM, group=self.group)

    def _allreduce_on_wire(self, lo, hi):
        """The bucket rounded to ``wire_dtype`` in a staging buffer, summed there, widened back over the local gradients — three
        launches on the CURRENT stream (the comm stream in the overlapped path), in stream order with the collective."""
        g = self.engine.flat_g
        if self._wire is None or self._wire.numel() != g.numel() or self._wire.device != g.device:
            self._wire = torch.empty(g.numel(), dtype=self.wire_dtype, device=g.device)
        w = self._wire[lo:hi]
        code = {torch.bfloat16: 1, torch.float16: 2}[self.wire_dtype]
        if g.is_cuda:
            s = torch.cuda.current_stream().cuda_stream
            lib.mi355_grads_to_wire(g[lo:hi], w, hi - lo, code, s)
        else:               # host rehearsal of the schedule (gloo, no kernels anywhere in it): the same rounding by torch
            w.copy_(g[lo:hi])
        if self.native:
            lib.mi355_allreduce_bucket(w, hi - lo, code, torch.cuda.current_stream().cuda_stream)
        else:
            dist.all_reduce(w, op=dist.ReduceOp.SUM, group=self.group)
        if g.is_cuda:
            lib.mi355_grads_from_wire(w, g[lo:hi], hi - lo, code, torch.cuda.current_stream().cuda_stream)
        else:
            g[lo:hi].copy_(w)

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
        events = self._events.get(id(plan))
        if events is None:                                  # two events per bucket, created once per plan and re-recorded every step
            events = self._events[id(plan)] = [(torch.cuda.Event(), torch.cuda.Event()) for _ in buckets]
        start = 0
        for (ready, lo, hi), (ev, ev_s) in zip(buckets, events):
            run(start, ready + 1)
            start = ready + 1
            ev.record(main)
            ev_side = None
            if plan._side is not None:                      # the bucket's last writer may be a side-stream wgrad reduce
                ev_side = ev_s
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
