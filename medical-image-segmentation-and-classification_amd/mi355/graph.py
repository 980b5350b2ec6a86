"""Static launch plans for the MI355X conv-net path.

A model's forward is described ONCE per (input shape, mode) as a sequence of C-ABI launches
over preallocated HBM buffers (`Builder`); the matching backward sequence is generated at the
same time by reverse-mode rules attached to each op.  Executing a step is then the replay of two
tables of pre-resolved launches by csrc/plan.cpp (one mi355_plan_run call per forward / backward;
MI355_PLAN_C=0: a flat Python loop of ctypes calls) — no tracing, no per-op autograd nodes, no
allocation, and the whole step is hipGraph-capturable.

Data layout in HBM: activations are NHWC rows of the compute dtype (bf16 or
fp32); a concatenation is one wide buffer whose producers write their channel slice directly
(`new_cat`), so torch.cat (AttentionUNet.py:101,106,111,116) costs nothing; statistics,
parameters, packed-weight masters and parameter gradients are fp32.
"""
from __future__ import annotations

import ctypes
import os
import struct
from typing import List, Optional

import torch
import torch.nn as nn

from .lib import lib, DTYPE_CODE

BN_EPS_DEFAULT = 1e-5
CPAD = 32          # network inputs are zero-padded to 32 channels (MFMA K granularity)
FUSE_POOL = os.environ.get("MI355_FUSE_POOL", "1") != "0"      # MaxPool2d(2, 2) inside the BatchNorm apply pass that feeds it (A/B switch)
FUSE_POOL_BWD = os.environ.get("MI355_FUSE_POOL_BWD", "1") != "0"   # ... and its gradient inside that layer's two BatchNorm backward passes
FUSE_GATE_BWD = os.environ.get("MI355_FUSE_GATE_BWD", "1") != "0"   # attention gate: both branches' BatchNorm backward in two passes, d(psi_in) never stored
FUSE_HEAD = os.environ.get("MI355_FUSE_HEAD", "1") != "0"           # relu(bn(.)) in front of the one-channel logit convolution: never stored, both directions
STEM_IM2COL = os.environ.get("MI355_STEM_IM2COL", "1") != "0"       # Conv2d(3, Co, 3, 1, 1) on the network input as a pointwise convolution over its 3 x 3 patches
SIDE_COLSUM = os.environ.get("MI355_SIDE_COLSUM", "1") != "0"       # psi / head weight-gradient folds (they only feed the optimiser) leave the main stream
BN_ACT_WINDOWS = os.environ.get("MI355_BN_ACT_WINDOWS", "1") != "0"  # plain BatchNorm apply passes on even images run the window-ordered kernel (mi355_bn_act_pool2 without a pooled output)
STATIC_PACKS = os.environ.get("MI355_STATIC_PACKS", "1") != "0"     # weight packs of FROZEN parameters are refreshed when the parameters change, not every step
BN_ACT_WINDOWS_RES = os.environ.get("MI355_BN_ACT_WINDOWS_RES", "1") != "0"   # ... and the passes with a residual operand
FUSE_RESIDUAL = os.environ.get("MI355_FUSE_RESIDUAL", "1") != "0"   # RRCNN_block's x0 + RCNN(x0) inside the last BatchNorm apply pass (A/B switch)
DEFER_POST = os.environ.get("MI355_DEFER_POST", "1") != "0"         # recurrent block: d x = the SUM of its applications' incoming gradients, formed in one pass by the last of them (A/B switch)


class T:
    """NHWC activation handle: rows of C channels with channel stride ld inside `buf`."""
    __slots__ = ("buf", "off", "N", "H", "W", "C", "ld", "_ng", "_grad", "parent", "_written", "name", "_plain_bn_relu", "_lazy_pool", "_bn_src", "_lazy_head",
                 "_post_uses", "_post_seen", "_post_pending")

    def __init__(self, buf, off, N, H, W, C, ld, parent=None):
        self.buf, self.off = buf, off
        self.N, self.H, self.W, self.C, self.ld = N, H, W, C, ld
        self._ng = False
        self._grad = None
        self.parent = parent
        self._written = False
        self.name = ""
        self._plain_bn_relu = False     # relu(bn(conv(.))) with nothing added: produced by Builder.conv_bn_act in training
        self._lazy_pool = None          # gradient of a MaxPool2d(2, 2) of this tensor left to the producer's BatchNorm backward
        self._bn_src = None             # (raw convolution output, BatchNorm coefficient buffers) this activation was computed from
        self._lazy_head = None          # (dz, conv): the one-channel convolution whose backward the producer's BatchNorm passes compute
        self._post_uses = 0             # conv_bn_act(..., post_add=this) applications in the forward plan / met so far in backward order
        self._post_seen = 0
        self._post_pending = []         # their incoming gradients not yet added into this tensor's gradient (Builder._bn_bwd)

    @property
    def needs_grad(self):
        return self._ng

    @needs_grad.setter
    def needs_grad(self, v):
        self._ng = bool(v)
        if v and self.parent is not None:       # a concat needs a gradient as soon as one slice does
            self.parent._ng = True

    @property
    def M(self):
        return self.N * self.H * self.W

    @property
    def ptr(self):
        return self.buf.data_ptr() + self.off * self.buf.element_size()

    def torch_view(self):
        """[N,H,W,C] strided torch view (tests / debugging only)."""
        flat = self.buf[self.off:]
        return flat.as_strided((self.N, self.H, self.W, self.C), (self.H * self.W * self.ld, self.W * self.ld, self.ld, 1))


class V:
    """Small fp32 matrix [B, F] (classifier heads)."""
    __slots__ = ("buf", "B", "F", "needs_grad", "_grad", "_written")

    def __init__(self, buf, B, F):
        self.buf, self.B, self.F = buf, B, F
        self.needs_grad = False
        self._grad = None
        self._written = False

    @property
    def ptr(self):
        return self.buf.data_ptr()


class GRef:
    """Location of a parameter's gradient inside the engine's flat fp32 gradient buffer."""
    __slots__ = ("tensor", "off", "param")

    def __init__(self, tensor, off, param):
        self.tensor, self.off, self.param = tensor, off, param


class Ws:
    """Late-bound shared workspace pointer (sized to the largest request of the plan)."""
    __slots__ = ("kind",)

    def __init__(self, kind):
        self.kind = kind


class Launch:
    """One C-ABI call.  `flops` is the ALGORITHMIC work of the launch (2*MACs of the convolution it
    implements, real channel counts) when it is a GEMM-shaped kernel, else 0; `tag` names the kernel
    variant the launcher dispatches to (for per-kernel roofline accounting in bench.py)."""
    __slots__ = ("name", "args", "flops", "tag", "bytes", "side", "cus")

    def __init__(self, name, *args, flops=0, tag="", nbytes=0, side=False, cus=1.0):
        self.name, self.args, self.flops, self.tag, self.bytes = name, args, flops, tag, nbytes
        self.side = side          # True: may run on the plan's side stream (weight-gradient launches)
        self.cus = cus            # share of the chip's CUs the launch is sized for (1.0 unless its workgroups own CUs by design:
                                  # the eight-wave weight gradient runs on 128 of 256; bench.py's chip-time accounting)


class Plan:
    """Executable product of a Builder."""

    def __init__(self, b: "Builder"):
        self.device = b.device
        self.dtype = b.dtype
        self.training = b.training
        self.pre, self.fwd, self.bwd = b.pre, b.fwd, b.bwd
        self.static_pack, self.static_params, self._static_sig = b.static_pack, b.static_params, None
        self.flat_p = b.engine.flat_p
        self.engine = b.engine            # (its pack_epoch is part of the frozen packs' signature)
        self.keep = b.keep
        self.acts = b.acts                # [(kind, handles...)] in forward order: activations at the network's kinks (tests / diagnostics)
        self.input = b.input
        self.output = b.output            # ("z", tensor[M], N,H,W) or ("v", V)
        self.dout = b.dout                # fp32 buffer the loss writes dL/dlogits into
        self.input_grad = getattr(b, "input_grad", None)   # NCHW fp32 dL/dx when requested
        self.ws = {k: (torch.empty(max(n, 16), dtype=torch.uint8 if k == "bytes" else torch.float32, device=b.device))
                   for k, n in b.ws_need.items()}
        self.param_ptrs = [(p, p.data_ptr()) for p in b.params_seen]
        self.grad_params = list(b.grad_params)   # parameters that receive a gradient, in write order
        self._bound = {}
        self._cplans = {}                 # stream -> (forward plan, backward plan, side stream) handles of csrc/plan.cpp
        self._chandles = []
        self.replay_in_c = os.environ.get("MI355_PLAN_C", "1") != "0"
        self._side = None
        self.side_stream_enabled = os.environ.get("MI355_SIDE_STREAM", "1") != "0"
        self.n_launches = (len(self.pre) + len(self.fwd), len(self.bwd))
        # index of the last backward launch that writes each parameter's gradient (data-parallel
        # buckets become ready right after it)
        self.last_write = {}
        for i, l in enumerate(self.bwd):
            for a in l.args:
                if isinstance(a, GRef):
                    self.last_write[id(a.param)] = i
        self.zero_grad_params = list(b.zero_grad_params)   # biases in front of a train-mode BN: gradient == 0, never launched
        for p in self.zero_grad_params:
            self.last_write.setdefault(id(p), -1)

    # -- binding: resolve pointers once per stream --------------------------------------------
    def _resolve(self, launches: List[Launch], stream):
        out = []
        for l in launches:
            fn = lib.raw(l.name)
            conv = []
            for a in l.args:
                if isinstance(a, (T, V)):
                    conv.append(a.ptr)
                elif isinstance(a, torch.Tensor):
                    conv.append(a.data_ptr())
                elif isinstance(a, Ws):
                    conv.append(self.ws[a.kind].data_ptr())
                elif isinstance(a, GRef):
                    conv.append(a.tensor.data_ptr() + a.off)
                elif isinstance(a, tuple):          # (tensor, byte offset)
                    conv.append(a[0].data_ptr() + a[1])
                else:
                    conv.append(a)
            conv.append(stream)
            if len(conv) != len(fn.argtypes):
                raise TypeError(f"{l.name}: built {len(conv)} args, ABI takes {len(fn.argtypes)}")
            out.append((fn, tuple(conv), l.name, l))
        return out

    def bind(self, stream):
        """Resolve pointers for `stream`.  Backward launches flagged `side` (weight gradients: they only feed
        the optimiser) are bound to a private side stream so that the MFMA-bound wgrad kernels overlap the
        HBM-bound BatchNorm / pooling kernels and the tails of the dgrad kernels of the main chain."""
        key = int(stream or 0)
        if key not in self._bound:
            side = None
            if self.side_stream_enabled and self.device.type == "cuda" and any(l.side for l in self.bwd):
                if self._side is None:
                    self._side = torch.cuda.Stream(device=self.device)
                side = self._side.cuda_stream
            fwd = self._resolve(self.pre + self.fwd, stream)
            bwd = []
            for l in self.bwd:
                bwd.extend(self._resolve([l], side if (l.side and side is not None) else stream))
            self._bound[key] = (fwd, bwd)
            if self.replay_in_c:
                self._cplans[key] = (self._to_c(fwd, False), self._to_c(bwd, side is not None), side)
        return self._bound[key]

    # -- replay in C: the resolved tables are handed to csrc/plan.cpp once; a step is then 2-3 calls (mi355_plan_run) ----------
    def _to_c(self, resolved, with_side):
        create, pset = lib.raw("mi355_plan_create"), lib.raw("mi355_plan_set")
        cp = create(len(resolved))
        if not cp:
            raise RuntimeError("mi355_plan_create failed")
        self._chandles.append(cp)
        for i, (fn, args, name, l) in enumerate(resolved):
            types = [t for t, _ in lib.protos[name][1]]
            slots = (ctypes.c_uint64 * len(args))()
            for j, (v, t) in enumerate(zip(args, types)):
                if v is None:
                    slots[j] = 0
                elif t is ctypes.c_float:
                    slots[j] = struct.unpack("<I", struct.pack("<f", float(v)))[0]
                else:
                    slots[j] = int(v) & 0xFFFFFFFFFFFFFFFF
            rc = pset(cp, i, name.encode(), slots, len(args), 1 if (with_side and l.side) else 0)
            if rc:
                raise RuntimeError(f"mi355_plan_set({name}) failed (rc={rc}): {lib.raw('mi355_last_error')().decode()}")
        return cp

    def _c_fail(self, cp, rc, calls):
        i = lib.raw("mi355_plan_last_index")(cp)
        name = calls[i][2] if 0 <= i < len(calls) else "?"
        raise RuntimeError(f"{name} (launch {i}) failed (rc={rc}): {lib.raw('mi355_last_error')().decode()}")

    def __del__(self):
        try:
            destroy = lib.raw("mi355_plan_destroy")
            for cp in self._chandles:
                destroy(cp)
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass

    @staticmethod
    def _run(calls):
        for fn, args, name, _ in calls:
            rc = fn(*args)
            if rc:
                raise RuntimeError(f"{name} failed (rc={rc}): {lib.raw('mi355_last_error')().decode()}")

    def refresh_static_packs(self, stream):
        """Weight packs of frozen parameters: one launch when they changed, nothing otherwise.  load_state_dict / in-place edits
        bump the parameters' version counters; writers that version counters do not see — a collective into the flat buffer
        (DataParallel.sync_state: dist.broadcast leaves ``_version`` alone), ``p.data`` edits, raw-pointer writes — call
        ``Engine.invalidate_packs()``, whose epoch is the third part of the signature."""
        if self.static_pack is None:
            return
        sig = (self.flat_p._version, sum(p._version for p in self.static_params), self.engine.pack_epoch)
        if sig != self._static_sig:
            self._run(self._resolve([self.static_pack], stream))
            self._static_sig = sig

    def run_forward(self, stream, x_ptr=None):
        """pre + forward launches; ``x_ptr``: the caller's NCHW fp32 input (read in place by the first launch)."""
        self.refresh_static_packs(stream)
        calls = self.bind(stream)[0]
        key = int(stream or 0)
        if self.replay_in_c:
            cp = self._cplans[key][0]
            if x_ptr is not None:
                lib.raw("mi355_plan_patch")(cp, 0, 0, x_ptr)
            rc = lib.raw("mi355_plan_run")(cp, 0, len(calls), stream, None)
            if rc:
                self._c_fail(cp, rc, calls)
            return
        if x_ptr is not None:
            fn, args, name, _ = calls[0]
            rc = fn(x_ptr, *args[1:])
            if rc:
                raise RuntimeError(f"{name} failed (rc={rc}): {lib.raw('mi355_last_error')().decode()}")
            calls = calls[1:]
        self._run(calls)

    def run_calls_two_streams(self, calls):
        """Run a slice of the bound backward list, forking side-flagged groups onto the side stream (Python loop)."""
        if self._side is None:
            self._run(calls)
            return
        main = torch.cuda.current_stream()
        prev_side = False
        for fn, args, name, l in calls:
            if l.side and not prev_side:                 # fork: the side group may start once its inputs exist
                ev = torch.cuda.Event()
                ev.record(main)
                self._side.wait_event(ev)
            prev_side = l.side
            rc = fn(*args)
            if rc:
                raise RuntimeError(f"{name} failed (rc={rc}): {lib.raw('mi355_last_error')().decode()}")

    def run_backward_range(self, stream, first, last):
        """Backward launches [first, last) (the data-parallel runner interleaves all-reduces between ranges)."""
        calls = self.bind(stream)[1]
        if self.replay_in_c:
            _, cp, side = self._cplans[int(stream or 0)]
            rc = lib.raw("mi355_plan_run")(cp, first, last, stream, side)
            if rc:
                self._c_fail(cp, rc, calls)
        else:
            self.run_calls_two_streams(calls[first:last])

    def join_side(self):
        if self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)

    def run_backward(self, stream):
        self.run_backward_range(stream, 0, len(self.bwd))
        self.join_side()

    def params_moved(self):
        return any(p.data_ptr() != ptr for p, ptr in self.param_ptrs)


class _StemAsPointwise:
    """Conv2d(C <= 3, Co, 3, 1, 1) on the network input seen as Conv2d(9 C, Co, 1) on its im2col (mi355_pack_input_im2col3): the
    SAME parameters — [Co][C][3][3] is [Co][9 C][1][1] in memory, so the weight packs and the weight gradient land where they belong."""

    def __init__(self, conv):
        self.weight, self.bias = conv.weight, conv.bias
        self.in_channels, self.out_channels = conv.in_channels * 9, conv.out_channels
        self.kernel_size, self.stride, self.padding, self.dilation, self.groups = (1, 1), (1, 1), (0, 0), (1, 1), 1
        self.pack_shape = (conv.out_channels, conv.in_channels * 9, 1)


class Builder:
    """Emits forward launches and registers reverse-mode rules; `finish()` returns a Plan."""
    fuse_residual = FUSE_RESIDUAL

    def __init__(self, engine, device, dtype, training, want_grad):
        self.engine = engine
        self.device = torch.device(device)
        self.dtype = dtype
        self.code = DTYPE_CODE[dtype]
        self.esz = 4 if dtype == torch.float32 else 2
        self.epc = 16 // self.esz
        self.training = training
        self.want_grad = want_grad and training
        self.pre: List[Launch] = []
        self.fwd: List[Launch] = []
        self.bwd: List[Launch] = []
        self._rules = []                 # closures, run in reverse at finish()
        self._conv_uses = {}             # id(conv) -> forward applications (recurrent blocks share a conv)
        self._pending_wgrad = {}         # id(conv) -> [(x, dy)] waiting for the other applications' backward
        self.multi_wgrad = os.environ.get("MI355_WGRAD_MULTI", "1") != "0"
        self.keep = []
        self.ws_need = {"bytes": 0, "f32": 0}
        self._packs = {}
        self._pack_table = []            # every weight pack of the plan goes into ONE batched launch
        self._grad_first = {}
        self.params_seen = []
        self.grad_params = []
        self.zero_grad_params = []
        self.input = None
        self.output = None
        self.dout = None
        self.bn_momentum = 0.1
        # ("relu", a) | ("relu_pre", y, scale, shift) | ("relu_pre2", g1, scale, shift, x1, scale, shift) | ("relu_v", y) | ("pool", x, y, k, s, p) | ("gmax", x, argmax)
        self.acts = []

    # ---- allocation ---------------------------------------------------------------------------
    def _alloc(self, numel, dtype=None):
        t = torch.empty(max(int(numel), 8), dtype=dtype or self.dtype, device=self.device)
        self.keep.append(t)
        return t

    def f32(self, n, fill=None):
        t = self._alloc(n, torch.float32)
        if fill is not None:
            t.fill_(fill)
        return t

    def new_tensor(self, N, H, W, C):
        return T(self._alloc(N * H * W * C), 0, N, H, W, C, C)

    def new_cat(self, N, H, W, parts):
        """One wide buffer + channel-slice views (slice i covers parts[i] channels)."""
        tot = sum(parts)
        full = self.new_tensor(N, H, W, tot)
        slices, o = [], 0
        for c in parts:
            slices.append(T(full.buf, o, N, H, W, c, tot, parent=full))
            o += c
        return full, slices

    def ws_bytes(self, n):
        self.ws_need["bytes"] = max(self.ws_need["bytes"], int(n))
        return Ws("bytes")

    def ws_f32(self, n):
        self.ws_need["f32"] = max(self.ws_need["f32"], int(n))
        return Ws("f32")

    # ---- gradient bookkeeping ---------------------------------------------------------------------
    def grad_of(self, t):
        """Gradient handle of an activation (allocated on first use; slices share the parent's)."""
        if t._grad is None:
            if isinstance(t, V):
                t._grad = V(self.f32(t.B * t.F), t.B, t.F)
            elif t.parent is not None:
                pg = self.grad_of(t.parent)
                t._grad = T(pg.buf, t.off, t.N, t.H, t.W, t.C, pg.ld, parent=pg)
            else:
                t._grad = T(self._alloc(t.M * t.ld), 0, t.N, t.H, t.W, t.C, t.ld)
        return t._grad

    def acc_flag(self, t):
        """0 for the first gradient contribution to `t` in backward order, 1 afterwards."""
        g = self.grad_of(t)
        root = g.parent if (not isinstance(g, V) and g.parent is not None) else g
        was = g._written or root._written
        g._written = True
        if root is not g and g.C == root.C:
            root._written = True
        return 1 if was else 0

    def grad_written(self, t):
        """Has any backward launch emitted so far written (part of) the gradient of `t`?  (no side effect, unlike acc_flag)"""
        g = t._grad
        if g is None:
            # a slice of a wider buffer (concat input) whose consumer wrote the PARENT's gradient through grad_of(parent): the
            # slice has no handle of its own yet, but its gradient exists
            parent = None if isinstance(t, V) else t.parent
            return bool(parent is not None and parent._grad is not None and parent._grad._written)
        root = g.parent if (not isinstance(g, V) and g.parent is not None) else g
        return bool(g._written or root._written)

    def mark_full_written(self, t):
        g = self.grad_of(t)
        g._written = True

    def pgrad(self, p):
        """(flat grad tensor, byte offset) of parameter p and the beta (0 first write, 1 after)."""
        ref = self.engine.grad_ref(p)
        first = id(p) not in self._grad_first
        self._grad_first[id(p)] = True
        if first:
            self.grad_params.append(p)
        return ref, (0.0 if first else 1.0)

    def see(self, *params):
        for p in params:
            if p is not None:
                self.params_seen.append(p)

    def rule(self, fn):
        if self.want_grad:
            self._rules.append(fn)

    # ---- input / output ----------------------------------------------------------------------------
    def set_input(self, x_shape):
        N, C, H, W = x_shape
        self.input = (self.f32(N * C * H * W), (N, C, H, W))
        cpad = (C + CPAD - 1) // CPAD * CPAD
        xin = self.new_tensor(N, H, W, cpad)
        self.pre.append(Launch("mi355_pack_input_nchw", self.input[0], xin, N, C, H, W, cpad, self.code))
        self.input_grad = None
        self._xin, self._xin_pack, self._xin_users, self._xcol, self._stem = xin, self.pre[-1], 0, None, {}
        return xin

    def _stem_as_pointwise(self, x, conv, up):
        """(im2col tensor, pointwise view of conv) when conv is the 3 x 3 stem on the network input, else None."""
        if not (STEM_IM2COL and x is getattr(self, "_xin", None) and self.esz == 2 and not up and not x.needs_grad
                and tuple(conv.kernel_size) == (3, 3) and tuple(conv.stride) == (1, 1) and tuple(conv.padding) == (1, 1)
                and conv.in_channels == self.input[1][1] and conv.in_channels * 9 <= CPAD and conv.groups == 1):
            return None
        if self._xcol is None:
            N, C, H, W = self.input[1]
            self._xcol = self.new_tensor(N, H, W, CPAD)
            self.pre.append(Launch("mi355_pack_input_im2col3", self.input[0], self._xcol, N, C, H, W, self.code))
        if id(conv) not in self._stem:
            self._stem[id(conv)] = _StemAsPointwise(conv)
        return self._xcol, self._stem[id(conv)]

    def want_input_grad(self, xin):
        """Make the packed network input differentiable (block-level tests / composed pipelines): after
        backward, `plan.input_grad` holds dL/dx as NCHW fp32."""
        xin.needs_grad = True
        N, C, H, W = self.input[1]
        self.input_grad = self.f32(N * C * H * W)

        def rule():
            g = self.grad_of(xin)
            self.bwd.append(Launch("mi355_unpack_output_nchw", g, self.input_grad, N, C, H, W, g.ld, self.code))
        if self.want_grad:
            self._rules.insert(0, rule)      # runs last in backward order

    def slice_channels(self, t, c0, c):
        """View of channels [c0, c0+c) of an activation (shares storage and gradient)."""
        root = t.parent if t.parent is not None else t
        s = T(t.buf, t.off + c0, t.N, t.H, t.W, c, t.ld, parent=root)
        s._ng = t.needs_grad
        return s

    def tensor_output(self, t):
        """Expose an activation as the model output: NCHW fp32 (multi-channel heads, block tests)."""
        N, C, H, W = t.N, t.C, t.H, t.W
        out = self.f32(N * C * H * W)
        self.fwd.append(Launch("mi355_unpack_output_nchw", t, out, N, C, H, W, t.ld, self.code))
        self.output = ("z", out, (N, C, H, W))
        if self.want_grad and t.needs_grad:
            self.dout = self.f32(N * C * H * W)

            def rule():
                if self.acc_flag(t):
                    raise NotImplementedError("tensor_output of an activation with other consumers")
                g = self.grad_of(t)
                self.bwd.append(Launch("mi355_pack_nchw", self.dout, g, N, C, H, W, g.ld, self.code))
            self.rule(rule)
        return out

    # ---- packed weights -------------------------------------------------------------------------------
    def packs(self, conv, cip, transposed=False, scale=None):
        """(Wf, Wb) packs of a conv parameter; ``scale`` (per output channel, device fp32) is folded into them."""
        key = (id(conv), cip, id(scale))
        if key not in self._packs:
            w = conv.weight
            if transposed:
                ci, co = w.shape[0], w.shape[1]
            else:
                co, ci = w.shape[0], w.shape[1]
            k = w.shape[2]
            if hasattr(conv, "pack_shape"):
                co, ci, k = conv.pack_shape
            wf = self._alloc(co * k * k * cip)
            wb = self._alloc(co * k * k * cip) if self.want_grad else None
            assert k * k <= 288, "batched weight pack: at most 288 taps"
            self._pack_table.append((w, wf, wb, co, ci, cip, k * k, 1 if transposed else 0, scale))
            self.see(w, conv.bias)
            self._packs[key] = (wf, wb)
        return self._packs[key]

    # ---- convolution -----------------------------------------------------------------------------------
    def _conv_geom(self, x, conv, up):
        k, s, p = conv.kernel_size[0], conv.stride[0], conv.padding[0]
        if conv.kernel_size[0] != conv.kernel_size[1] or conv.dilation[0] != 1 or conv.groups != 1:
            raise NotImplementedError("square, undilated, ungrouped convolutions only")
        hl, wl = (2 * x.H, 2 * x.W) if up else (x.H, x.W)
        return k, s, p, (hl + 2 * p - k) // s + 1, (wl + 2 * p - k) // s + 1

    def conv_raw(self, x, conv, up=False, out=None, stats=False, relu=False, fold=None):
        """y = conv(x) (+bias), raw output in the compute dtype; returns (y, bwd(dy, bias_done)).  With
        ``stats`` the BatchNorm partial sums of y are produced by the conv epilogue when the kernel supports
        it (``self._last_stat_rows`` > 0 afterwards).  ``relu``: max(0, .) in the epilogue.  ``fold = (scale, bias)``:
        eval-mode BatchNorm folded into the packed weights and the bias (forward-only plans)."""
        stem = self._stem_as_pointwise(x, conv, up)
        if stem is not None:
            return self.conv_raw(stem[0], stem[1], out=out, stats=stats, relu=relu, fold=fold)
        if x is getattr(self, "_xin", None):
            self._xin_users += 1
        k, s, p, Ho, Wo = self._conv_geom(x, conv, up)
        Co = conv.out_channels
        assert x.C >= conv.in_channels and (x.C == conv.in_channels or conv.in_channels < CPAD), (x.C, conv.in_channels)
        wf, wb = self.packs(conv, x.C, scale=fold[0] if fold else None)
        bias = fold[1] if fold else conv.bias
        y = out if out is not None else self.new_tensor(x.N, Ho, Wo, Co)
        assert (y.N, y.H, y.W, y.C) == (x.N, Ho, Wo, Co)
        flops = 2 * x.N * Ho * Wo * Co * k * k * conv.in_channels
        # algorithmic HBM bytes of one conv launch: input read once, output written once, weights read once
        nbytes = (x.N * x.H * x.W * x.C + x.N * Ho * Wo * Co + Co * k * k * x.C) * self.esz
        stat_part = None
        self._last_stat_rows = 0
        if stats and self.training:
            rows = lib.mi355_conv2d_igemm_stat_rows(x.N, x.H, x.W, x.C, Ho, Wo, Co, k, k, s, 1, -p, 1, 1 if up else 0, self.code)
            if rows > 0:            # the kernel serving this shape folds the BatchNorm statistics into its epilogue
                stat_part = self.ws_f32(rows * 2 * Co)
                self._last_stat_rows = rows
        self.fwd.append(Launch("mi355_conv2d_igemm", x, wf, bias, y, x.N, x.H, x.W, x.C, x.ld, Ho, Wo, Co, y.ld,
                               k, k, s, 1, -p, 1, 1 if up else 0, 2 if relu else 0, stat_part, self.code, flops=flops, nbytes=nbytes,
                               tag=self.igemm_tag(x.N, x.H, x.W, x.C, Ho, Wo, Co, k, s, 1, -p, 1, 1 if up else 0)))
        y.needs_grad = x.needs_grad or conv.weight.requires_grad
        self._conv_uses[id(conv)] = self._conv_uses.get(id(conv), 0) + 1

        def bwd(dy, bias_done=False):
            uses = self._conv_uses[id(conv)]
            if (conv.weight.requires_grad and 1 < uses <= 6 and self.multi_wgrad and k == 3 and s == 1 and p == 1 and self.esz == 2
                    and lib.mi355_conv2d_wgrad_multi_ok(x.N, Ho, Wo, self.code)):
                # a convolution applied several times (recurrent block): ONE weight-gradient launch over all (x, dy) pairs, emitted
                # with the last application's backward (the first in forward order), ONE set of partial slabs, ONE reduce
                pend = self._pending_wgrad.setdefault(id(conv), [])
                pend.append((x, dy))
                assert (x.N, x.H, x.W, x.C, x.ld, dy.ld) == tuple(getattr(pend[0][0], f) for f in ("N", "H", "W", "C", "ld")) + (pend[0][1].ld,)
                if len(pend) == uses:
                    self._emit_multi_wgrad(conv, pend, Ho, Wo, Co, k, up, flops, nbytes)
                    del self._pending_wgrad[id(conv)]
            elif conv.weight.requires_grad:
                splits = lib.mi355_conv2d_wgrad_splits(x.N, Ho, Wo, x.C, Co, k, k)
                ws = self.ws_bytes(splits * Co * k * k * x.C * 4)
                self.bwd.append(Launch("mi355_conv2d_wgrad", x, dy, ws, splits, x.N, x.H, x.W, x.C, x.ld, Ho, Wo, Co, dy.ld,
                                       k, k, s, p, 1 if up else 0, self.code, flops=flops, nbytes=nbytes, side=True,
                                       tag=self.wgrad_tag(Co, x.C, k, s, Ho, Wo, x.N, p), cus=self.wgrad_cus(x.N, Ho, Wo, x.C, Co, k, s, p, splits)))
                ref, beta = self.pgrad(conv.weight)
                self.bwd.append(Launch("mi355_conv2d_wgrad_reduce", ws, splits, ref, Co, x.C, conv.in_channels, k, k, 0, beta, side=True))
            if conv.bias is not None and conv.bias.requires_grad and not bias_done:
                self.bias_grad_from(dy, conv.bias)
            if x.needs_grad:
                if up:
                    acc = self.acc_flag(x)
                    xg = self.grad_of(x)
                    tag = self.igemm_tag(x.N, Ho, Wo, Co, 2 * x.H, 2 * x.W, x.C, k, 1, -1, p, s, 0)
                    # (the launcher's own choice for THIS batch: the shape-level variant may fall back, by batch size, to the LDS-DMA ring
                    #  kernel, which has no 2x2-sum epilogue)
                    if lib.mi355_conv2d_igemm_variant_n(x.N, Ho, Wo, Co, 2 * x.H, 2 * x.W, x.C, k, k, 1, -1, p, s, 0, self.code) in (2, 3, 5, 6, 7, 8):
                        # the data gradient lives on the up-sampled grid; its 2x2 sums go straight to the half-resolution
                        # gradient in the kernel epilogue (no full-resolution temporary, no separate pass)
                        self.bwd.append(Launch("mi355_conv2d_igemm", dy, wb, None, xg, x.N, Ho, Wo, Co, dy.ld, 2 * x.H, 2 * x.W,
                                               x.C, xg.ld, k, k, 1, -1, p, s, 0, 4 | (1 if acc else 0), None, self.code, flops=flops,
                                               nbytes=nbytes, tag=tag))
                    else:
                        tmp = self.new_tensor(x.N, 2 * x.H, 2 * x.W, x.C)
                        self.bwd.append(Launch("mi355_conv2d_igemm", dy, wb, None, tmp, x.N, Ho, Wo, Co, dy.ld, 2 * x.H, 2 * x.W,
                                               x.C, tmp.ld, k, k, 1, -1, p, s, 0, 0, None, self.code, flops=flops, nbytes=nbytes, tag=tag))
                        self.bwd.append(Launch("mi355_upsample2_bwd", tmp, tmp.ld, xg, xg.ld, x.N, x.H, x.W, x.C, acc, self.code))
                else:
                    acc = self.acc_flag(x)
                    xg = self.grad_of(x)
                    self.bwd.append(Launch("mi355_conv2d_igemm", dy, wb, None, xg, x.N, Ho, Wo, Co, dy.ld, x.H, x.W, x.C,
                                           xg.ld, k, k, 1, -1, p, s, 0, acc, None, self.code, flops=flops, nbytes=nbytes,
                                           tag=self.igemm_tag(x.N, Ho, Wo, Co, x.H, x.W, x.C, k, 1, -1, p, s, 0)))
        return y, bwd

    def _emit_multi_wgrad(self, conv, pend, Ho, Wo, Co, k, up, flops, nbytes):
        x0 = pend[0][0]
        n = len(pend)
        splits = lib.mi355_conv2d_wgrad_splits(x0.N * n, Ho, Wo, x0.C, Co, k, k)
        ws = self.ws_bytes(splits * Co * k * k * x0.C * 4)
        ops = []
        for i in range(6):
            ops += list(pend[i]) if i < n else [None, None]
        self.bwd.append(Launch("mi355_conv2d_wgrad_multi", *ops, n, ws, splits, x0.N, x0.H, x0.W, x0.C, x0.ld, Ho, Wo, Co,
                               pend[0][1].ld, 1 if up else 0, self.code, flops=flops * n, nbytes=nbytes * n, side=True,
                               tag=self.wgrad_tag(Co, x0.C, k, 1, Ho, Wo, x0.N, 1), cus=self.wgrad_cus(x0.N, Ho, Wo, x0.C, Co, k, 1, 1, splits)))
        ref, beta = self.pgrad(conv.weight)
        self.bwd.append(Launch("mi355_conv2d_wgrad_reduce", ws, splits, ref, Co, x0.C, conv.in_channels, k, k, 0, beta, side=True))

    def igemm_tag(self, N, Hi, Wi, ci, Ho, Wo, co, k, mul, kmul, off, div, up):
        """Name of the kernel mi355_conv2d_igemm runs for this launch: the launcher's own choice, batch-dependent fall-backs
        included (mi355_conv2d_igemm_variant_n; csrc/conv_igemm.hip pick_variant / resolve_variant), so that bench.py's
        per-kernel time and FLOP sums never mix two kernels under one name."""
        bn = 128 if co % 128 == 0 else (64 if co % 64 == 0 else 32)
        if self.dtype == torch.float32:
            return f"conv_igemm_kernel<f32,{lib.mi355_conv2d_igemm_generic_tile(N, Ho, Wo, co)},16>"
        v = lib.mi355_conv2d_igemm_variant_n(N, Hi, Wi, ci, Ho, Wo, co, k, k, mul, kmul, off, div, up, self.code)
        k64 = ci % 64 == 0
        if v == 0:
            return f"conv_igemm_kernel<bf16,{lib.mi355_conv2d_igemm_generic_tile(N, Ho, Wo, co)},{64 if k64 else 32}>"
        if v == 1:
            bn = lib.mi355_conv2d_igemm_dma_tile(N, Ho, Wo, ci, co)      # (narrower than Co allows when the grid would be small)
            if bn == 128:
                return "conv_igemm_dma_kernel<128,64,2>" if k64 else "conv_igemm_dma_kernel<128,32,3>"
            return "conv_igemm_dma_kernel<64,32,3>" if bn == 64 else "conv_igemm_dma_kernel<32,64,3>"
        return {2: "conv3x3_halo_rw_kernel<8,32>", 3: "conv3x3_halo_rw_kernel<16,16>", 4: f"conv1x1_stream_kernel<{ci},{co}>",
                5: "conv3x3_halo_pp_kernel", 6: "conv3x3_halo_pp128_kernel", 7: "conv3x3_ws_kernel<64,8>", 8: "conv3x3_ws_kernel<128,4>",
                9: "conv_gemm256_kernel"}[v]

    def wgrad_tag(self, co, ci, k=1, s=1, Ho=0, Wo=0, N=0, p=None):
        """Name of the kernel mi355_conv2d_wgrad runs: the launcher's own choice (mi355_conv2d_wgrad_variant)."""
        t = "f32" if self.dtype == torch.float32 else "bf16"       # (the fp16 build runs the same variants as bf16)
        v = lib.mi355_conv2d_wgrad_variant(N, Ho, Wo, k, k, s, (k // 2) if p is None else p, self.code) if t == "bf16" else 0
        if v:
            return "wgrad3x3_halo8_kernel" if v >= 3 else "wgrad3x3_halo_kernel"
        return f"conv_wgrad_kernel<{t},{128 if co % 128 == 0 else 64},{128 if ci % 128 == 0 else 64}>"

    def wgrad_cus(self, N, Ho, Wo, ci, co, k, s, p, splits):
        """Share of the 256 CUs a weight-gradient launch is sized for: the eight-wave nine-tap kernel's workgroups own their CU
        (512 threads x 256 registers), and its grid — (Co / 64) x (Ci / 64) tiles x splits — is a deliberate share of the chip."""
        if self.esz != 2 or lib.mi355_conv2d_wgrad_variant(N, Ho, Wo, k, k, s, p, self.code) < 3:
            return 1.0
        return min(1.0, -(-co // 64) * -(-ci // 64) * splits / 256.0)

    def bias_grad_from(self, dy, bias):
        nb = lib.mi355_rowreduce_blocks(dy.M)
        part = self.ws_f32(nb * dy.C)
        self.bwd.append(Launch("mi355_colsum", dy, dy.ld, part, dy.M, dy.C, self.code))
        ref, beta = self.pgrad(bias)
        self.bwd.append(Launch("mi355_colsum_finalize", part, nb, 1, dy.C, ref, beta))

    # ---- batch norm state ---------------------------------------------------------------------------------
    def _bn_coeffs(self, y, bn, fused_rows=0):
        """Emit statistics (train) or running-stat coefficients (eval); returns dict of fp32 buffers.
        ``fused_rows`` > 0: the producing conv already left that many partial rows in the f32 workspace."""
        C = bn.num_features
        st = {k: self.f32(C) for k in ("scale", "shift", "mean", "invstd")}
        self.see(bn.weight, bn.bias)
        if self.training:
            if fused_rows > 0:
                nb = fused_rows
                part = self.ws_f32(nb * 2 * C)
                if nb > 2048:       # one row per conv tile (4 MB of partials at 256^2): pre-fold to 64 rows with a wide grid
                    folded = self.f32(64 * 2 * C)
                    self.fwd.append(Launch("mi355_fold_rows", part, nb, 2 * C, folded, 64))
                    part, nb = folded, 64
            else:
                nb = lib.mi355_rowreduce_blocks(y.M)
                part = self.ws_f32(nb * 2 * C)
                self.fwd.append(Launch("mi355_bn_stats", y, part, y.M, C, y.ld, self.code))
            mom = bn.momentum if bn.momentum is not None else 0.1
            track = bn.track_running_stats
            self.fwd.append(Launch("mi355_bn_finalize", part, nb, y.M, C, bn.weight, bn.bias,
                                   bn.running_mean if track else None, bn.running_var if track else None,
                                   bn.num_batches_tracked if track else None, float(mom), float(bn.eps),
                                   st["scale"], st["shift"], st["mean"], st["invstd"]))
        else:
            self.fwd.append(Launch("mi355_bn_eval_coeffs", bn.weight, bn.bias, bn.running_mean, bn.running_var, float(bn.eps),
                                   C, st["scale"], st["shift"]))
        return st

    def _bn_bwd(self, da, a, y, bn, st, act, dres_to=None, bias=None, post_to=None, pool_dp=None, head=None):
        """Emit BN(+ReLU) backward: returns dy (grad of the raw input y).  ``pool_dp``: the gradient of a MaxPool2d(2, 2) of the
        activation that maxpool() left for these passes to add on the fly (no mi355_maxpool_bwd pass over da)."""
        C = bn.num_features
        nb = lib.mi355_rowreduce_blocks(y.M)
        part = self.ws_f32(nb * 2 * C)
        if head is not None:
            # relu(bn(y)) feeds ONLY a one-channel 1x1 convolution (the logit head): its gradient dz[m] * w[c] * [a > 0] is recomputed
            # from dz and y by the attention gate's two-pass kernels with ONE normalised operand; the same passes leave the head's
            # weight / bias gradients (mi355_rowdot_bwd and the stored activation and its stored gradient all disappear)
            dz, hconv = head
            assert act and dres_to is None and post_to is None and pool_dp is None
            part = self.f32(nb * 5 * C) if SIDE_COLSUM else self.ws_f32(nb * 5 * C)      # (its own buffer: read from the side stream)
            co = (st["scale"], st["shift"], st["mean"], st["invstd"], None, None, None, None)
            self.bwd.append(Launch("mi355_gate_bn_bwd_reduce", dz, y, y.ld, None, 0, *co, hconv.weight, part, y.M, C, self.code,
                                   nbytes=y.M * C * self.esz + 4 * y.M))
            head_folds = []
            if hconv.weight.requires_grad:
                wref, wbeta = self.pgrad(hconv.weight)
                head_folds.append(Launch("mi355_colsum_finalize", self._ws_off(part, 3 * C * 4), nb, 5, C, wref, wbeta, side=SIDE_COLSUM))
                if hconv.bias is not None:
                    bref2, bbeta = self.pgrad(hconv.bias)
                    head_folds.append(Launch("mi355_colsum_finalize", self._ws_off(part, 4 * C * 4), nb, 5 * C, 1, bref2, bbeta, side=SIDE_COLSUM))
            sums = self.f32(2 * C)
            need_pg = bn.weight.requires_grad
            if need_pg:
                gref, gbeta = self.pgrad(bn.weight)
                bref, _ = self.pgrad(bn.bias)
            self.bwd.append(Launch("mi355_bn_bwd_finalize_at", part, min(nb, lib.mi355_gate_bn_bwd_reduce_rows(y.M)), 5, 0, 1, C, sums,
                                   gref if need_pg else None, bref if need_pg else None, gbeta if need_pg else 0.0))
            if bias is not None and bias.requires_grad and id(bias) not in self._grad_first:
                self.pgrad(bias)
                self.zero_grad_params.append(bias)
            dy = self.grad_of(y)
            self.bwd.append(Launch("mi355_gate_bn_bwd_apply", dz, y, y.ld, None, 0, *co, hconv.weight, bn.weight, None, sums, None,
                                   dy, dy.ld, None, 0, y.M, C, self.code, nbytes=2 * y.M * C * self.esz + 4 * y.M))
            self.bwd += head_folds        # (they only feed the optimiser: behind the apply pass, off the main stream)
            return dy
        if pool_dp is not None:
            assert act and dres_to is None and post_to is None
            dal = da.ld if da is not None else 0          # (da None: the pooled gradient is the activation's whole gradient)
            self.bwd.append(Launch("mi355_bn_bwd_reduce_pool2", da, dal, pool_dp, pool_dp.ld, y, y.ld, st["mean"], st["invstd"],
                                   st["scale"], st["shift"], part, y.N, y.H, y.W, C, self.code,
                                   nbytes=int((1.25 + (da is not None)) * y.M * C * self.esz)))
            sums = self.f32(2 * C)
            need_pg = bn.weight.requires_grad
            if need_pg:
                gref, gbeta = self.pgrad(bn.weight)
                bref, _ = self.pgrad(bn.bias)
            self.bwd.append(Launch("mi355_bn_bwd_finalize", part, min(nb, lib.mi355_bn_bwd_reduce_pool2_rows(y.M)), C, sums,
                                   gref if need_pg else None, bref if need_pg else None, gbeta if need_pg else 0.0))
            dy = self.grad_of(y)
            if bias is not None and bias.requires_grad and id(bias) not in self._grad_first:
                self.pgrad(bias)
                self.zero_grad_params.append(bias)
            self.bwd.append(Launch("mi355_bn_bwd_apply_pool2", da, dal, pool_dp, pool_dp.ld, y, y.ld, bn.weight, st["mean"],
                                   st["invstd"], st["scale"], st["shift"], sums, dy, dy.ld, y.N, y.H, y.W, C, self.code,
                                   nbytes=int((2.25 + (da is not None)) * y.M * C * self.esz)))
            return dy
        # the ReLU mask is recomputed from the raw input with the forward's coefficients unless something was
        # added in front of the ReLU (residual / second operand), in which case the activated tensor is read
        am = a if (act and dres_to is not None) else None
        self.bwd.append(Launch("mi355_bn_bwd_reduce", da, da.ld, am, am.ld if am is not None else 0, y, y.ld,
                               st["mean"], st["invstd"], st["scale"], st["shift"], part, y.M, C, 1 if act else 0, self.code,
                               nbytes=(2 + (am is not None)) * y.M * C * self.esz))
        sums = self.f32(2 * C)
        need_pg = bn.weight.requires_grad
        if need_pg:
            gref, gbeta = self.pgrad(bn.weight)
            bref, _ = self.pgrad(bn.bias)
        # (the reduction runs on at most 256 workgroups — rowred.hpp, ops that keep several rows in flight — and zero-fills the partial
        # rows beyond its grid: the fold only has to read the rows that can be non-zero)
        nb_fold = min(nb, lib.mi355_bn_bwd_reduce_rows(y.M))
        self.bwd.append(Launch("mi355_bn_bwd_finalize", part, nb_fold, C, sums, gref if need_pg else None,
                               bref if need_pg else None, gbeta if need_pg else 0.0))
        dy = self.grad_of(y)
        dres = None
        if dres_to is not None and dres_to.needs_grad:
            if self.acc_flag(dres_to):
                dres = self.new_tensor(y.N, y.H, y.W, C)      # accumulate through a temporary
            else:
                dres = self.grad_of(dres_to)
        # The gradient of a conv bias that feeds a train-mode BatchNorm is exactly zero (sum_m dy = 0 because
        # sum_m xhat = 0); torch computes ~1e-9 of round-off there.  It is not computed: the slot in the flat
        # gradient buffer stays at its initial zero, the parameter is still registered as "has a gradient".
        if bias is not None and bias.requires_grad and id(bias) not in self._grad_first:
            self.pgrad(bias)
            self.zero_grad_params.append(bias)
        # an operand added AFTER the activation (recurrent block x + relu(bn(.))) receives the incoming gradient itself: the
        # apply pass reads it anyway and writes / accumulates it (no separate mi355_add pass over da)
        pg, pacc = None, 0
        if post_to is not None and post_to.needs_grad:
            # Several applications add the SAME operand (the recurrent block's x, R2AttU_Net.py:41-44): their incoming gradients are
            # left pending and the last application's pass adds them all into d x at once (mi355_bn_bwd_apply_post4: up to four
            # earlier ones; one fp32 sum and one rounding instead of a read-modify-write of d x per application)
            post_to._post_seen += 1
            pend = post_to._post_pending
            last_one = post_to._post_seen >= post_to._post_uses
            if DEFER_POST and dres is None and not last_one and len(pend) < 4 and all(t.ld == da.ld for t in pend):
                pend.append(da)
            else:
                pacc = self.acc_flag(post_to)
                pg = self.grad_of(post_to)
                if pend:
                    ex = pend + [None] * (4 - len(pend))
                    self.bwd.append(Launch("mi355_bn_bwd_apply_post4", da, da.ld, am, am.ld if am is not None else 0, y, y.ld, bn.weight,
                                           st["mean"], st["invstd"], st["scale"], st["shift"], sums, dy, dy.ld, pg, pg.ld,
                                           1 if pacc else 0, *ex, pend[0].ld, y.M, C, 1 if act else 0, self.code,
                                           nbytes=(4 + (am is not None) + (1 if pacc else 0) + len(pend)) * y.M * C * self.esz))
                    post_to._post_pending = []
                    assert dres is None
                    return dy
        self.bwd.append(Launch("mi355_bn_bwd_apply", da, da.ld, am, am.ld if am is not None else 0, y, y.ld, bn.weight,
                               st["mean"], st["invstd"], st["scale"], st["shift"], sums, dy, dy.ld,
                               dres, dres.ld if dres is not None else 0, pg, pg.ld if pg is not None else 0, 1 if pacc else 0,
                               None, y.M, C, 1 if act else 0, self.code,
                               nbytes=(3 + (am is not None) + (dres is not None) + (pg is not None) * (2 if pacc else 1)) * y.M * C * self.esz))
        if dres is not None and dres is not dres_to._grad:
            rg = self.grad_of(dres_to)
            self.bwd.append(Launch("mi355_add", rg, rg.ld, dres, dres.ld, rg, rg.ld, y.M, C, self.code))
        return dy

    class _WsOff:
        __slots__ = ("ws", "off")

        def __init__(self, ws, off):
            self.ws, self.off = ws, off

    def _ws_off(self, ws, off):
        """`off` bytes into a shared workspace (bound when the plan is finished) or into a buffer of its own."""
        return Builder._WsOff(ws, off) if isinstance(ws, Ws) else (ws, off)

    # ---- fused block ops -------------------------------------------------------------------------------------
    def conv_bn_act(self, x, conv, bn, act=True, up=False, out=None, res=None, post_add=None):
        """act(bn(conv(x)) [+ res]) [+ post_add] — the workhorse (AttentionUNet.py:4-13,15-27; ResNet.py:36-44;
        ``post_add``: the recurrent block's x + relu(bn(conv(.))) (R2AttU_Net.py:44) produced in one pass)."""
        assert res is None or post_add is None
        if not self.training and not self.want_grad and res is None and post_add is None and bn.track_running_stats:
            # inference: BN(running statistics) folded into the packed weights and the bias, ReLU in the conv epilogue —
            # one launch, no normalisation pass (pipeline.py:324-357 runs the models in eval mode)
            sc, sh, fb = self.f32(bn.num_features), self.f32(bn.num_features), self.f32(bn.num_features)
            self.see(bn.weight, bn.bias)
            self.pre.append(Launch("mi355_bn_eval_coeffs", bn.weight, bn.bias, bn.running_mean, bn.running_var, float(bn.eps),
                                   bn.num_features, sc, sh))
            self.pre.append(Launch("mi355_bn_fold_bias", conv.bias, sc, sh, fb, bn.num_features))
            a, _ = self.conv_raw(x, conv, up, out=out, relu=act, fold=(sc, fb))
            a.needs_grad = False
            return a
        if post_add is not None:
            post_add._post_uses += 1
        y, conv_bwd = self.conv_raw(x, conv, up, stats=True)
        st = self._bn_coeffs(y, bn, self._last_stat_rows)
        a = out if out is not None else self.new_tensor(y.N, y.H, y.W, y.C)
        r = res if res is not None else post_add
        flags = (1 if act else 0) | (2 if post_add is not None else 0)
        self.fwd.append(Launch("mi355_bn_act", y, y.ld, st["scale"], st["shift"], None, 0, None, None,
                               r, r.ld if r is not None else 0, a, a.ld, y.M, y.C, flags, self.code,
                               nbytes=(2 + (r is not None)) * y.M * y.C * self.esz))
        a.needs_grad = y.needs_grad or bn.weight.requires_grad or (r is not None and r.needs_grad)
        if act:
            self.acts.append(("relu", a) if post_add is None else ("relu_pre", y, st["scale"], st["shift"]))
        a._plain_bn_relu = bool(act and r is None and self.training)      # (maxpool(): its gradient may ride in this layer's backward)
        a._bn_src = (y, st)

        def rule():
            if not a.needs_grad:
                return
            if a._lazy_head is not None:      # logit_conv(): the gradient of `a` is dz[m] * w[c], recomputed inside the two passes
                dy = self._bn_bwd(None, a, y, bn, st, act, bias=conv.bias, head=a._lazy_head)
            elif a._lazy_pool is not None and not self.grad_written(a):      # the pooling was the only consumer: its routed gradient alone
                dy = self._bn_bwd(None, a, y, bn, st, act, bias=conv.bias, pool_dp=a._lazy_pool)
            else:
                da = self.grad_of(a)
                # d(x + relu(.)) / dx = identity: folded into the BatchNorm apply pass
                dy = self._bn_bwd(da, a, y, bn, st, act, dres_to=res, bias=conv.bias, post_to=post_add, pool_dp=a._lazy_pool)
            conv_bwd(dy, bias_done=True)
        self.rule(rule)
        return a

    def bn_act(self, x, bn, act=False, out=None):
        """Stand-alone BatchNorm (+ReLU) on an activation (ResNet.py:134: bn1 applied a second time)."""
        st = self._bn_coeffs(x, bn)
        a = out if out is not None else self.new_tensor(x.N, x.H, x.W, x.C)
        self.fwd.append(Launch("mi355_bn_act", x, x.ld, st["scale"], st["shift"], None, 0, None, None, None, 0, a, a.ld,
                               x.M, x.C, 1 if act else 0, self.code))
        a.needs_grad = x.needs_grad or bn.weight.requires_grad

        def rule():
            if not a.needs_grad:
                return
            da = self.grad_of(a)
            if x.needs_grad and self.acc_flag(x):
                raise NotImplementedError("bn_act input with several consumers")
            self._bn_bwd(da, a, x, bn, st, act)
        self.rule(rule)
        return a

    def conv_act(self, x, conv, relu=False, up=False, out=None):
        """conv (+bias) with optional ReLU and no normalisation (VGG.py:9-41; R2AttU_Net.py:54)."""
        y, conv_bwd = self.conv_raw(x, conv, up, out=out, relu=relu)        # ReLU rides in the conv epilogue
        if not relu:
            def rule():
                if y.needs_grad:
                    conv_bwd(self.grad_of(y))
            self.rule(rule)
            return y
        a = y                                                                 # (relu'(.) is recovered from a > 0)
        self.acts.append(("relu", a))

        def rule():
            if not a.needs_grad:
                return
            da, dy = self.grad_of(a), self.grad_of(y)
            self.bwd.append(Launch("mi355_relu_bwd", da, da.ld, a, a.ld, dy, dy.ld, y.M, y.C, self.code))
            conv_bwd(dy)
        self.rule(rule)
        return a

    def conv_transpose(self, x, mod, out=None):
        """ConvTranspose2d(k, stride=k) (ResnetUnet.py:21,51) as the data-gradient form of the igemm."""
        k, s = mod.kernel_size[0], mod.stride[0]
        assert k == s and mod.padding[0] == 0 and mod.output_padding[0] == 0
        Ci, Co = mod.in_channels, mod.out_channels
        assert x.C == Ci
        wf, wb = self.packs(mod, Ci, transposed=True)
        Ho, Wo = x.H * s, x.W * s
        y = out if out is not None else self.new_tensor(x.N, Ho, Wo, Co)
        flops = 2 * x.N * x.H * x.W * Ci * Co * k * k
        nbytes = (x.N * x.H * x.W * Ci + x.N * Ho * Wo * Co + Ci * Co * k * k) * self.esz
        self.fwd.append(Launch("mi355_conv2d_igemm", x, wf, mod.bias, y, x.N, x.H, x.W, Ci, x.ld, Ho, Wo, Co, y.ld, k, k,
                               1, -1, 0, s, 0, 0, None, self.code, flops=flops, nbytes=nbytes,
                               tag=self.igemm_tag(x.N, x.H, x.W, Ci, Ho, Wo, Co, k, 1, -1, 0, s, 0)))
        y.needs_grad = x.needs_grad or mod.weight.requires_grad

        def rule():
            if not y.needs_grad:
                return
            dy = self.grad_of(y)
            if mod.weight.requires_grad:
                # roles swap: the big tensor dy is gathered with stride-s addressing, x is the "dy" operand
                splits = lib.mi355_conv2d_wgrad_splits(x.N, x.H, x.W, Co, Ci, k, k)
                ws = self.ws_bytes(splits * Ci * k * k * Co * 4)
                # side=True like every other user of the split-K slab workspace: all of them serialise on the side stream
                self.bwd.append(Launch("mi355_conv2d_wgrad", dy, x, ws, splits, x.N, Ho, Wo, Co, dy.ld, x.H, x.W, Ci, x.ld,
                                       k, k, s, 0, 0, self.code, side=True))
                ref, beta = self.pgrad(mod.weight)
                self.bwd.append(Launch("mi355_conv2d_wgrad_reduce", ws, splits, ref, Ci, Co, Co, k, k, 0, beta, side=True))
                if mod.bias is not None:
                    self.bias_grad_from(dy, mod.bias)
            if x.needs_grad:
                acc = self.acc_flag(x)
                xg = self.grad_of(x)
                self.bwd.append(Launch("mi355_conv2d_igemm", dy, wb, None, xg, x.N, Ho, Wo, Co, dy.ld, x.H, x.W, Ci, xg.ld,
                                       k, k, s, 1, 0, 1, 0, acc, None, self.code, flops=flops, nbytes=nbytes,
                                       tag=self.igemm_tag(x.N, Ho, Wo, Co, x.H, x.W, Ci, k, s, 1, 0, 1, 0)))
        self.rule(rule)
        return y

    # ---- pooling / add -----------------------------------------------------------------------------------------
    def maxpool(self, x, k=2, s=2, p=0):
        Ho, Wo = (x.H + 2 * p - k) // s + 1, (x.W + 2 * p - k) // s + 1
        y = self.new_tensor(x.N, Ho, Wo, x.C)
        last = self.fwd[-1] if self.fwd else None
        if (FUSE_POOL and (k, s, p) == (2, 2, 0) and x.H % 2 == 0 and x.W % 2 == 0 and last is not None and last.name == "mi355_bn_act"
                and last.args[10] is x and last.args[4] is None and last.args[8] is None and last.args[12] == x.M):
            # the pooled tensor leaves the BatchNorm apply pass that has just produced x (a plain one: no second operand, no
            # residual): one read of the raw convolution output instead of that plus a re-read of the activation
            a = last.args
            self.fwd[-1] = Launch("mi355_bn_act_pool2", a[0], a[1], a[2], a[3], x, x.ld, y, y.ld, x.N, x.H, x.W, x.C, a[14], self.code,
                                  nbytes=last.bytes + y.M * y.C * self.esz)
        else:
            self.fwd.append(Launch("mi355_maxpool_fwd", x, x.ld, y, y.ld, x.N, x.H, x.W, x.C, k, s, p, self.code))
        self.acts.append(("pool", x, y, k, s, p))
        y.needs_grad = x.needs_grad

        def rule():
            if not y.needs_grad:
                return
            dy = self.grad_of(y)
            if (FUSE_POOL_BWD and (k, s, p) == (2, 2, 0) and getattr(x, "_plain_bn_relu", False)
                    and lib.mi355_bn_bwd_pool2_ok(x.H, x.W, x.C, self.code)):
                # x = relu(bn(conv(.))): the layer's two BatchNorm backward passes add the pooled gradient on the fly instead of a
                # scatter pass over dx — on top of the other consumers' parts (a U-Net's skip: they come later in the forward, so
                # they are in dx by then), or ALONE when the pooling is the only consumer (VGG.py's feature stack): conv_bn_act's
                # rule looks whether anything has been written
                x._lazy_pool = dy
                return
            acc = self.acc_flag(x)
            xg = self.grad_of(x)
            self.bwd.append(Launch("mi355_maxpool_bwd", x, x.ld, dy, dy.ld, xg, xg.ld, x.N, x.H, x.W, x.C, k, s, p, acc, self.code))
        self.rule(rule)
        return y

    def add(self, a, b, out=None):
        y = out if out is not None else self.new_tensor(a.N, a.H, a.W, a.C)
        self.fwd.append(Launch("mi355_add", a, a.ld, b, b.ld, y, y.ld, a.M, a.C, self.code))
        y.needs_grad = a.needs_grad or b.needs_grad

        def rule():
            if not y.needs_grad:
                return
            dy = self.grad_of(y)
            for t in (a, b):
                if t.needs_grad:
                    acc = self.acc_flag(t)
                    g = self.grad_of(t)
                    self.bwd.append(Launch("mi355_add", dy, dy.ld, g if acc else None, g.ld, g, g.ld, a.M, a.C, self.code))
        self.rule(rule)
        return y

    def copy(self, a, out):
        """out = a (a plain skip tensor placed into a concat slice, R2U_Net.py:89)."""
        self.fwd.append(Launch("mi355_add", a, a.ld, None, 0, out, out.ld, a.M, a.C, self.code))
        out.needs_grad = a.needs_grad

        def rule():
            if not a.needs_grad:
                return
            dy = self.grad_of(out)
            acc = self.acc_flag(a)
            ga = self.grad_of(a)
            self.bwd.append(Launch("mi355_add", dy, dy.ld, ga if acc else None, ga.ld, ga, ga.ld, a.M, a.C, self.code))
        self.rule(rule)
        return out

    def head(self, mod, x):
        """Classifier head: a Linear or a Sequential ending in one (helpers.py:124-143 may have
        inserted Dropout in front); the last Linear's fp32 output is the model output."""
        mods = list(mod) if isinstance(mod, nn.Sequential) else [mod]
        assert isinstance(mods[-1], nn.Linear), "classifier head must end in nn.Linear"
        v = self.seq(mods[:-1], x) if len(mods) > 1 else x
        return self.linear(v, mods[-1], relu=False, is_output=True)

    # ---- attention gate (AttentionUNet.py:29-54) -------------------------------------------------------------------
    def gate(self, att, g, x, out=None):
        cg, bg = att.W_g[0], att.W_g[1]
        cx, bx = att.W_x[0], att.W_x[1]
        cp, bp = att.psi[0], att.psi[1]
        F_int = cg.out_channels
        M = x.M
        g1, g1_bwd = self.conv_raw(g, cg, stats=True)
        sg = self._bn_coeffs(g1, bg, self._last_stat_rows)
        x1, x1_bwd = self.conv_raw(x, cx, stats=True)
        sx = self._bn_coeffs(x1, bx, self._last_stat_rows)
        # psi_in = relu(bn(g1) + bn(x1)) is not materialised when both directions recompute it from the raw branch outputs
        fused = FUSE_GATE_BWD and bool(lib.mi355_gate_psi_fwd_ok(F_int, self.code))
        z = self.f32(M)
        nb = lib.mi355_rowreduce_blocks(M)

        def psi_fwd(part):
            self.fwd.append(Launch("mi355_gate_psi_fwd", g1, g1.ld, x1, x1.ld, sg["scale"], sg["shift"], sx["scale"], sx["shift"],
                                   cp.weight, cp.bias, z, part, M, F_int, self.code, nbytes=2 * M * F_int * self.esz + 4 * M))
        if fused:
            p = None
            self.acts.append(("relu_pre2", g1, sg["scale"], sg["shift"], x1, sx["scale"], sx["shift"]))
        else:
            p = self.new_tensor(x.N, x.H, x.W, F_int)
            self.fwd.append(Launch("mi355_bn_act", g1, g1.ld, sg["scale"], sg["shift"], x1, x1.ld, sx["scale"], sx["shift"],
                                   None, 0, p, p.ld, M, F_int, 1, self.code))
            self.acts.append(("relu", p))
        sp = {k: self.f32(1) for k in ("scale", "shift", "mean", "invstd")}
        self.see(cp.weight, cp.bias, bp.weight, bp.bias)
        if self.training:
            part = self.ws_f32(nb * 2)
            if fused:
                psi_fwd(part)
            else:
                self.fwd.append(Launch("mi355_rowdot_fwd", p, p.ld, cp.weight, cp.bias, z, part, M, F_int, 0, 1, self.code))
            track = bp.track_running_stats
            self.fwd.append(Launch("mi355_bn_finalize", part, nb, M, 1, bp.weight, bp.bias,
                                   bp.running_mean if track else None, bp.running_var if track else None,
                                   bp.num_batches_tracked if track else None, float(bp.momentum or 0.1), float(bp.eps),
                                   sp["scale"], sp["shift"], sp["mean"], sp["invstd"]))
        else:
            if fused:
                psi_fwd(None)
            else:
                self.fwd.append(Launch("mi355_rowdot_fwd", p, p.ld, cp.weight, cp.bias, z, None, M, F_int, 0, 1, self.code))
            self.fwd.append(Launch("mi355_bn_eval_coeffs", bp.weight, bp.bias, bp.running_mean, bp.running_var, float(bp.eps), 1,
                                   sp["scale"], sp["shift"]))
        y = out if out is not None else self.new_tensor(x.N, x.H, x.W, x.C)
        self.fwd.append(Launch("mi355_gate_mul_fwd", x, x.ld, z, sp["scale"], sp["shift"], y, y.ld, M, x.C, self.code))
        trainable = any(q.requires_grad for q in att.parameters())
        y.needs_grad = x.needs_grad or g.needs_grad or trainable

        def rule():
            if not y.needs_grad:
                return
            dy = self.grad_of(y)
            # x*psi : dx += dy*psi, dzn = (sum_c dy*x) psi (1-psi) and its BN(1) reductions
            dzn = self.f32(M)
            part2 = self.ws_f32(nb * 2)
            if x.needs_grad:
                acc = self.acc_flag(x)
                xg = self.grad_of(x)
            else:
                acc, xg = 0, self.new_tensor(x.N, x.H, x.W, x.C)
            self.bwd.append(Launch("mi355_gate_mul_bwd", dy, dy.ld, x, x.ld, z, sp["scale"], sp["shift"], sp["mean"], sp["invstd"],
                                   xg, xg.ld, acc, dzn, part2, M, x.C, self.code))
            sums = self.f32(2)
            gref, gbeta = self.pgrad(bp.weight)
            bref, _ = self.pgrad(bp.bias)
            self.bwd.append(Launch("mi355_bn_bwd_finalize", part2, nb, 1, sums, gref, bref, gbeta))
            dz = self.f32(M)
            self.bwd.append(Launch("mi355_bn1_bwd_apply", dzn, z, bp.weight, sp["mean"], sp["invstd"], sums, dz, M))
            wref, wbeta = self.pgrad(cp.weight)
            bref2, bbeta = self.pgrad(cp.bias)
            if fused:
                # psi conv (F_int -> 1) and the two normalised branches in two passes: dp = dz * w masked by p > 0 is recomputed from
                # the raw branch outputs where it is needed (mi355_rowdot_bwd would write it, four BatchNorm passes read it)
                # (a buffer of its own, not the shared workspace: the side stream reads it while the main stream moves on)
                part3 = self.f32(nb * 5 * F_int) if SIDE_COLSUM else self.ws_f32(nb * 5 * F_int)
                co = (sg["scale"], sg["shift"], sg["mean"], sg["invstd"], sx["scale"], sx["shift"], sx["mean"], sx["invstd"])
                self.bwd.append(Launch("mi355_gate_bn_bwd_reduce", dz, g1, g1.ld, x1, x1.ld, *co, cp.weight, part3, M, F_int, self.code,
                                       nbytes=2 * M * F_int * self.esz + 4 * M))
                nbf = min(nb, lib.mi355_gate_bn_bwd_reduce_rows(M))
                sums = []
                for q1, bn_, bias_ in ((1, bg, cg.bias), (2, bx, cx.bias)):
                    sm = self.f32(2 * F_int)
                    need_pg = bn_.weight.requires_grad
                    if need_pg:
                        gref_, gbeta_ = self.pgrad(bn_.weight)
                        bref_, _ = self.pgrad(bn_.bias)
                    self.bwd.append(Launch("mi355_bn_bwd_finalize_at", part3, nbf, 5, 0, q1, F_int, sm, gref_ if need_pg else None,
                                           bref_ if need_pg else None, gbeta_ if need_pg else 0.0))
                    if bias_ is not None and bias_.requires_grad and id(bias_) not in self._grad_first:      # (see _bn_bwd)
                        self.pgrad(bias_)
                        self.zero_grad_params.append(bias_)
                    sums.append(sm)
                dg1, dx1 = self.grad_of(g1), self.grad_of(x1)
                self.bwd.append(Launch("mi355_gate_bn_bwd_apply", dz, g1, g1.ld, x1, x1.ld, *co, cp.weight, bg.weight, bx.weight,
                                       sums[0], sums[1], dg1, dg1.ld, dx1, dx1.ld, M, F_int, self.code,
                                       nbytes=4 * M * F_int * self.esz + 4 * M))
                # (the psi convolution's own gradients only feed the optimiser: folded behind the apply pass, off the main stream)
                self.bwd.append(Launch("mi355_colsum_finalize", self._ws_off(part3, 3 * F_int * 4), nb, 5, F_int, wref, wbeta, side=SIDE_COLSUM))
                self.bwd.append(Launch("mi355_colsum_finalize", self._ws_off(part3, 4 * F_int * 4), nb, 5 * F_int, 1, bref2, bbeta, side=SIDE_COLSUM))
                g1_bwd(dg1, bias_done=True)
                x1_bwd(dx1, bias_done=True)
                return
            # psi conv (F_int -> 1): dp = dz*w masked by p>0, dw, db
            dp = self.new_tensor(x.N, x.H, x.W, F_int)
            part3 = self.ws_f32(nb * 2 * F_int)
            self.bwd.append(Launch("mi355_rowdot_bwd", dz, p, p.ld, cp.weight, dp, dp.ld, part3, M, F_int, 1, 0, 1, 0, self.code))
            self.bwd.append(Launch("mi355_colsum_finalize", part3, nb, 2, F_int, wref, wbeta))
            self.bwd.append(Launch("mi355_colsum_finalize", self._ws_off(part3, F_int * 4), nb, 2 * F_int, 1, bref2, bbeta))
            # the two normalised branches share dp
            dg1 = self._bn_bwd(dp, None, g1, bg, sg, False, bias=cg.bias)
            g1_bwd(dg1, bias_done=True)
            dx1 = self._bn_bwd(dp, None, x1, bx, sx, False, bias=cx.bias)
            x1_bwd(dx1, bias_done=True)
        self.rule(rule)
        return y

    # ---- 1x1 head producing the fp32 logit map [N,K,H,W] (AttentionUNet.py:84,119; R2AttU_Net.py:117,156; ResnetUnet.py:58) ------
    def logit_conv(self, x, conv):
        """``out_channel`` = K per-pixel dot products (K = 1 in every configuration the reference runs); plane k of the
        NCHW fp32 output is one ``mi355_rowdot_fwd`` launch, the loss reads the buffer in place."""
        assert conv.kernel_size[0] == 1 and conv.stride[0] == 1 and conv.padding[0] == 0
        M, K, C, HW = x.M, conv.out_channels, x.C, x.H * x.W
        z = self.f32(M * K)
        self.see(conv.weight, conv.bias)
        last = self.fwd[-1] if self.fwd else None
        fused = (FUSE_HEAD and K == 1 and x._plain_bn_relu and x._bn_src is not None and last is not None and last.name == "mi355_bn_act"
                 and last.args[10] is x and self.acts and self.acts[-1] == ("relu", x) and bool(lib.mi355_gate_psi_fwd_ok(C, self.code))
                 and (x.needs_grad or not conv.weight.requires_grad))      # (the head's weight gradient comes out of the layer's backward passes)
        if fused:
            # x = relu(bn(y)) is read by this convolution only: one pass over y computes it on the fly (mi355_gate_psi_fwd with one
            # normalised operand) — the activation is stored in neither direction (the backward recomputes it from y as well)
            y, st = x._bn_src
            self.fwd[-1] = Launch("mi355_gate_psi_fwd", y, y.ld, None, 0, st["scale"], st["shift"], None, None, conv.weight, conv.bias, z,
                                  None, M, C, self.code, nbytes=M * C * self.esz + 4 * M)
            self.acts[-1] = ("relu_pre", y, st["scale"], st["shift"])
        else:
            for k in range(K):
                self.fwd.append(Launch("mi355_rowdot_fwd", x, x.ld, (conv.weight, k * C * 4), (conv.bias, k * 4) if conv.bias is not None else None,
                                       (z, k * HW * 4), None, M, C, HW, K, self.code))
        self.output = ("z", z, (x.N, K, x.H, x.W))
        needs = x.needs_grad or conv.weight.requires_grad
        if self.want_grad and needs:
            self.dout = self.f32(M * K)

        def rule():
            if not needs:
                return
            if fused:
                x._lazy_head = (self.dout, conv)      # conv_bn_act's rule (it runs next) emits the two passes
                return
            nb = lib.mi355_rowreduce_blocks(M)
            dx = None
            if x.needs_grad:
                if self.acc_flag(x):
                    raise NotImplementedError("logit_conv input with several consumers")
                dx = self.grad_of(x)
            for k in range(K):
                part = self.ws_f32(nb * 2 * C)
                self.bwd.append(Launch("mi355_rowdot_bwd", (self.dout, k * HW * 4), x, x.ld, (conv.weight, k * C * 4), dx,
                                       dx.ld if dx is not None else 0, part, M, C, 0, HW, K, 1 if k else 0, self.code))
                if conv.weight.requires_grad:
                    wref, wbeta = self.pgrad(conv.weight)
                    self.bwd.append(Launch("mi355_colsum_finalize", part, nb, 2, C, GRef(wref.tensor, wref.off + k * C * 4, wref.param),
                                           wbeta if k == 0 else 0.0))
                    if conv.bias is not None:
                        bref, bbeta = self.pgrad(conv.bias)
                        self.bwd.append(Launch("mi355_colsum_finalize", self._ws_off(part, C * 4), nb, 2 * C, 1,
                                               GRef(bref.tensor, bref.off + k * 4, bref.param), bbeta if k == 0 else 0.0))
        self.rule(rule)
        return z

    # ---- classifier heads ---------------------------------------------------------------------------------------------
    def global_pool(self, x, is_max):
        v = V(self.f32(x.N * x.C), x.N, x.C)
        am = self._alloc(x.N * x.C, torch.int32)
        self.fwd.append(Launch("mi355_global_pool_fwd", x, x.ld, v, am, x.N, x.H * x.W, x.C, 1 if is_max else 0, self.code))
        if is_max:
            self.acts.append(("gmax", x, am))
        v.needs_grad = x.needs_grad

        def rule():
            if not v.needs_grad:
                return
            dv = self.grad_of(v)
            if self.acc_flag(x):
                raise NotImplementedError("global_pool input with several consumers")
            xg = self.grad_of(x)
            self.bwd.append(Launch("mi355_global_pool_bwd", dv, am, xg, xg.ld, x.N, x.H * x.W, x.C, 1 if is_max else 0, self.code))
        self.rule(rule)
        return v

    def adaptive_avgpool(self, x, oh, ow):
        """nn.AdaptiveAvgPool2d((oh, ow)) + Flatten -> fp32 [N, C*oh*ow] in NCHW order (torchvision VGG head)."""
        v = V(self.f32(x.N * x.C * oh * ow), x.N, x.C * oh * ow)
        self.fwd.append(Launch("mi355_adaptive_avgpool_fwd", x, x.ld, v, x.N, x.H, x.W, x.C, oh, ow, self.code))
        v.needs_grad = x.needs_grad

        def rule():
            if not v.needs_grad:
                return
            dv = self.grad_of(v)
            if self.acc_flag(x):
                raise NotImplementedError("adaptive_avgpool input with several consumers")
            xg = self.grad_of(x)
            self.bwd.append(Launch("mi355_adaptive_avgpool_bwd", dv, xg, xg.ld, x.N, x.H, x.W, x.C, oh, ow, self.code))
        self.rule(rule)
        return v

    def linear(self, v, lin, relu=False, is_output=False):
        O = lin.out_features
        y = V(self.f32(v.B * O), v.B, O)
        self.see(lin.weight, lin.bias)
        self.fwd.append(Launch("mi355_linear_fwd", v, lin.weight, lin.bias, y, v.B, v.F, O, 1 if relu else 0))
        if relu:
            self.acts.append(("relu_v", y))
        y.needs_grad = v.needs_grad or lin.weight.requires_grad
        if is_output:
            self.output = ("v", y, (v.B, O))
            if self.want_grad and y.needs_grad:
                self.dout = self.f32(v.B * O)
                y._grad = V(self.dout, v.B, O)

        def rule():
            if not y.needs_grad:
                return
            dy = self.grad_of(y)
            dx = None
            if v.needs_grad:
                if self.acc_flag(v):
                    raise NotImplementedError("linear input with several consumers")
                dx = self.grad_of(v)
            dw = db = None
            beta = 0.0
            if lin.weight.requires_grad:
                dw, beta = self.pgrad(lin.weight)
                if lin.bias is not None:
                    db, _ = self.pgrad(lin.bias)
            need = lib.mi355_linear_bwd_scratch(v.B, v.F, O) if dx is not None else 0
            assert need >= 0, "linear layer too large for the int32 scratch query"
            scratch = self.ws_f32(need) if need else None
            self.bwd.append(Launch("mi355_linear_bwd", v, lin.weight, y, dy, dx, dw, db, v.B, v.F, O, 1 if relu else 0, beta, scratch))
        self.rule(rule)
        return y

    def dropout(self, v, p):
        if not self.training or p == 0.0:
            return v
        n = v.B * v.F
        y = V(self.f32(n), v.B, v.F)
        mask = self._alloc(n, torch.uint8)
        seed, counter = self.engine.dropout_stream()
        self.fwd.append(Launch("mi355_dropout_fwd", v, y, mask, n, float(p), seed, counter))
        y.needs_grad = v.needs_grad

        def rule():
            if not y.needs_grad:
                return
            dy = self.grad_of(y)
            if self.acc_flag(v):
                raise NotImplementedError("dropout input with several consumers")
            dv = self.grad_of(v)
            self.bwd.append(Launch("mi355_dropout_bwd", dy, mask, dv, n, float(p)))
        self.rule(rule)
        return y

    # ---- nn.Sequential walker with the fusions the reference's blocks allow ----------------------------------------------
    def seq(self, mods, x, out=None):
        mods = list(mods)
        i, n = 0, len(mods)
        while i < n:
            m = mods[i]
            last = lambda j: j >= n - 1
            up = False
            if isinstance(m, nn.Upsample):
                assert float(m.scale_factor) == 2.0 and m.mode == "nearest" and isinstance(mods[i + 1], nn.Conv2d)
                up, i = True, i + 1
                m = mods[i]
            if isinstance(m, nn.Conv2d):
                nxt = mods[i + 1] if i + 1 < n else None
                nxt2 = mods[i + 2] if i + 2 < n else None
                if isinstance(nxt, nn.BatchNorm2d):
                    act = isinstance(nxt2, nn.ReLU)
                    step = 3 if act else 2
                    x = self.conv_bn_act(x, m, nxt, act=act, up=up, out=out if last(i + step - 1) else None)
                    i += step
                elif isinstance(nxt, nn.ReLU):
                    x = self.conv_act(x, m, relu=True, up=up, out=out if last(i + 1) else None)
                    i += 2
                else:
                    x = self.conv_act(x, m, relu=False, up=up, out=out if last(i) else None)
                    i += 1
            elif isinstance(m, nn.ConvTranspose2d):
                x = self.conv_transpose(x, m, out=out if last(i) else None)
                i += 1
            elif isinstance(m, nn.BatchNorm2d):
                act = i + 1 < n and isinstance(mods[i + 1], nn.ReLU)
                x = self.bn_act(x, m, act=act, out=out if last(i + (1 if act else 0)) else None)
                i += 2 if act else 1
            elif isinstance(m, nn.MaxPool2d):
                k = m.kernel_size if isinstance(m.kernel_size, int) else m.kernel_size[0]
                s = m.stride if isinstance(m.stride, int) else m.stride[0]
                p = m.padding if isinstance(m.padding, int) else m.padding[0]
                x = self.maxpool(x, k, s, p)
                i += 1
            elif isinstance(m, nn.Sequential):
                x = self.seq(m, x, out=out if last(i) else None)
                i += 1
            elif isinstance(m, (nn.AdaptiveAvgPool2d, nn.AdaptiveMaxPool2d)):
                osz = m.output_size if isinstance(m.output_size, (tuple, list)) else (m.output_size, m.output_size)
                if tuple(osz) == (1, 1):
                    x = self.global_pool(x, isinstance(m, nn.AdaptiveMaxPool2d))
                else:
                    assert isinstance(m, nn.AdaptiveAvgPool2d), "AdaptiveMaxPool2d: only output size 1"
                    x = self.adaptive_avgpool(x, int(osz[0]), int(osz[1]))
                i += 1
            elif isinstance(m, nn.Flatten) or isinstance(m, nn.Identity):
                i += 1
            elif isinstance(m, nn.Linear):
                relu = i + 1 < n and isinstance(mods[i + 1], nn.ReLU)
                x = self.linear(x, m, relu=relu)
                i += 2 if relu else 1
            elif isinstance(m, nn.Dropout):
                x = self.dropout(x, m.p)
                i += 1
            else:
                raise NotImplementedError(f"no MI355X lowering for {type(m).__name__}")
        return x

    # ---- finish -------------------------------------------------------------------------------------------------------------
    def finish(self):
        if BN_ACT_WINDOWS:
            # plain BatchNorm apply passes on even images (no second operand, no residual, not absorbed by a pooling / gate / head
            # peephole meanwhile): the window-ordered kernel without a pooled output — the same values, ≈8 % faster
            for i, l in enumerate(self.fwd):
                a_ = l.args
                if l.name == "mi355_bn_act" and a_[4] is None and isinstance(a_[10], T) and a_[10].H % 2 == 0 \
                        and a_[10].W % 2 == 0 and a_[12] == a_[10].M:
                    t = a_[10]
                    if a_[8] is None:
                        self.fwd[i] = Launch("mi355_bn_act_pool2", a_[0], a_[1], a_[2], a_[3], t, t.ld, None, 0, t.N, t.H, t.W, t.C, a_[14],
                                             self.code, nbytes=l.bytes)
                    elif BN_ACT_WINDOWS_RES:      # residual added before / after the activation (ResNet.py:43, R2AttU_Net.py:44)
                        self.fwd[i] = Launch("mi355_bn_act_windows", a_[0], a_[1], a_[2], a_[3], a_[8], a_[9], t, t.ld, t.N, t.H, t.W, t.C,
                                             a_[14], self.code, nbytes=l.bytes)
        if getattr(self, "_xcol", None) is not None:
            # the stem reads the im2col of the input: that pack takes the place of the plain one as launch 0 (the launch whose source
            # pointer Plan.run_forward patches to the caller's tensor)
            if self._xin_users:
                raise NotImplementedError("network input read both by a 3x3 stem (as im2col) and by another layer")
            col = [l for l in self.pre if l.name == "mi355_pack_input_im2col3"]
            self.pre = col + [l for l in self.pre if l is not self._xin_pack and l.name != "mi355_pack_input_im2col3"]
        self.static_pack, self.static_params = None, []
        if self._pack_table:
            def pack_launch(entries):
                rows = [[w.data_ptr(), wf.data_ptr(), wb.data_ptr() if wb is not None else 0, co, ci, cip, taps, tr,
                         sc.data_ptr() if sc is not None else 0] for (w, wf, wb, co, ci, cip, taps, tr, sc) in entries]
                table = torch.tensor(rows, dtype=torch.int64).to(self.device)
                self.keep.append(table)
                return Launch("mi355_pack_conv_weights_batched", table, len(rows), len(rows[0]), self.code)
            # a training plan re-packs what the optimiser changes every step; the packs of FROZEN parameters (the ResNet-50
            # encoder of ResnetUnet.py:60-66, a classifier's backbone in stage 1 of helpers.py:246-262) are refreshed only
            # when those parameters change (Plan.refresh_static_packs watches their version counters).  Packs with an
            # eval-mode BatchNorm scale folded in depend on running statistics that kernels update: they stay per-run.
            frozen = [e for e in self._pack_table if STATIC_PACKS and self.want_grad and not e[0].requires_grad and e[8] is None]
            live = [e for e in self._pack_table if not any(e is f for f in frozen)]
            if frozen:
                self.static_pack = pack_launch(frozen)
                self.static_params = [e[0] for e in frozen]
            if live:
                self.pre.append(pack_launch(live))
        for r in reversed(self._rules):
            r()
        assert not self._pending_wgrad, "a shared convolution's weight gradient is still waiting for an application's backward"
        # The side launches BEHIND the last main-stream launch of the backward (the first layer's weight gradient and its reduce)
        # have nothing left to overlap with: on the side stream they only put a cross-queue join (~50 us until the main queue sees the
        # side queue's signal, production trace profiles/r04d_trace_step_timeline.txt) in front of the optimiser.  They run on the
        # main stream; the join then finds the side stream long finished.  MI355_TAIL_ON_MAIN=0 switches it off (A/B).
        # (the slab workspace is side-stream property, tests/test_plan_cpu.py::test_slab_workspace_is_owned_by_one_stream: the tail
        #  gets a slab of its own, sized from its reduce launch)
        if os.environ.get("MI355_TAIL_ON_MAIN", "1") != "0":
            tail = []
            for l in reversed(self.bwd):
                if not l.side:
                    break
                tail.append(l)
            need = 0
            for l in tail:
                if l.name == "mi355_conv2d_wgrad_reduce":       # (ws, splits, dw, Co, Ci, Ci_real, KH, KW, transposed, beta)
                    need = max(need, int(l.args[1]) * int(l.args[3]) * int(l.args[4]) * int(l.args[6]) * int(l.args[7]) * 4)
            own = self._alloc(need, torch.uint8) if need else None
            if all(not isinstance(a, Ws) or (a.kind == "bytes" and own is not None) for l in tail for a in l.args):
                for l in tail:
                    l.side = False
                    l.args = tuple(own if isinstance(a, Ws) else a for a in l.args)
        # resolve _WsOff into (tensor, byte offset) late-bound pairs
        plan = Plan(self)
        for lst in (plan.pre, plan.fwd, plan.bwd):
            for l in lst:
                if any(isinstance(a, Builder._WsOff) for a in l.args):
                    l.args = tuple((plan.ws[a.ws.kind], a.off) if isinstance(a, Builder._WsOff) else a for a in l.args)
        return plan
