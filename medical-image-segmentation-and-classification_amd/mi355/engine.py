"""Execution engine behind the drop-in model classes.

``Net`` is an ``nn.Module`` whose module tree (plain ``torch.nn`` containers — they only hold
parameters/buffers, so ``state_dict`` keys, shapes, default init, ``.train()/.eval()``,
``requires_grad`` toggling all behave exactly like the reference classes) is lowered by
``build()`` into a static launch plan (graph.py).  ``forward`` runs that plan on the HIP
library; autograd sees ONE node per model whose backward runs the plan's backward launches
and hands out views of a flat fp32 gradient buffer.

Parameters live in one flat fp32 HBM buffer (and their gradients in a second one) so that
gradient clipping, AdamW and the data-parallel all-reduce are single launches / single
collectives over contiguous memory.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import graph
from .lib import lib

_DEFAULT_DTYPE = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "fp16": torch.float16, "float16": torch.float16,
                  "fp32": torch.float32, "float32": torch.float32}[os.environ.get("MI355_DTYPE", "bf16").lower()]


def set_default_dtype(dtype):
    """Compute/storage dtype of activations for models built afterwards (torch.bfloat16, float16 or float32)."""
    global _DEFAULT_DTYPE
    assert dtype in (torch.bfloat16, torch.float16, torch.float32)
    _DEFAULT_DTYPE = dtype


def get_default_dtype():
    return _DEFAULT_DTYPE


class Engine:
    ALIGN = 4      # floats; keeps every parameter 16-byte aligned inside the flat buffers

    def __init__(self, net: "Net"):
        self.net = net
        self.plans: Dict[tuple, graph.Plan] = {}
        self.flat_p: Optional[torch.Tensor] = None
        self.flat_g: Optional[torch.Tensor] = None
        self.offsets: Dict[int, tuple] = {}
        self.params: List[nn.Parameter] = []
        self._drop_counter = None
        self._drop_seed = 0x5EED
        self.grad_hooks = []            # callables(plan) run after the backward launches (data parallel)
        self.bwd_runner = None          # optional replacement for plan.run_backward (overlapped all-reduce)
        self.pack_epoch = 0             # bumped by invalidate_packs(): frozen-weight packs are rebuilt on the next forward

    # ---- flat parameter / gradient storage -----------------------------------------------------------
    def flatten(self):
        params = [p for _, p in self.net.named_parameters()]
        if not params:
            raise RuntimeError("model has no parameters")
        dev = params[0].device
        total, offs = 0, {}
        for p in params:
            if p.dtype != torch.float32:
                raise RuntimeError("MI355X path keeps fp32 master parameters")
            offs[id(p)] = (total, p.numel())
            total += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p in params:
                o, n = offs[id(p)]
                flat_p[o:o + n].copy_(p.detach().reshape(-1))
                p.data = flat_p[o:o + n].view(p.shape)
        self.flat_p, self.flat_g, self.offsets, self.params = flat_p, flat_g, offs, params
        self.plans.clear()
        from . import optim
        optim.register_flat(flat_p, flat_g)

    def _check_storage(self):
        if self.flat_p is None:
            self.flatten()
            return
        base = self.flat_p.data_ptr()
        for p in (self.params[0], self.params[-1]):
            if p.data_ptr() != base + self.offsets[id(p)][0] * 4 or p.device != self.flat_p.device:
                self.flatten()          # .to(device) / .cuda() re-created the parameter storage
                return

    def invalidate_packs(self):
        """Parameters were written behind autograd's version counters (a broadcast into the flat buffer, ``p.data`` edits, raw
        pointers): every plan re-packs its FROZEN weights on its next forward (trainable ones are re-packed every step anyway)."""
        self.pack_epoch += 1

    def grad_ref(self, p):
        o, _ = self.offsets[id(p)]
        return graph.GRef(self.flat_g, o * 4, p)

    def grad_view(self, p):
        o, n = self.offsets[id(p)]
        return self.flat_g[o:o + n].view(p.shape)

    def dropout_stream(self):
        if self._drop_counter is None:
            self._drop_counter = torch.zeros(1, dtype=torch.int32, device=self.flat_p.device)
        self._drop_seed += 0x9E37
        return self._drop_seed, self._drop_counter

    # ---- plans --------------------------------------------------------------------------------------------
    def plan_for(self, x_shape, training, want_grad, dtype):
        sig = tuple(p.requires_grad for p in self.params) if want_grad else ()
        key = (tuple(x_shape), training, want_grad, dtype, sig)
        plan = self.plans.get(key)
        if plan is None or plan.params_moved():
            b = graph.Builder(self, self.flat_p.device, dtype, training, want_grad)
            xin = b.set_input(x_shape)
            self.net.build(b, xin)
            if b.output is None:
                raise RuntimeError(f"{type(self.net).__name__}.build() did not define an output")
            plan = b.finish()
            plan.has_dropout = self._drop_counter is not None and any(l.name == "mi355_dropout_fwd" for l in plan.fwd)
            self.plans[key] = plan
        return plan

    # ---- execution ------------------------------------------------------------------------------------------
    @staticmethod
    def _stream():
        return torch.cuda.current_stream().cuda_stream

    def run_forward(self, plan, x):
        if x.dtype != torch.float32 or not x.is_contiguous():
            x = x.float().contiguous()
        if plan.training and getattr(plan, "has_dropout", False):
            self._drop_counter += 1
        plan.run_forward(self._stream(), x.data_ptr())      # mi355_pack_input_nchw reads the caller's tensor in place
        kind, buf, shape = plan.output
        t = buf if kind == "z" else buf.buf
        return t[: int(torch.Size(shape).numel())].view(shape)

    def run_backward(self, plan):
        if self.bwd_runner is not None:
            self.bwd_runner(plan, self._stream())
        else:
            plan.run_backward(self._stream())
        for h in self.grad_hooks:
            h(plan)

    def forward(self, x):
        if x.device.type != "cuda":
            raise RuntimeError("the MI355X path needs CUDA/HIP tensors (there is no CPU fallback); "
                               "move the model and its input to a GPU device")
        self._check_storage()
        net = self.net
        dtype = net.compute_dtype or _DEFAULT_DTYPE
        training = net.training
        want_grad = training and torch.is_grad_enabled() and any(p.requires_grad for p in self.params)
        plan = self.plan_for(x.shape, training, want_grad, dtype)
        if not want_grad:
            out = self.run_forward(plan, x)
            return out.clone() if net.clone_eval_output else out
        out = _NetFn.apply(self, plan, x, *self.params)
        out._mi355_plan = plan
        return out


class _NetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, engine, plan, x, *params):
        ctx.engine, ctx.plan = engine, plan
        return engine.run_forward(plan, x)

    @staticmethod
    def backward(ctx, dout):
        engine, plan = ctx.engine, ctx.plan
        if plan.dout is None:
            raise RuntimeError("backward through a plan built without gradients")
        if dout.data_ptr() != plan.dout.data_ptr():
            plan.dout[: dout.numel()].copy_(dout.reshape(-1))
        engine.run_backward(plan)
        written = {id(p) for p in plan.grad_params}
        grads = tuple(engine.grad_view(p) if (id(p) in written and p.requires_grad) else None for p in engine.params)
        return (None, None, None) + grads


class Net(nn.Module):
    """Base class of the drop-in models.  Subclasses create the reference's module tree in
    ``__init__`` and describe the forward topology in ``build(g, x)`` with graph.Builder ops."""

    compute_dtype = None          # None -> engine default (MI355_DTYPE env / set_default_dtype)
    clone_eval_output = True      # eval outputs are detached copies (safe to keep across batches)

    def __init__(self):
        super().__init__()
        object.__setattr__(self, "_mi355_engine", None)

    @property
    def engine(self) -> Engine:
        e = self.__dict__.get("_mi355_engine")
        if e is None:
            e = Engine(self)
            object.__setattr__(self, "_mi355_engine", e)
        return e

    def _apply(self, fn, recurse=True):
        """.to()/.cuda() re-create parameter storage: re-establish the flat fp32 buffers right away so
        that optimisers built from ``model.parameters()`` see the engine's storage."""
        r = super()._apply(fn, recurse)
        p = next(self.parameters(), None)
        if p is not None and p.device.type == "cuda" and p.dtype == torch.float32:
            # only when the storage really moved: a no-op .to(device) / .cuda() must not re-point the parameters away from
            # the flat buffers an optimiser or the data-parallel wrapper already holds
            self.engine._check_storage()
        return r

    def build(self, g: graph.Builder, x: graph.T):
        raise NotImplementedError

    def forward(self, x):
        return self.engine.forward(x)

    def plan_summary(self, x_shape, training=True):
        """(forward launches, backward launches) of the plan for an input shape (host-side only)."""
        self.engine._check_storage()
        p = self.engine.plan_for(tuple(x_shape), training, training, self.compute_dtype or _DEFAULT_DTYPE)
        return p.n_launches
