"""Loss modules backed by the HIP library (utils/helpers.py:244-246 call sites).

``criterion(out, y)`` keeps torch's call shape; when ``out`` comes from a mi355 ``Net`` the
gradient w.r.t. the logits is written by the backward launch straight into the plan's static
``dout`` buffer, so no torch kernel sits between the loss and the model's backward plan."""
from __future__ import annotations

import torch
import torch.nn as nn

from .lib import lib


class _BCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, out, target, plan):
        if target.dtype != torch.float32 or not target.is_contiguous():
            target = target.float().contiguous()
        if target.numel() != out.numel():
            raise ValueError(f"target size {tuple(target.shape)} must match input size {tuple(out.shape)}")
        o = out.detach()
        if not o.is_contiguous():
            o = o.contiguous()
        loss = torch.empty(1, dtype=torch.float32, device=out.device)
        lib.mi355_bce_logits(o, target, loss, None, None, o.numel())
        ctx.o, ctx.t, ctx.plan = o, target, plan
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        o, t, plan = ctx.o, ctx.t, ctx.plan
        n = o.numel()
        dz = plan.dout if (plan is not None and plan.dout is not None and plan.dout.numel() >= n) else \
            torch.empty(n, dtype=torch.float32, device=o.device)
        scratch = torch.empty(1, dtype=torch.float32, device=o.device)
        gs = g.detach().float().reshape(1).contiguous()
        lib.mi355_bce_logits(o, t, scratch, dz, gs, n)
        return dz[:n].view(o.shape), None, None


class BCEWithLogitsLoss(nn.Module):
    """nn.BCEWithLogitsLoss() (mean reduction) on the HIP library."""

    def forward(self, out, target):
        if out.dtype != torch.float32:
            out = out.float()
        return _BCEFn.apply(out, target, getattr(out, "_mi355_plan", None))


class _CEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, out, target, smoothing, plan):
        o = out.detach().contiguous()
        t = target.to(torch.int64).contiguous()
        loss = torch.empty(1, dtype=torch.float32, device=out.device)
        lib.mi355_ce_smooth(o, t, loss, None, None, o.shape[0], o.shape[1], float(smoothing))
        ctx.o, ctx.t, ctx.plan, ctx.s = o, t, plan, float(smoothing)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        o, t, plan = ctx.o, ctx.t, ctx.plan
        n = o.numel()
        dz = plan.dout if (plan is not None and plan.dout is not None and plan.dout.numel() >= n) else \
            torch.empty(n, dtype=torch.float32, device=o.device)
        scratch = torch.empty(1, dtype=torch.float32, device=o.device)
        gs = g.detach().float().reshape(1).contiguous()
        lib.mi355_ce_smooth(o, t, scratch, dz, gs, o.shape[0], o.shape[1], ctx.s)
        return dz[:n].view(o.shape), None, None, None


class CrossEntropyLoss(nn.Module):
    """nn.CrossEntropyLoss(label_smoothing=s) (mean reduction) on the HIP library."""

    def __init__(self, label_smoothing=0.0):
        super().__init__()
        self.label_smoothing = label_smoothing

    def forward(self, out, target):
        return _CEFn.apply(out.float(), target, self.label_smoothing, getattr(out, "_mi355_plan", None))
