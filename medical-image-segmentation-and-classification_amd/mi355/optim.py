"""Gradient clipping and AdamW on the engine's flat fp32 buffers (utils/helpers.py:251,279,304,
332-336).  One launch each over contiguous HBM instead of per-tensor loops; lr, step count and
the clip coefficient stay in device memory so the whole step is graph-capturable."""
from __future__ import annotations

import weakref
from typing import Iterable

import torch

from .lib import lib


_REGISTRY = {}     # storage pointer of a flat parameter buffer -> (weakref(flat_p), weakref(flat_g))


def register_flat(flat_p, flat_g):
    """Called by Engine.flatten().  Weak references: a model that re-flattens (``.to()`` onto another device) or is
    dropped must not keep its old parameter / gradient buffers alive through this table."""
    for k in [k for k, (wp, wg) in _REGISTRY.items() if wp() is None or wg() is None]:
        del _REGISTRY[k]
    _REGISTRY[flat_p.untyped_storage().data_ptr()] = (weakref.ref(flat_p), weakref.ref(flat_g))


def _locate(params):
    """Locate params inside an engine's flat buffers -> (flat_p, flat_g, runs) or None.  ``runs`` are the maximal
    contiguous [lo, hi) element ranges the parameters cover (holes of up to 3 floats are the 16-byte alignment padding
    between neighbouring slots and do not split a run; anything else — a frozen encoder in front of a trainable
    decoder, ResnetUnet.py:60-66 — does)."""
    params = list(params)
    if not params:
        return None
    owner = None
    spans = []
    for p in params:
        reg = _REGISTRY.get(p.untyped_storage().data_ptr())
        if reg is None or reg[0]() is None or reg[1]() is None:
            return None
        fp, fg = reg[0](), reg[1]()
        if owner is None:
            owner = (fp, fg)
        elif fp is not owner[0]:
            return None
        off = (p.data_ptr() - fp.data_ptr()) // 4
        spans.append((off, off + p.numel()))
    spans.sort()
    runs = [list(spans[0])]
    for a0, a1 in spans[1:]:
        if a0 < runs[-1][1]:
            return None                      # overlapping views: not the engine's layout
        if a0 - runs[-1][1] <= 3:
            runs[-1][1] = a1
        else:
            runs.append([a0, a1])
    return owner[0], owner[1], [tuple(r) for r in runs]


class _Shared:
    """Clip state handed from clip_grad_norm_ to the next optimizer.step() on the same flat buffer."""
    by_buffer = {}    # flat parameter buffer pointer -> state (keyed by BUFFER: the ranges clipped and the ranges an optimizer
    #                   owns may differ — helpers.py:333 clips model.parameters(), the optimizer may hold a superset)
    amp = {}          # flat parameter buffer pointer -> device 1/scale of an active loss scaler (amp.GradScaler.unscale_)


def _clip_state(base, nruns):
    st = _Shared.by_buffer.get(base.data_ptr())
    if st is None or st["partial"].numel() < 1024 * nruns:
        dev = base.device
        st = {"partial": torch.empty(1024 * nruns, device=dev), "norm": torch.zeros(1, device=dev),
              "coef": torch.ones(1, device=dev), "finf": torch.zeros(1, device=dev), "fresh": False}
        _Shared.by_buffer[base.data_ptr()] = st
    return st


def _norm_pass(base, g, runs, max_norm, inv_scale):
    """sum of squares of g over ``runs`` -> norm, clip coefficient, found_inf (device scalars)."""
    st = _clip_state(base, len(runs))
    nb = 0
    for lo, hi in runs:
        lo4 = lo - lo % 4                   # (parameter slots are 16-byte aligned: lo4 == lo for runs that start at a slot)
        lib.mi355_sumsq_partial(g[lo4:hi], st["partial"][nb:], hi - lo4)
        nb += lib.mi355_rowreduce_blocks(hi - lo4)
    lib.mi355_clip_coef(st["partial"], nb, float(max_norm), float(inv_scale), _Shared.amp.get(base.data_ptr()), st["norm"],
                        st["coef"], st["finf"], None)
    st["fresh"] = True
    return st


def clip_grad_norm_(parameters: Iterable[torch.Tensor], max_norm: float, inv_scale: float = 1.0):
    """torch.nn.utils.clip_grad_norm_ semantics (global L2 norm over the parameters that have a gradient,
    coef = min(1, max/(norm+1e-6))).  The scaling itself is folded into the following AdamW launch; returns the norm
    as a 0-dim device tensor (no host sync)."""
    params = [p for p in parameters if p.requires_grad]
    r = _locate(params)
    if r is None:
        raise RuntimeError("mi355.optim.clip_grad_norm_ needs the parameters of a mi355 Net (flat storage); "
                           "there is no torch fallback on this path")
    base, g, runs = r
    st = _norm_pass(base, g, runs, max_norm, inv_scale)
    return st["norm"].view(())


class AdamW(torch.optim.Optimizer):
    """torch.optim.AdamW-compatible front (param_groups / lr schedulers work unchanged) whose step() is one fused
    launch per contiguous TRAINABLE run of a parameter group over the engine's flat buffers.  Like torch.optim.AdamW,
    parameters without a gradient (``requires_grad == False``: the frozen encoder of ResnetUnet.py:60-66 inside
    ``AdamW(model.parameters())``, helpers.py:251) are skipped entirely — no weight decay, no moment update."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._st = []
        self.inv_scale = 1.0          # e.g. 1/world_size: gradients are summed, not averaged, by the all-reduce
        for gr in self.param_groups:
            ps = [p for p in gr["params"]]
            r = _locate(ps)
            if r is None:
                raise RuntimeError("mi355.optim.AdamW needs parameters of a mi355 Net that is already on the GPU "
                                   "(call model.to(device) and run / flatten it first)")
            base, g, runs = r
            lo, hi = runs[0][0], runs[-1][1]
            dev = base.device
            self._st.append({
                "p": base, "g": g, "lo": lo, "hi": hi, "runs": {},
                "m": torch.zeros(hi - lo, device=dev), "v": torch.zeros(hi - lo, device=dev),
                "lr": torch.tensor([gr["lr"]], dtype=torch.float32, device=dev), "lr_host": gr["lr"],
                "step": torch.zeros(1, dtype=torch.int32, device=dev),
                "one": torch.ones(1, device=dev), "scratch": torch.zeros(2, device=dev),
            })

    def _trainable_runs(self, gr, st):
        """Contiguous ranges of the group's parameters that currently have a gradient (cached per requires_grad mask)."""
        sig = tuple(p.requires_grad for p in gr["params"])
        runs = st["runs"].get(sig)
        if runs is None:
            p0 = gr["params"][0]
            base = st["p"]
            if not (base.data_ptr() <= p0.data_ptr() < base.data_ptr() + base.numel() * 4):
                raise RuntimeError("mi355.optim.AdamW: the model's parameters no longer live in the flat buffer this optimizer "
                                   "was built on (the model was moved / re-flattened); create the optimizer after model.to(device)")
            tr = [p for p in gr["params"] if p.requires_grad]
            r = _locate(tr) if tr else None
            runs = st["runs"][sig] = (r[2] if r is not None else [])
        return runs

    @torch.no_grad()
    def step(self, closure=None):
        consumed = []
        # A loss scaler has unscale_()d this optimizer and nobody called clip_grad_norm_: the inf / nan check must cover the
        # gradients of EVERY group on the buffer before any group steps (torch's GradScaler.step skips the whole optimizer), so
        # it runs once over the union of the groups' trainable runs, not per group.
        unchecked = {}
        for gr, st in zip(self.param_groups, self._st):
            key = st["p"].data_ptr()
            clip = _Shared.by_buffer.get(key)
            if _Shared.amp.get(key) is not None and not (clip is not None and clip["fresh"]):
                unchecked.setdefault(key, (st, []))[1].extend(self._trainable_runs(gr, st))
        for st, runs in unchecked.values():
            if runs:
                _norm_pass(st["p"], st["g"], sorted(runs), 0.0, self.inv_scale)   # found_inf without clipping
        for gr, st in zip(self.param_groups, self._st):
            if gr["lr"] != st["lr_host"]:
                st["lr"].fill_(gr["lr"])
                st["lr_host"] = gr["lr"]
            runs = self._trainable_runs(gr, st)
            if not runs:
                continue
            clip = _Shared.by_buffer.get(st["p"].data_ptr())
            amp_inv = _Shared.amp.get(st["p"].data_ptr())           # a loss scaler has unscale_()d this optimizer
            coef, finf = st["one"], None
            if clip is not None and clip["fresh"]:
                coef = clip["coef"]
                consumed.append(clip)
                if amp_inv is not None:
                    finf = clip["finf"]                              # GradScaler.step: skip on inf / nan gradients
            st["finf"] = finf
            lib.mi355_step_tick(st["step"], finf)
            b1, b2 = gr["betas"]
            base_lo = st["lo"]
            for lo, hi in runs:
                lib.mi355_adamw(st["p"][lo:hi], st["g"][lo:hi], st["m"][lo - base_lo:hi - base_lo], st["v"][lo - base_lo:hi - base_lo],
                                hi - lo, st["lr"], float(b1), float(b2), float(gr["eps"]), float(gr["weight_decay"]), coef,
                                float(self.inv_scale), amp_inv, finf, st["step"])
        for clip in consumed:               # one clip result serves every group of this step, then expires
            clip["fresh"] = False
        return None

    # ---- checkpoints in torch.optim.AdamW's own format (the reference saves only model weights, helpers.py:394-400;
    # SURVEY.md 8f N3 asks for resumable optimiser state) ------------------------------------------------------------
    def state_dict(self):
        """Same structure as torch.optim.AdamW.state_dict(): per-parameter ``step`` / ``exp_avg`` / ``exp_avg_sq``
        (views of the flat moment buffers are copied out), so either optimiser can load the other's checkpoint."""
        state, groups, idx = {}, [], 0
        for gr, st in zip(self.param_groups, self._st):
            step = float(st["step"].item())
            ids = []
            for p in gr["params"]:
                off = (p.data_ptr() - st["p"].data_ptr()) // 4 - st["lo"]
                state[idx] = {"step": torch.tensor(step), "exp_avg": st["m"][off:off + p.numel()].view(p.shape).clone(),
                              "exp_avg_sq": st["v"][off:off + p.numel()].view(p.shape).clone()}
                ids.append(idx)
                idx += 1
            groups.append({**{k: v for k, v in gr.items() if k != "params"}, "params": ids})
        return {"state": state, "param_groups": groups}

    @torch.no_grad()
    def load_state_dict(self, sd):
        groups = sd["param_groups"]
        if len(groups) != len(self.param_groups) or any(len(g["params"]) != len(s["params"]) for g, s in zip(groups, self.param_groups)):
            raise ValueError("loaded state dict has a different parameter-group layout")
        for gr, st, g_in in zip(self.param_groups, self._st, groups):
            for k, v in g_in.items():
                if k != "params":
                    gr[k] = v
            steps = set()
            for p, i in zip(gr["params"], g_in["params"]):
                ps = sd["state"].get(i)
                if ps is None:
                    continue
                off = (p.data_ptr() - st["p"].data_ptr()) // 4 - st["lo"]
                st["m"][off:off + p.numel()].copy_(ps["exp_avg"].reshape(-1))
                st["v"][off:off + p.numel()].copy_(ps["exp_avg_sq"].reshape(-1))
                steps.add(int(float(ps["step"])))
            if len(steps) > 1:
                raise ValueError("mi355.optim.AdamW keeps one step counter per parameter group; the checkpoint has several")
            if steps:
                st["step"].fill_(steps.pop())
            st["lr_host"] = None                  # force the device-side lr to be refreshed from the loaded group

    def zero_grad(self, set_to_none: bool = True):
        # gradients are (re)written, not accumulated, by every backward plan: dropping the views is enough
        for gr in self.param_groups:
            for p in gr["params"]:
                p.grad = None
