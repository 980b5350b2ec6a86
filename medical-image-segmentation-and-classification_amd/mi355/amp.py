"""Loss scaling for the fp16 build (utils/helpers.py:285, 323-336: ``GradScaler`` around backward /
``unscale_`` / clip / ``step`` / ``update``).  Same call sequence as ``torch.amp.GradScaler``; the scale, its
inverse, the growth tracker and ``found_inf`` live in device memory and every decision (skip the step, back
off, grow) is taken by the kernels, so a training step never synchronises with the host."""
from __future__ import annotations

import torch

from .lib import lib
from . import optim as _optim


class GradScaler:
    def __init__(self, device="cuda", init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000,
                 enabled=True):
        self._enabled = bool(enabled)
        self.growth_factor, self.backoff_factor, self.growth_interval = float(growth_factor), float(backoff_factor), int(growth_interval)
        self._init_scale = float(init_scale)
        self._device = torch.device(device)
        self._scale = self._inv = self._tracker = None
        self._stepped = []          # found_inf tensors of the optimizers stepped since the last update()
        self._unscaled = []         # flat-buffer keys registered by unscale_()

    def is_enabled(self):
        return self._enabled

    def _lazy(self, device):
        if self._scale is None:
            self._scale = torch.full((1,), self._init_scale, dtype=torch.float32, device=device)
            self._inv = torch.full((1,), 1.0 / self._init_scale, dtype=torch.float32, device=device)
            self._tracker = torch.zeros(1, dtype=torch.int32, device=device)

    def scale(self, loss):
        if not self._enabled:
            return loss
        self._lazy(loss.device)
        return loss * self._scale.view(())

    def unscale_(self, optimizer):
        """Registers the device-side 1/scale with the optimizer's flat gradient buffers: the division is folded
        into the norm pass of ``mi355.optim.clip_grad_norm_`` and into the AdamW launch (no pass of its own)."""
        if not self._enabled:
            return
        if not isinstance(optimizer, _optim.AdamW):
            raise RuntimeError("mi355.amp.GradScaler works with mi355.optim.AdamW (flat gradient buffers)")
        self._lazy(optimizer._st[0]["p"].device)
        for st in optimizer._st:
            key = st["p"].data_ptr()
            _optim._Shared.amp[key] = self._inv
            self._unscaled.append(key)

    def step(self, optimizer, *args, **kwargs):
        if not self._enabled:
            return optimizer.step(*args, **kwargs)
        if not self._unscaled:
            self.unscale_(optimizer)
        out = optimizer.step(*args, **kwargs)          # AdamW skips itself when found_inf is set
        self._stepped += [st["finf"] for st in optimizer._st if st.get("finf") is not None]
        return out

    def update(self, new_scale=None):
        if not self._enabled:
            return
        if new_scale is not None:
            self._scale.fill_(float(new_scale))
            self._inv.fill_(1.0 / float(new_scale))
        elif self._stepped:
            finf = self._stepped[0] if len(self._stepped) == 1 else torch.stack(self._stepped).amax(0)
            lib.mi355_amp_update(self._scale, self._inv, self._tracker, finf, self.growth_factor, self.backoff_factor,
                                 self.growth_interval)
        for key in self._unscaled:
            _optim._Shared.amp.pop(key, None)
        self._stepped, self._unscaled = [], []

    def get_scale(self):
        return self._init_scale if self._scale is None else float(self._scale)      # (host sync, as in torch)

    def state_dict(self):
        return {"scale": self.get_scale(), "growth_factor": self.growth_factor, "backoff_factor": self.backoff_factor,
                "growth_interval": self.growth_interval,
                "_growth_tracker": 0 if self._tracker is None else int(self._tracker)} if self._enabled else {}

    def load_state_dict(self, sd):
        if not self._enabled or not sd:
            return
        self._init_scale = float(sd["scale"])
        self.growth_factor, self.backoff_factor = float(sd["growth_factor"]), float(sd["backoff_factor"])
        self.growth_interval = int(sd["growth_interval"])
        if self._scale is not None:
            self._scale.fill_(self._init_scale)
            self._inv.fill_(1.0 / self._init_scale)
            self._tracker.fill_(int(sd.get("_growth_tracker", 0)))
