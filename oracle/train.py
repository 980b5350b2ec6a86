"""CPU restatement of the reference's optimisation protocol and metrics
(TEST INFRASTRUCTURE — see oracle/__init__.py).

Anchors: utils/helpers.py:219-227 (acc, iou), :231-412 (train);
utils/tester.py:92-193 (segmentation metrics).  All arithmetic is spelled out in
plain tensor math (no torch.optim / torch.nn.utils) so it is an independent
statement of the algorithm; ``tests/test_oracle_pins.py`` checks it against the
reference's own ``train()`` trajectory stored under ``tests/golden/``.
"""
from __future__ import annotations

import math

import torch

from . import nets


# ----------------------------------------------------------------------------
# losses (helpers.py:244-246)
# ----------------------------------------------------------------------------
def bce_with_logits(z, t):
    """nn.BCEWithLogitsLoss() — mean over every element, stable form."""
    return (z.clamp_min(0) - z * t + torch.log1p(torch.exp(-z.abs()))).mean()


def cross_entropy_ls(z, y, smoothing=0.1):
    """nn.CrossEntropyLoss(label_smoothing=0.1): (1-s)*nll + s*mean_c(-logp)."""
    logp = z - torch.logsumexp(z, 1, keepdim=True)
    nll = -logp.gather(1, y[:, None]).squeeze(1)
    uni = -logp.mean(1)
    return ((1 - smoothing) * nll + smoothing * uni).mean()


# ----------------------------------------------------------------------------
# metrics
# ----------------------------------------------------------------------------
def iou_train(pred, mask, t=0.5):
    """helpers.py:223-227 — whole-batch tensor, eps only in the denominator."""
    p = (pred > t).float()
    inter = (p * mask).sum()
    union = ((p + mask) > 0).float().sum()
    return (inter / (union + 1e-7)).item()


def acc(logits, y):
    """helpers.py:219-220."""
    return int((logits.argmax(1) == y).sum().item()), int(y.shape[0])


def seg_metrics(pred, target, threshold=0.5):
    """tester.py:92-193 for ONE sample; percentages."""
    p = (pred > threshold).double()
    t = (target > threshold).double()
    tp = float((p * t).sum()); fp = float((p * (1 - t)).sum()); fn = float(((1 - p) * t).sum())
    union = float(((p + t) > 0).double().sum())
    e = 1e-7
    iou = (tp + e) / (union + e)
    dice = (2 * tp + e) / (float(p.sum()) + float(t.sum()) + e)
    pa = float((p == t).double().sum()) / t.numel()
    prec = (tp + e) / (tp + fp + e)
    rec = (tp + e) / (tp + fn + e)
    f1 = 2 * prec * rec / (prec + rec + e)
    return {"iou": iou * 100, "dice": dice * 100, "pixel_accuracy": pa * 100,
            "precision": prec * 100, "recall": rec * 100, "f1": f1 * 100}


# ----------------------------------------------------------------------------
# optimiser pieces (helpers.py:249-255, 332-336)
# ----------------------------------------------------------------------------
def clip_grad_norm(grads, max_norm=1.0):
    """torch.nn.utils.clip_grad_norm_: global L2; coef = max/(norm+1e-6) clamped to 1."""
    total = math.sqrt(sum(float((g.double() ** 2).sum()) for g in grads))
    coef = min(1.0, max_norm / (total + 1e-6))
    for g in grads:
        g.mul_(coef)
    return total


class AdamW:
    """torch.optim.AdamW defaults (betas .9/.999, eps 1e-8), decoupled decay."""

    def __init__(self, keys, lr, weight_decay=5e-4, betas=(0.9, 0.999), eps=1e-8):
        self.keys = list(keys)
        self.lr, self.wd, self.betas, self.eps = lr, weight_decay, betas, eps
        self.t = 0
        self.m, self.v = {}, {}

    def step(self, sd, grads):
        self.t += 1
        b1, b2 = self.betas
        bc1 = 1 - b1 ** self.t
        bc2 = 1 - b2 ** self.t
        for k in self.keys:
            g = grads.get(k)
            if g is None:
                continue
            p = sd[k]
            if k not in self.m:
                self.m[k] = torch.zeros_like(p); self.v[k] = torch.zeros_like(p)
            m, v = self.m[k], self.v[k]
            p.mul_(1 - self.lr * self.wd)
            m.mul_(b1).add_(g, alpha=1 - b1)
            v.mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = (v.sqrt() / math.sqrt(bc2)).add_(self.eps)
            p.addcdiv_(m, denom, value=-self.lr / bc1)


def cosine_lr(base_lr, epoch_done, t_max, eta_min=0.0):
    """Closed form of CosineAnnealingLR after ``epoch_done`` scheduler.step() calls."""
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * epoch_done / t_max)) / 2


# ----------------------------------------------------------------------------
# one optimisation step + the segmentation training loop
# ----------------------------------------------------------------------------
def forward_backward(net, sd, x, y, seg=True, trainable=None, **net_kw):
    """zero_grad -> forward(train) -> loss -> backward.  Returns (loss, logits, grads).  ``trainable``: the parameter keys that
    require a gradient (default: all) — ResNetUnet(freeze=True) keeps its encoder frozen (ResnetUnet.py:60-66)."""
    fn = nets.NETS[net]
    pk = nets.param_keys(sd) if trainable is None else list(trainable)
    for k in pk:
        sd[k].requires_grad_(True)
        sd[k].grad = None
    out = fn(sd, x, True, **net_kw)
    if seg and out.dim() == 3:
        out = out.unsqueeze(1)
    loss = bce_with_logits(out, y) if seg else cross_entropy_ls(out, y)
    loss.backward()
    grads = {k: sd[k].grad.detach().clone() for k in pk if sd[k].grad is not None}
    for k in pk:
        sd[k].requires_grad_(False)
        sd[k].grad = None
    return float(loss.detach()), out.detach(), grads


def train_step(net, sd, x, y, opt: AdamW, seg=True, trainable=None, **net_kw):
    """helpers.py:320-336 on CPU (autocast and GradScaler are disabled no-ops there)."""
    loss, out, grads = forward_backward(net, sd, x, y, seg, trainable, **net_kw)
    gnorm = clip_grad_norm(list(grads.values()), 1.0)
    with torch.no_grad():
        opt.step(sd, grads)
    return loss, out, gnorm


def train_seg(net, sd, train_batches, val_batches, epochs, lr, log=None):
    """Segmentation branch of helpers.train (helpers.py:249-255, 292-406) without
    checkpoint I/O: AdamW(all, lr, wd 5e-4) + CosineAnnealingLR(T_max=epochs) stepped
    per epoch, val loss = sample-weighted mean, val IoU = mean over batches, best =
    lowest val loss, patience 10.  Returns (best_score, history)."""
    fn = nets.NETS[net]
    opt = AdamW(nets.param_keys(sd), lr)
    n_train = sum(x.shape[0] for x, _ in train_batches)
    n_val = sum(x.shape[0] for x, _ in val_batches)
    best, patience, hist = float("inf"), 0, []
    for ep in range(1, epochs + 1):
        run = 0.0
        for x, y in train_batches:
            loss, _, _ = train_step(net, sd, x, y, opt, True)
            run += loss * x.shape[0]
        vl = vm = 0.0
        with torch.no_grad():
            for x, y in val_batches:
                out = fn(sd, x, False)
                vl += float(bce_with_logits(out, y)) * x.shape[0]
                vm += iou_train(torch.sigmoid(out), y)
        vl /= n_val
        hist.append((run / n_train, vl, vm / len(val_batches)))
        if log:
            log(f"[{net}] Ep{ep}: TrainLoss {run / n_train:.3f} | ValLoss {vl:.3f} | IoU {vm / len(val_batches):.3f}")
        opt.lr = cosine_lr(lr, ep, epochs)
        if vl < best:
            best, patience = vl, 0
        else:
            patience += 1
        if patience >= 10:
            break
    return best, hist


# ----------------------------------------------------------------------------
# deterministic synthetic data
# ----------------------------------------------------------------------------
def synthetic_batch(b, hw, seed=0, classes=None):
    """images ~ N(0,1); seg target = one filled ellipse per image (25-40 % foreground);
    cls labels uniform over ``classes``."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(b, 3, hw, hw, generator=g)
    if classes is not None:
        return x, torch.randint(0, classes, (b,), generator=g)
    yy, xx = torch.meshgrid(torch.arange(hw, dtype=torch.float32),
                            torch.arange(hw, dtype=torch.float32), indexing="ij")
    m = torch.zeros(b, 1, hw, hw)
    for i in range(b):
        r = torch.rand(4, generator=g)
        cy = (0.4 + 0.2 * r[0]) * hw; cx = (0.4 + 0.2 * r[1]) * hw
        area = (0.25 + 0.15 * r[2]) * hw * hw
        ratio = 0.7 + 0.6 * r[3]
        a = math.sqrt(area / math.pi * ratio); bb = math.sqrt(area / math.pi / ratio)
        m[i, 0] = ((((yy - cy) / a) ** 2 + ((xx - cx) / bb) ** 2) <= 1.0).float()
    return x, m


def closed_form_input(b, hw, c=3):
    """RNG-free input/mask pair used by the golden fixtures."""
    n = b * c * hw * hw
    i = torch.arange(n, dtype=torch.float64)
    x = (1.3 * torch.sin(0.113 * i + 0.3) + 0.4 * torch.cos(0.0171 * i)).float().reshape(b, c, hw, hw)
    yy, xx = torch.meshgrid(torch.arange(hw, dtype=torch.float32),
                            torch.arange(hw, dtype=torch.float32), indexing="ij")
    m = torch.zeros(b, 1, hw, hw)
    for k in range(b):
        cy, cx = hw * (0.45 + 0.05 * k), hw * (0.5 - 0.04 * k)
        m[k, 0] = ((((yy - cy) / (0.33 * hw)) ** 2 + ((xx - cx) / (0.27 * hw)) ** 2) <= 1.0).float()
    return x, m
