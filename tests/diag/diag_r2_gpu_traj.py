"""GPU counterpart of diag_r2_chaos.py: the protocol of tests/test_gpu_train_bf16.py[R2AttU_Net] on the HIP path in fp32 and bf16,
Dice on the 32 held-out images and last-batch loss after 8 / 12 / 20 / 32 / 48 steps (no oracle: its numbers are in
diag_r2_chaos.py's output).   python tests/diag/diag_r2_gpu_traj.py [marks...]
R2_DECAY_AT=k R2_LR2=x: the learning rate drops to x after k steps (the conditioned criterion of round 4: the trajectory is allowed to
settle before Dice is read); R2_HELD=n: Dice over n held-out images, in chunks of 32 (train-mode BatchNorm per chunk)."""
import os
import sys

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "medical-image-segmentation-and-classification_amd"))
import torch
from oracle import nets, train as otrain
from mi355 import nn as mnn, optim as moptim
from utils.helpers import get_seg_model


def task(b, hw, seed):
    x, m = otrain.synthetic_batch(b, hw, seed=seed)
    return 0.6 * x + m * torch.tensor([1.0, -0.7, 0.4]).view(1, 3, 1, 1), m


def dice(logit, m):
    p = (torch.sigmoid(logit) > 0.5).double(); t = (m > 0.5).double()
    return float((2 * (p * t).sum() + 1e-7) / (p.sum() + t.sum() + 1e-7))


marks = [int(a) for a in sys.argv[1:]] or [8, 12, 20, 32, 48]
hw, b, lr = 64, 4, 1e-3
batches = [task(b, hw, s) for s in range(4)]
held = int(os.environ.get("R2_HELD", "32"))
decay_at, lr2 = int(os.environ.get("R2_DECAY_AT", "0")), float(os.environ.get("R2_LR2", "1e-4"))
held_sets = [task(32, hw, 99 + c) for c in range(held // 32)]


def held_dice(m):
    num = den = 0.0
    for xv, mv in held_sets:
        p = (torch.sigmoid(m(xv.cuda()).float().cpu()) > 0.5).double(); t = (mv > 0.5).double()
        num += float(2 * (p * t).sum()); den += float(p.sum() + t.sum())
    return (num + 1e-7) / (den + 1e-7)
sd0 = nets.default_init_state("R2AttU_Net", seed=0)
for dtype in (torch.float32, torch.bfloat16):
    m = get_seg_model("r2attunet"); m.load_state_dict(sd0); m.compute_dtype = dtype
    m = m.cuda().train(); opt = moptim.AdamW(m.parameters(), lr=lr, weight_decay=5e-4); crit = mnn.BCEWithLogitsLoss()
    out = []
    for i in range(max(marks)):
        x, y = batches[i % 4]
        if decay_at and i == decay_at:
            for g in opt.param_groups:
                g["lr"] = lr2
        opt.zero_grad(); loss = crit(m(x.cuda()), y.cuda()); loss.backward(); moptim.clip_grad_norm_(m.parameters(), 1.0); opt.step()
        if i + 1 in marks:
            with torch.no_grad():
                out.append((i + 1, held_dice(m), float(loss.detach())))
    print(f"HIP {str(dtype):15s} " + "  ".join(f"step {s}: Dice {d:.5f} loss {l:.5f}" for s, d, l in out), flush=True)
