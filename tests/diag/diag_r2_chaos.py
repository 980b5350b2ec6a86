"""One-off CPU control for the R2AttU_Net trained-Dice criterion (VERDICT r2 'weak' item 1): the protocol of
tests/test_gpu_train_bf16.py (64x64, batch 4, lr 1e-3, default init seed 0, learnable synthetic task) run on the CPU oracle
ONLY, several ways that differ by nothing but floating-point summation order / precision:

    fp32 (N threads)   fp32 (1 thread)   fp64   fp32 with the batch's images processed in reversed order

After 8 / 12 / 20 / 32 optimiser steps: Dice of each run on the 32 held-out images, and the largest |dDice| between runs.
If two CPU evaluations of the SAME algorithm drift apart by more than 1e-3, a 1e-3 bound on |Dice_HIP - Dice_oracle| of a
HIP-TRAINED run is a statement about chaos, not about arithmetic.

    python tests/diag/diag_r2_chaos.py [R2AttU_Net|AttentionUNet] [steps...]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import nets                      # noqa: E402
from oracle import train as otrain           # noqa: E402


def task(b, hw, seed):
    x, m = otrain.synthetic_batch(b, hw, seed=seed)
    return 0.6 * x + m * torch.tensor([1.0, -0.7, 0.4]).view(1, 3, 1, 1), m


def dice(logit, m):
    p = (torch.sigmoid(logit) > 0.5).double()
    t = (m > 0.5).double()
    return float((2 * (p * t).sum() + 1e-7) / (p.sum() + t.sum() + 1e-7))


def run(name, marks, dtype=torch.float32, threads=None, flip=False, hw=64, b=4, lr=1e-3):
    if threads:
        torch.set_num_threads(threads)
    batches = [task(b, hw, s) for s in range(4)]
    xv, mv = task(32, hw, 99)
    sd = {k: (v.to(dtype) if v.is_floating_point() else v.clone()) for k, v in nets.default_init_state(name, seed=0).items()}
    opt = otrain.AdamW(nets.param_keys(sd), lr)
    out = {}
    for i in range(max(marks)):
        x, y = batches[i % 4]
        if flip:
            x, y = x.flip(0), y.flip(0)
        loss, _, _ = otrain.train_step(name, sd, x.to(dtype), y.to(dtype), opt, True)
        if i + 1 in marks:
            with torch.no_grad():
                d = dice(nets.NETS[name]({k: v.clone() for k, v in sd.items()}, xv.to(dtype), True), mv)
            out[i + 1] = (d, loss)
    return out


if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "R2AttU_Net"
    marks = [int(a) for a in sys.argv[2:]] or [8, 12, 20, 32]
    n = torch.get_num_threads()
    runs = {}
    for tag, kw in ((f"fp32 x{n} threads", {}), ("fp32 x1 thread", {"threads": 1}), ("fp64", {"dtype": torch.float64, "threads": n}),
                    ("fp32 batch reversed", {"flip": True, "threads": n})):
        t0 = time.time()
        runs[tag] = run(name, marks, **kw)
        print(f"{name} {tag:22s} " + "  ".join(f"step {s}: Dice {d:.5f} loss {l:.5f}" for s, (d, l) in runs[tag].items())
              + f"   ({time.time() - t0:.0f} s)", flush=True)
    for s in marks:
        ds = [r[s][0] for r in runs.values()]
        ls = [r[s][1] for r in runs.values()]
        print(f"step {s}: max |dDice| between CPU runs {max(ds) - min(ds):.2e}; loss spread {(max(ls) - min(ls)) / min(ls):.1%}")
