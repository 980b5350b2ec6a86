"""-m gpu: the north-star Dice criterion (configs C3 and C4 in their own dtypes).  Train Attention U-Net for 32 optimiser steps on a learnable
synthetic task (ellipse visible in the image) three ways from identical weights and batches —
HIP bf16, HIP fp16 (+ loss scaling, helpers.py:285,323-336), HIP fp32, CPU fp32 oracle (reference semantics:
BCEWithLogits, clip 1.0, AdamW wd 5e-4) —
and compare the Dice of the binarised predictions (tester.py:114-134) on 32 held-out images.
Bound: |Dice - Dice_oracle| <= 1e-3 (0..1 scale) for every GPU mode; final losses within 2 %.

R2AttU_Net (R2AttU_Net.py:88-158, config C4: bf16) runs the same protocol for 32 steps; fp32 with the same 1e-3 bound on Dice,
bf16 with 3e-3 — the spread of EQUALLY VALID bf16 arithmetic: tests/diag/diag_r2_bf16_spread.py trains the HIP path once per
combination of the plan-level fusion switches (the same function, bf16 roundings in different places: a gradient rounded before or
after a sum, a dot product with or without a fused multiply-add).  In fp32 all combinations are bit-identical (Dice 0.99088 at
step 32, oracle 0.99080); in bf16 they are 2.5e-3 apart at step 32 (0.99045 .. 0.99291), 0.75e-3 at steps 48 and 64, 2.2e-3 at
80, and the fp32 trajectory itself moves by 3e-3 from one mark to the next (profiles/r03c_r2attunet_bf16_dice_spread.txt).  A
1e-3 bound on ONE bf16 trajectory of this network is met or missed by rounding placement, not by correctness (it held in round 3
until a dot product gained a fused multiply-add), so it is asserted where it is a statement about arithmetic: fp32 training,
and the oracle-trained weights through the bf16 forward.  Why 32 steps: with 108 shared-weight convolutions per forward the early trajectory is chaotic ON THE CPU ALONE — tests/diag/diag_r2_chaos.py
runs this protocol four ways that differ only in summation order / precision (fp32 with 8 threads, 1 thread, the batch reversed;
fp64) and those runs are 3.9e-3 apart in Dice at step 8, 1.9e-3 at step 12, 2.4e-4 at step 20 and 4.1e-4 at step 32, where Dice
has reached its plateau (0.9907-0.9911); their last-batch losses stay 8-20 % apart throughout (tests/test_oracle_kinks.py
re-measures a short version of this control on every CPU run).  The HIP runs (tests/diag/diag_r2_gpu_traj.py): fp32 0.99088, bf16
0.99152 at step 32.  So the Dice bound is asserted where it is a statement about arithmetic (the plateau), the loss bound is the
CPU-vs-CPU spread (15 %), and the ORACLE-TRAINED weights evaluated by the HIP forward must give the oracle's Dice within 1e-3 in
every precision (no trajectory involved)."""
import pytest
import torch

from oracle import nets
from oracle import train as otrain

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _task(b, hw, seed):
    x, m = otrain.synthetic_batch(b, hw, seed=seed)
    return 0.6 * x + m * torch.tensor([1.0, -0.7, 0.4]).view(1, 3, 1, 1), m


def _dice(logit, m):
    p = (torch.sigmoid(logit) > 0.5).float()
    t = (m > 0.5).float()
    return float((2 * (p * t).sum() + 1e-7) / (p.sum() + t.sum() + 1e-7))


@pytest.mark.parametrize("name,steps,dtypes", [
    ("AttentionUNet", 32, (torch.float32, torch.bfloat16, torch.float16)),          # C3 (bf16) and C5's segmenter (fp16)
    ("R2AttU_Net", 32, (torch.float32, torch.bfloat16)),                           # C4 (bf16)
])
def test_dice_after_training_matches_oracle(name, steps, dtypes):
    from mi355 import nn as mnn, optim as moptim, amp as mamp
    from utils.helpers import get_seg_model
    hw, b, lr = 64, 4, 1e-3
    batches = [_task(b, hw, s) for s in range(4)]
    xv, mv = _task(32, hw, 99)
    sd0 = nets.default_init_state(name, seed=0)
    fwd = nets.NETS[name]

    sd = {k: v.clone() for k, v in sd0.items()}
    opt = otrain.AdamW(nets.param_keys(sd), lr)
    for i in range(steps):
        x, y = batches[i % 4]
        ref_loss, _, _ = otrain.train_step(name, sd, x, y, opt, True)
    with torch.no_grad():
        ref_dice = _dice(fwd({k: v.clone() for k, v in sd.items()}, xv, True), mv)
    assert ref_dice > 0.95          # the task is learnable: the comparison is not degenerate
    strict = name == "AttentionUNet"
    loss_tol = 0.02 if strict else 0.15          # (loss: two CPU runs of the R2 protocol are 8-20 % apart, module docstring)
    # Dice: 1e-3, except a bf16 TRAJECTORY of R2AttU_Net — equally valid bf16 arithmetic is 2.5e-3 apart there (module docstring)
    dice_tols = {dt: (3e-3 if (not strict and dt == torch.bfloat16) else 1e-3) for dt in dtypes}

    # the oracle-trained weights through the HIP forward: Dice within 1e-3 in every precision (no trajectory involved)
    for dtype in dtypes:
        m = get_seg_model({"AttentionUNet": "attentionunet", "R2AttU_Net": "r2attunet"}[name])
        m.load_state_dict(sd)
        m.compute_dtype = dtype
        m = m.to(DEV).train()
        with torch.no_grad():
            d = _dice(m(xv.to(DEV)).float().cpu(), mv)
        assert abs(d - ref_dice) <= 1e-3, ("oracle weights", name, str(dtype), d, ref_dice)
        del m

    for dtype in dtypes:
        m = get_seg_model({"AttentionUNet": "attentionunet", "R2AttU_Net": "r2attunet"}[name])
        m.load_state_dict(sd0)
        m.compute_dtype = dtype
        m = m.to(DEV).train()
        o = moptim.AdamW(m.parameters(), lr=lr, weight_decay=5e-4)
        crit = mnn.BCEWithLogitsLoss()
        # the reference's sequence (helpers.py:320-336); the scaler is a no-op unless fp16.  A small growth interval
        # makes the scale move inside the 32 steps, and the first steps overflow (d loss / d logit = 2^34 / 16384 > 65504)
        scaler = mamp.GradScaler(init_scale=2.0 ** 34, growth_interval=8, enabled=dtype == torch.float16)
        done = it = 0
        while done < steps and it < steps + 24:
            x, y = batches[done % 4]                     # a skipped (overflowed) step is repeated on the same batch
            o.zero_grad(set_to_none=True)
            loss = crit(m(x.to(DEV)), y.to(DEV))
            scaler.scale(loss).backward()
            scaler.unscale_(o)
            moptim.clip_grad_norm_(m.parameters(), 1.0)
            scaler.step(o)
            scaler.update()
            it += 1
            done = int(o._st[0]["step"])
        assert done == steps
        if dtype == torch.float16:
            assert it > steps and scaler.get_scale() < 2.0 ** 34       # overflow steps were skipped, the scale backed off
        else:
            assert it == steps
        with torch.no_grad():
            d = _dice(m(xv.to(DEV)).float().cpu(), mv)       # train-mode BN, like the oracle evaluation above
        assert abs(d - ref_dice) <= dice_tols[dtype], (name, str(dtype), d, ref_dice)
        assert abs(float(loss.detach()) - ref_loss) <= loss_tol * ref_loss, (name, str(dtype), float(loss.detach()), ref_loss)
