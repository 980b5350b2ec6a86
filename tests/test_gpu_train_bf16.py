"""-m gpu: the north-star Dice criterion (configs C3 and C4 in their own dtypes).  Train Attention U-Net for 32 optimiser steps on a learnable
synthetic task (ellipse visible in the image) three ways from identical weights and batches —
HIP bf16, HIP fp16 (+ loss scaling, helpers.py:285,323-336), HIP fp32, CPU fp32 oracle (reference semantics:
BCEWithLogits, clip 1.0, AdamW wd 5e-4) —
and compare the Dice of the binarised predictions (tester.py:114-134) on 32 held-out images.
Bound: |Dice - Dice_oracle| <= 1e-3 (0..1 scale) for every GPU mode; final losses within 2 %.

R2AttU_Net (R2AttU_Net.py:88-158, config C4: bf16) is held to the SAME 1e-3 in every precision since round 4, under a criterion that
is well conditioned for it.  Round 3 read Dice on 32 held-out images after 32 steps at lr 1e-3 and had to allow a bf16 trajectory
3e-3: equally valid bf16 arithmetic (the plan-level fusion switches move roundings around, tests/diag/diag_r2_bf16_spread.py) was
2.5e-3 apart there, and the fp32 curve itself moved 3e-3 between marks — with 108 shared-weight convolutions per forward the
trajectory is chaotic on the CPU alone before the plateau (tests/diag/diag_r2_chaos.py: four CPU runs that differ only in summation
order are 3.9e-3 apart at step 8, 4e-4 at step 32), and at lr 1e-3 a 32-image Dice measured the step index, not arithmetic.  Now: 32
steps at lr 1e-3, then 16 at lr 1e-4 (the plateau is allowed to settle, as the reference's schedulers do at the end of training,
helpers.py:254,307-309), Dice over 256 held-out images (eight batches of 32, counts pooled), averaged over the checkpoints at steps
44 and 48.  Measured on the HIP path (profiles/r04b_r2_conditioned.txt): fp32 moves 4.5e-4 between marks (was 3e-3), the four bf16
roundings are 0.7e-3 apart at step 48 (was 2.5e-3), the default bf16 plan is 6e-5 ... 1.7e-4 from the fp32 plan at steps 40-64.
Losses: two CPU runs of this protocol are 8-20 % apart before the decay (module history), hence 15 %.  The ORACLE-TRAINED weights
evaluated by the HIP forward must give the oracle's Dice within 1e-3 in every precision as before (no trajectory involved)."""
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


def _pooled_dice(forward, held):
    """Dice of the binarised predictions over several held-out batches, intersection and mask counts pooled (train-mode BatchNorm
    per batch of 32, as the single-batch evaluation above)."""
    num = den = 0.0
    for xv, mv in held:
        p = (torch.sigmoid(forward(xv)) > 0.5).double()
        t = (mv > 0.5).double()
        num += float(2 * (p * t).sum())
        den += float(p.sum() + t.sum())
    return (num + 1e-7) / (den + 1e-7)


# name, steps, precisions, step after which lr drops to 1e-4 (0: never), held-out batches of 32, checkpoints whose Dice is averaged
@pytest.mark.parametrize("name,steps,dtypes,decay_at,n_held,marks", [
    ("AttentionUNet", 32, (torch.float32, torch.bfloat16, torch.float16), 0, 1, (32,)),      # C3 (bf16) and C5's segmenter (fp16)
    ("R2AttU_Net", 48, (torch.float32, torch.bfloat16), 32, 8, (44, 48)),                     # C4 (bf16): the conditioned criterion
])
def test_dice_after_training_matches_oracle(name, steps, dtypes, decay_at, n_held, marks):
    from mi355 import nn as mnn, optim as moptim, amp as mamp
    from utils.helpers import get_seg_model
    hw, b, lr, lr2 = 64, 4, 1e-3, 1e-4
    batches = [_task(b, hw, s) for s in range(4)]
    held = [_task(32, hw, 99 + c) for c in range(n_held)]
    sd0 = nets.default_init_state(name, seed=0)
    fwd = nets.NETS[name]

    sd = {k: v.clone() for k, v in sd0.items()}
    opt = otrain.AdamW(nets.param_keys(sd), lr)
    ref_marks = []
    for i in range(steps):
        if decay_at and i == decay_at:
            opt.lr = lr2
        x, y = batches[i % 4]
        ref_loss, _, _ = otrain.train_step(name, sd, x, y, opt, True)
        if i + 1 in marks:
            with torch.no_grad():
                ref_marks.append(_pooled_dice(lambda xv: fwd({k: v.clone() for k, v in sd.items()}, xv, True), held))
    ref_dice = sum(ref_marks) / len(ref_marks)
    assert ref_dice > 0.95          # the task is learnable: the comparison is not degenerate
    strict = name == "AttentionUNet"
    loss_tol = 0.02 if strict else 0.15          # (loss: two CPU runs of the R2 protocol are 8-20 % apart, module docstring)
    dice_tols = {dt: 1e-3 for dt in dtypes}      # the north star's bound, every model, every precision

    # the oracle-trained weights through the HIP forward: Dice within 1e-3 in every precision (no trajectory involved)
    for dtype in dtypes:
        m = get_seg_model({"AttentionUNet": "attentionunet", "R2AttU_Net": "r2attunet"}[name])
        m.load_state_dict(sd)
        m.compute_dtype = dtype
        m = m.to(DEV).train()
        with torch.no_grad():
            d = _pooled_dice(lambda xv: m(xv.to(DEV)).float().cpu(), held)
        assert abs(d - ref_marks[-1]) <= 1e-3, ("oracle weights", name, str(dtype), d, ref_marks[-1])
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
        got_marks, seen_marks = [], set()
        while done < steps and it < steps + 24:
            x, y = batches[done % 4]                     # a skipped (overflowed) step is repeated on the same batch
            if decay_at and done == decay_at:
                for gr in o.param_groups:
                    gr["lr"] = lr2
            o.zero_grad(set_to_none=True)
            loss = crit(m(x.to(DEV)), y.to(DEV))
            scaler.scale(loss).backward()
            scaler.unscale_(o)
            moptim.clip_grad_norm_(m.parameters(), 1.0)
            scaler.step(o)
            scaler.update()
            it += 1
            done = int(o._st[0]["step"])
            if done in marks and done not in seen_marks:
                seen_marks.add(done)
                with torch.no_grad():                    # train-mode BN, like the oracle evaluation above
                    got_marks.append(_pooled_dice(lambda xv: m(xv.to(DEV)).float().cpu(), held))
        assert done == steps and len(got_marks) == len(marks)
        if dtype == torch.float16:
            assert it > steps and scaler.get_scale() < 2.0 ** 34       # overflow steps were skipped, the scale backed off
        else:
            assert it == steps
        d = sum(got_marks) / len(got_marks)
        print(f"[{name} {dtype}] Dice {d:.5f} (oracle {ref_dice:.5f}), checkpoints {[round(v, 5) for v in got_marks]} vs {[round(v, 5) for v in ref_marks]}")
        assert abs(d - ref_dice) <= dice_tols[dtype], (name, str(dtype), d, ref_dice, got_marks, ref_marks)
        assert abs(float(loss.detach()) - ref_loss) <= loss_tol * ref_loss, (name, str(dtype), float(loss.detach()), ref_loss)
