"""-m gpu: the plan-level fusions (MaxPool2d and its gradient inside BatchNorm passes, the attention gate and the logit head
without their stored intermediate activations, RRCNN_block's residual inside a BatchNorm apply pass — DESIGN.md section 4,
"Round 3"; round 4: the recurrent block's d x summed once by its last application's pass) compute the SAME training step as the separate launches.  The switches are module globals of mi355.graph read when
a plan is built, so one process builds both plans: one fp32 step (forward, BCE, backward) with all fusions on against each
switch off alone and against all off.

What "the same" means: a fused pass may sum a dot product with fused multiply-adds where the separate kernel does not, so a
forward can differ in the last bits; an activation within an ulp of zero then falls on the other side of a ReLU (or two equal
neighbours swap in a max-pool), and the gradient — a sum over a few hundred pixels at the deep levels of a 64 x 64 input —
changes by that pixel's share (tests/diag/diag_fusion_switches.py: the gate switch flips 4 of 10^7 decisions and moves one
BatchNorm gain by 3 %).  The test therefore reads both plans' decisions back (Plan.acts, as tests/test_gpu_kinks.py does): where
they are IDENTICAL every gradient must agree to summation-order rounding; where a few differ, to the size of those pixels' share."""
import numpy as np
import pytest
import torch

from gpu_util import DEV, gpu_kinks
from oracle import nets, train as otrain

pytestmark = pytest.mark.gpu
SW = ("FUSE_POOL", "FUSE_POOL_BWD", "FUSE_GATE_BWD", "FUSE_HEAD", "FUSE_RESIDUAL", "DEFER_POST")


def _step(name, off):
    from mi355 import graph, nn as mnn
    from utils.helpers import get_seg_model
    saved = {s: getattr(graph, s) for s in SW}
    try:
        for s in SW:
            setattr(graph, s, s not in off)
        graph.Builder.fuse_residual = graph.FUSE_RESIDUAL
        x, y = otrain.synthetic_batch(3, 64, seed=5)
        m = get_seg_model({"AttentionUNet": "attentionunet", "R2AttU_Net": "r2attunet"}[name])
        m.load_state_dict(nets.default_init_state(name, seed=1))
        m.compute_dtype = torch.float32
        m = m.to(DEV).train()
        out = m(x.to(DEV))
        loss = mnn.BCEWithLogitsLoss()(out, y.to(DEV))
        loss.backward()
        torch.cuda.synchronize()
        plan = out._mi355_plan
        relu, pool = gpu_kinks(plan)
        return {"logits": out.detach().cpu().double(), "loss": float(loss.detach()), "relu": relu, "pool": pool,
                "grads": {k: p.grad.double().cpu() for k, p in m.named_parameters()},
                "bufs": {k: b.double().cpu() for k, b in m.named_buffers()},
                "names": {l.name for l in plan.fwd + plan.bwd}, "launches": len(plan.fwd) + len(plan.bwd)}
    finally:
        for s, v in saved.items():
            setattr(graph, s, v)
        graph.Builder.fuse_residual = graph.FUSE_RESIDUAL


# (flip cap, logits, loss): the recurrent network carries a last-bit difference of the first gate through 100 more shared-weight
# convolutions — 25 x the logit difference of the feed-forward one, and two orders of magnitude more decisions near zero
LIMITS = {"AttentionUNet": (32, 1e-4, 1e-5), "R2AttU_Net": (2000, 2e-3, 2e-4)}


def _compare(on, other, what, name):
    cap, tol_logits, tol_loss = LIMITS[name]
    flips = sum(int((a != b).sum()) for a, b in zip(on["relu"], other["relu"])) + \
        sum(int((a != b).sum()) for a, b in zip(on["pool"], other["pool"]))
    assert flips <= cap, (what, flips)                    # of ~10^7 decisions
    lo = float((on["logits"] - other["logits"]).abs().max() / other["logits"].abs().max())
    assert lo < tol_logits and abs(on["loss"] - other["loss"]) < tol_loss, (what, lo)
    gmax = max(float(g.abs().max()) for g in other["grads"].values())
    worst_max, worst_l2 = (0.0, ""), (0.0, "")
    for k, g in other["grads"].items():
        sc = float(g.abs().max())
        if sc < 1e-5 * gmax:          # conv bias in front of a train-mode BatchNorm: exactly zero, both plans compute round-off
            assert float(on["grads"][k].abs().max()) <= 1e-4 * gmax, (what, k)
            continue
        d = on["grads"][k] - g
        worst_max = max(worst_max, (float(d.abs().max()) / sc, k))
        worst_l2 = max(worst_l2, (float(d.norm() / g.norm()), k))
    for k, b in other["bufs"].items():
        assert float((on["bufs"][k] - b).abs().max()) <= 10 * tol_logits * max(1.0, float(b.abs().max())), (what, k)
    if flips == 0:
        assert worst_max[0] < 2e-4, (what, worst_max)       # summation order only
    else:
        assert worst_l2[0] < (6e-2 if name == "AttentionUNet" else 0.5), (what, flips, worst_max, worst_l2)
    return flips, worst_max[0]


@pytest.mark.parametrize("name", ["AttentionUNet", "R2AttU_Net"])
def test_fused_plan_computes_the_same_step(name):
    on = _step(name, ())
    off = _step(name, SW)
    # the switches do switch: no scatter pass of the pooling gradient, no stored psi_in / head activation, fewer launches
    assert "mi355_rowdot_bwd" in off["names"] and "mi355_maxpool_bwd" in off["names"] and "mi355_gate_psi_fwd" not in off["names"]
    assert "mi355_gate_psi_fwd" in on["names"] and "mi355_gate_bn_bwd_apply" in on["names"] and on["launches"] < off["launches"]
    if name == "AttentionUNet":
        assert "mi355_bn_bwd_apply_pool2" in on["names"] and not ({"mi355_maxpool_bwd", "mi355_rowdot_bwd", "mi355_rowdot_fwd"} & on["names"])
    _compare(on, off, "all off", name)
    for s in SW:
        one = _step(name, (s,))
        flips, worst = _compare(on, one, s, name)
        if s == "FUSE_POOL" or (s == "FUSE_RESIDUAL" and name == "AttentionUNet"):      # these keep the order of every sum: bit-identical
            assert flips == 0 and worst == 0.0 and torch.equal(on["logits"], one["logits"]), s
        if s in ("FUSE_RESIDUAL", "DEFER_POST"):
            # forward untouched.  Backward: the recurrent block's d x is ONE fp32 sum of its applications' incoming gradients
            # (mi355_bn_bwd_apply_post4) instead of a chain of read-modify-writes, and RRCNN_block's residual x0 is one more
            # addend of the first block's sum when it is fused — the same terms in another order
            assert flips == 0 and torch.equal(on["logits"], one["logits"]) and worst < 2e-4, (s, flips, worst)
            if name == "R2AttU_Net" and s == "DEFER_POST":
                assert "mi355_bn_bwd_apply_post4" in on["names"] and "mi355_bn_bwd_apply_post4" not in one["names"]
