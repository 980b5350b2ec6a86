"""CPU: the kink record / replay of the oracle (oracle/nets.py::Kinks).  The gradients of these ReLU / max-pool networks are
discontinuous: fp32 and fp64 evaluations disagree on a handful of ReLU masks (pre-activations within rounding distance of zero)
and therefore on every gradient by ~1e-3, whatever the implementation.  With the fp32 evaluation's masks replayed inside the
fp64 evaluation both compute the same smooth function and agree to rounding accuracy — the basis of the sharp whole-model
gradient checks of tests/test_gpu_kinks.py."""
import numpy as np
import pytest
import torch

from oracle import nets
from oracle import train as otrain


def _grads(name, sd, x, y, dtype, mode=None, kinks=None):
    s = {k: (v.to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    if mode:
        nets.Kinks.start(mode, *(kinks or ()))
    try:
        loss, out, g = otrain.forward_backward(name, s, x.to(dtype), y.to(dtype), True)
    finally:
        relu, pool, used = nets.Kinks.stop()
    return loss, out, g, (relu, pool), used


def _errs(g, ref):
    gmax = max(float(v.abs().max()) for v in ref.values())
    out = []
    for k, r in ref.items():
        sc = float(r.abs().max())
        if sc >= 1e-6 * gmax:
            out.append(float((g[k].double() - r).abs().max()) / sc)
    return np.array(out)


def test_replayed_masks_make_fp32_and_fp64_gradients_agree():
    name = "AttentionUNet"
    sd = nets.closed_form_state(name)
    x, y = otrain.closed_form_input(2, 32)
    _, o32, g32, kinks, _ = _grads(name, sd, x, y, torch.float32, "record")
    assert len(kinks[0]) == 26 and len(kinks[1]) == 4            # 18 block + 4 up-conv + 4 gate ReLUs, 4 max-pools
    _, o64, g64, _, _ = _grads(name, sd, x, y, torch.float64)
    _, o64r, g64r, _, used = _grads(name, sd, x, y, torch.float64, "replay", kinks)
    assert used == (26, 4)
    plain, replay = _errs(g32, g64), _errs(g32, g64r)
    assert np.median(replay) < 3e-5 and replay.max() < 3e-4, (np.median(replay), replay.max())
    assert np.median(replay) <= np.median(plain)                 # (plain: 4e-4 on this fixture — a few flipped masks)
    assert float((o32.double() - o64r).abs().max() / o64r.abs().max()) < 1e-4
    # replaying an evaluation's own masks reproduces it
    _, _, g64s, kinks64, _ = _grads(name, sd, x, y, torch.float64, "record")
    _, _, g64t, _, _ = _grads(name, sd, x, y, torch.float64, "replay", kinks64)
    assert _errs(g64t, g64s).max() < 1e-12


def test_r2attunet_training_trajectory_is_chaotic_on_the_cpu_alone():
    """CPU-only control for the trained-Dice criterion of config C4 (tests/test_gpu_train_bf16.py[R2AttU_Net]): the SAME
    protocol (64 x 64, batch 4, lr 1e-3, default init) run three ways on the oracle that differ by nothing but summation order /
    precision — fp32, fp32 with every batch's images in reversed order, fp64.  Eight steps in, their Dice values on the held-out
    images are already further apart than the 1e-3 the north-star allows between the HIP path and the reference (full protocol,
    tests/diag/diag_r2_chaos.py: 3.9e-3 at step 8, 1.9e-3 at 12, 2.4e-4 at 20, 4.1e-4 at 32; losses 8-20 % apart throughout): a
    1e-3 bound on a HIP-TRAINED run before the plateau would measure chaos, not arithmetic — which is why the GPU test trains to
    step 32.  No GPU is involved here."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("diag_r2_chaos", os.path.join(os.path.dirname(__file__), "diag", "diag_r2_chaos.py"))
    chaos = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chaos)
    marks = [8]
    runs = {"fp32": chaos.run("R2AttU_Net", marks), "fp32 reversed": chaos.run("R2AttU_Net", marks, flip=True),
            "fp64": chaos.run("R2AttU_Net", marks, dtype=torch.float64)}
    dice = np.array([r[8][0] for r in runs.values()])
    loss = np.array([r[8][1] for r in runs.values()])
    print("R2AttU_Net CPU control, step 8: Dice", dict(zip(runs, dice.round(5))), "loss", dict(zip(runs, loss.round(5))))
    assert dice.min() > 0.95                                    # every run learns the task
    assert dice.max() - dice.min() > 5e-4, dice                 # ... and they have visibly parted (measured 2.7e-3 among these three)


def test_trajectory_fixture_learning_rate_control():
    """CPU-only control for the learning rate of the recurrent nets' trajectory fixtures (tests/golden/train_traj_R2AttU_Net.npz:
    the reference's own train(), helpers.py:231-412, 3 epochs x 2 batches at 64 x 64, lr 1e-5; round 3 had generated it at 1e-4 /
    32 x 32 first and the HIP run missed its validation loss by 1.5 %).  The SAME protocol is evaluated twice on the oracle — all
    cores and ONE thread, i.e. the same function with fp32 sums in another order — at lr 1e-4 and at lr 1e-5.  Adam turns the
    SIGN of noise-level gradients into full +-lr steps, so the two evaluations part in proportion to lr: measured here (8 cores),
    epoch-1 validation loss 3.4e-3 apart at 1e-4 (23.275 / 23.196) and 6.4e-4 apart at 1e-5 (22.2707 / 22.2565); the bottleneck's
    running mean likewise.  Asserted: the 1e-5 pair agrees to the 2e-3 that tests/test_oracle_pins.py::test_train_trajectory_seg
    allows between the oracle and the reference's log, and the 1e-4 pair is at least twice as far apart."""
    name, hw = "R2AttU_Net", 64
    b = [otrain.synthetic_batch(4, hw, seed=s) for s in (0, 1, 2)]
    n_threads = torch.get_num_threads()
    gaps = {}
    try:
        for lr in (1e-4, 1e-5):
            vals = []
            for threads in (n_threads, 1):
                torch.set_num_threads(threads)
                sd = nets.closed_form_state(name)
                _, hist = otrain.train_seg(name, sd, b[:2], b[2:], 1, lr)
                vals.append(hist[0][1])                  # epoch-1 validation loss
            gaps[lr] = abs(vals[0] - vals[1]) / abs(vals[0])
    finally:
        torch.set_num_threads(n_threads)
    print(f"R2AttU_Net trajectory control ({n_threads} threads vs 1): relative gap of the epoch-1 validation loss:", gaps)
    if n_threads == 1:
        pytest.skip("one core: the two evaluations are the same arithmetic")
    assert gaps[1e-5] <= 2e-3, gaps
    assert gaps[1e-4] >= 2 * gaps[1e-5], gaps
