"""-m gpu: numerics at BENCHMARK scale — Attention U-Net 256x256, batch 32 (config C3 on one GPU).  Every other model test runs
32x32..64x64 images; at M = 32*256*256 = 2 097 152 pixel rows the launch plan takes paths small shapes never reach (thousands
of per-tile BatchNorm statistic rows pre-folded by mi355_fold_rows, the XCD-aware tile order over 16 384 tiles, 32-way split
weight-gradient reductions, two-image 16x16 weight-gradient tiles at N = 32).  Checks:
  * eval forward (BatchNorm uses running statistics, so samples are independent): images 0 and 31 of the batch against the CPU
    oracle run on those two images alone — fp32 HIP within 1e-3, bf16 within bf16 rounding;
  * train step: bf16 against fp32 on the HIP path (logits, loss, masks, per-tensor gradient norms, gradient direction);
  * kernel A/B at full size: the nine-tap weight-gradient kernel against the generic split-K kernel (MI355_WGRAD_HALO=0) and the
    halo / streaming forward + data-gradient kernels against the generic implicit GEMM (MI355_IGEMM_VARIANT=0)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run(tmp_path, tag, dtype, **env):
    out = str(tmp_path / f"{tag}.npz")
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, os.path.join(HERE, "bench_scale_worker.py"), dtype, out], env=e, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    return np.load(out)


def _l2rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def _cos(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b)))


def test_attention_unet_256_batch32(tmp_path):
    from oracle import nets
    sys.path.insert(0, HERE)
    import bench_scale_worker as w
    A = _run(tmp_path, "bf16", "bf16")
    F = _run(tmp_path, "fp32", "fp32")
    assert bool(A["finite"]) and bool(F["finite"])
    assert "conv3x3_halo_rw_kernel<8,32>" in set(A["tags"]) and "wgrad3x3_halo_kernel" in set(A["tags"])

    # ---- eval forward vs the CPU oracle on two of the 32 images -------------------------------------------------------
    import bench
    x, _ = bench.make_batch(32, 256, seed=0, device="cpu")
    sd = w.he_state()
    torch.set_num_threads(max(1, len(os.sched_getaffinity(0))))
    with torch.no_grad():
        ref = {i: nets.attention_unet({k: v.clone() for k, v in sd.items()}, x[i:i + 1], False)[0].numpy() for i in (0, 31)}
    for key, i in (("eval_first", 0), ("eval_last", 31)):
        r = ref[i]
        assert np.abs(F[key] - r).max() <= 1e-3 * np.abs(r).max(), key                    # fp32: north-star bound
        assert _l2rel(A[key], r) <= 3e-2, (key, _l2rel(A[key], r))                        # bf16: 22 stacked bf16 layers
        agree = ((A[key] > 0) == (r > 0)).mean()
        assert agree >= 0.99, (key, agree)

    # ---- train step: bf16 vs fp32, both on the HIP path ------------------------------------------------------------------
    assert abs(float(A["loss"]) - float(F["loss"])) <= 1e-2 * float(F["loss"])
    assert _l2rel(A["logits"], F["logits"]) <= 5e-2
    pa, pf = A["logits"] > 0, F["logits"] > 0
    dice = 2.0 * (pa & pf).sum() / max(pa.sum() + pf.sum(), 1)
    assert dice >= 0.99 or (pa != pf).mean() <= 5e-3, dice
    big = F["grad_norm"] > 1e-4 * F["grad_norm"].max()
    ratio = A["grad_norm"][big] / F["grad_norm"][big]
    assert np.median(np.abs(ratio - 1)) <= 3e-2 and np.abs(ratio - 1).max() <= 0.3, (np.median(np.abs(ratio - 1)), np.abs(ratio - 1).max())
    assert _cos(A["grad_sample"], F["grad_sample"]) >= 0.99
    assert abs(float(A["grad_total"]) / float(F["grad_total"]) - 1) <= 3e-2

    # ---- kernel A/B at full size (same bf16 inputs, different kernels: only the order of the fp32 sums differs) --------------
    B = _run(tmp_path, "wgrad_generic", "bf16", MI355_WGRAD_HALO="0")
    assert "wgrad3x3_halo_kernel" not in set(B["tags"])
    assert np.array_equal(A["logits"], B["logits"])                                          # forward untouched
    r = B["grad_norm"][big] / A["grad_norm"][big]
    assert np.abs(r - 1).max() <= 2e-3, np.abs(r - 1).max()
    assert _l2rel(B["grad_sample"], A["grad_sample"]) <= 2e-3
    C = _run(tmp_path, "igemm_generic", "bf16", MI355_IGEMM_VARIANT="0")
    assert not any(t.startswith("conv3x3_halo") for t in C["tags"])
    assert _l2rel(C["eval_first"], A["eval_first"]) <= 1e-2 and _l2rel(C["logits"], A["logits"]) <= 3e-2
    assert _cos(C["grad_sample"], A["grad_sample"]) >= 0.995
    r = C["grad_norm"][big] / A["grad_norm"][big]
    assert np.median(np.abs(r - 1)) <= 2e-2, np.median(np.abs(r - 1))
