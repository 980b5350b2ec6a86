"""Time single 3x3 convolution launches through the C ABI (HIP events, median of `reps`): python scripts/conv_bench.py [N,H,W,Ci,Co[,up[,w]]] ...
(a seventh field `1` times the WEIGHT GRADIENT of that layer instead).  Default shapes: the Ci <= 128 layers of AttentionUNet at batch 32 (forward layout; a data gradient of Ci -> Co is the forward of Co -> Ci)."""
import os
import sys

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(R, "medical-image-segmentation-and-classification_amd"))
import torch
from mi355.lib import lib, DTYPE_CODE

shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:] if "," in a] or [
    (32, 256, 256, 64, 64), (32, 128, 128, 64, 128), (32, 256, 256, 64, 128), (32, 256, 256, 128, 64), (32, 128, 128, 128, 128),
    (32, 128, 128, 128, 256), (32, 64, 64, 128, 256), (16, 256, 256, 64, 64)]
reps = 30
dt = torch.bfloat16
code = DTYPE_CODE[dt]
for shp in shapes:
    n, h, w, ci, co = shp[:5]
    up = shp[5] if len(shp) > 5 else 0
    wgrad = len(shp) > 6 and shp[6]
    ho, wo = (2 * h, 2 * w) if up else (h, w)
    x = torch.randn(n, h, w, ci, device="cuda").to(dt)
    wk = (torch.randn(co, 9, ci, device="cuda") / (9 * ci) ** 0.5).to(dt)
    y = torch.empty(n, ho, wo, co, device="cuda", dtype=dt)
    rows = lib.mi355_conv2d_igemm_stat_rows(n, h, w, ci, ho, wo, co, 3, 3, 1, 1, -1, 1, up, code)
    part = torch.empty(max(rows, 1) * 2 * co, device="cuda")
    var = lib.mi355_conv2d_igemm_variant_n(n, h, w, ci, ho, wo, co, 3, 3, 1, 1, -1, 1, up, code)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    if wgrad:
        sp = lib.mi355_conv2d_wgrad_splits(n, ho, wo, ci, co, 3, 3)
        ws = torch.empty(sp, co, 9, ci, device="cuda")
        var = f"wgrad splits {sp}"
    for i in range(reps + 3):
        if i >= 3:
            ev[i - 3][0].record()
        if wgrad:
            lib.mi355_conv2d_wgrad(x, y, ws, sp, n, h, w, ci, ci, ho, wo, co, co, 3, 3, 1, 1, up, code)
        else:
            lib.mi355_conv2d_igemm(x, wk, None, y, n, h, w, ci, ci, ho, wo, co, co, 3, 3, 1, 1, -1, 1, up, 0, part if rows else None, code)
        if i >= 3:
            ev[i - 3][1].record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    fl = 2.0 * n * ho * wo * co * 9 * ci
    gb = (n * h * w * ci + n * ho * wo * co) * 2 / 1e9
    print(f"{n}x{h}x{w}x{ci}->{co} up{up} variant {var}: median {ms[reps // 2]:.4f} ms  min {ms[0]:.4f}  {fl / ms[reps // 2] / 1e9:7.1f} TFLOP/s  {gb / ms[reps // 2] * 1e3:6.0f} GB/s algorithmic", flush=True)
