"""Attention-gate streaming kernels at the four levels of the headline configuration (bs 32, 256 x 256): ms and algorithmic GB/s of
mi355_gate_psi_fwd, mi355_gate_bn_bwd_reduce, mi355_gate_bn_bwd_apply (HIP events, 20 launches after 3 warm-ups).
  python scripts/gate_bench.py            (A/B: MI355_LIB=ab/x.so, MI355_RR_WGS=512)"""
import os
import sys

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(R, "medical-image-segmentation-and-classification_amd"))
import torch
from mi355.lib import lib

dev = "cuda:0"
dt, code, es = torch.bfloat16, 1, 2
for M, C in ((2097152, 32), (524288, 64), (131072, 128), (32768, 256)):
    g1 = torch.randn(M, C, device=dev).to(dt); x1 = torch.randn(M, C, device=dev).to(dt)
    co = [torch.rand(C, device=dev) + 0.5 if i % 2 == 0 else torch.randn(C, device=dev) * 0.1 for i in range(8)]
    w = torch.randn(C, device=dev); b = torch.zeros(1, device=dev)
    z = torch.empty(M, device=dev); dz = torch.randn(M, device=dev)
    nb = lib.mi355_rowreduce_blocks(M)
    zp = torch.empty(nb * 2, device=dev); part = torch.empty(nb * 5 * C, device=dev)
    sums = [torch.randn(2 * C, device=dev) for _ in range(2)]
    dg = torch.empty(M, C, device=dev, dtype=dt); dx = torch.empty(M, C, device=dev, dtype=dt)
    runs = {
        "gate_psi_fwd": (lambda: lib.mi355_gate_psi_fwd(g1, C, x1, C, co[0], co[1], co[4], co[5], w, b, z, zp, M, C, code), 2 * M * C * es + 4 * M),
        "gate_bn_bwd_reduce": (lambda: lib.mi355_gate_bn_bwd_reduce(dz, g1, C, x1, C, *co, w, part, M, C, code), 2 * M * C * es + 4 * M),
        "gate_bn_bwd_apply": (lambda: lib.mi355_gate_bn_bwd_apply(dz, g1, C, x1, C, *co, w, co[0], co[4], sums[0], sums[1], dg, C, dx, C, M, C, code),
                              4 * M * C * es + 4 * M),
    }
    Cx = 2 * C         # the gated skip tensor has twice the gate's inner channels
    xs = torch.randn(M, Cx, device=dev).to(dt); dys = torch.randn(M, Cx, device=dev).to(dt); dxs = torch.empty(M, Cx, device=dev, dtype=dt)
    dzn = torch.empty(M, device=dev); p2 = torch.empty(nb * 2, device=dev); one = torch.ones(1, device=dev); zero = torch.zeros(1, device=dev)
    runs["gate_mul_bwd"] = (lambda: lib.mi355_gate_mul_bwd(dys, Cx, xs, Cx, z, one, zero, zero, one, dxs, Cx, 0, dzn, p2, M, Cx, code), 3 * M * Cx * es)
    if C == 64:        # the logit head's one-channel convolution (AttentionUNet.py:84) at the full resolution
        Mh = 2097152
        xh = torch.randn(Mh, C, device=dev).to(dt); zh = torch.empty(Mh, device=dev)
        runs["rowdot_fwd (head, M=2097152)"] = (lambda: lib.mi355_rowdot_fwd(xh, C, w, b, zh, None, Mh, C, 0, 1, code), Mh * C * es + 4 * Mh)
    for name, (fn, nbytes) in runs.items():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"M={M:8d} C={C:4d} {name:20s} {ms * 1e3:7.1f} us  {nbytes / ms / 1e6:7.0f} GB/s", flush=True)
