"""The wide Linear layers of the torchvision VGG head (25088 -> 4096 -> 4096, batch 16) through the C ABI: ms and GB/s of the weight
matrix (fp32, read once forward, read once + written once backward).   python scripts/linear_bench.py   (A/B: MI355_LIB=ab/x.so)"""
import os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(R, "medical-image-segmentation-and-classification_amd"))
import torch
from mi355.lib import lib

dev = "cuda:0"
for B, I, O in ((16, 25088, 4096), (16, 4096, 4096), (8, 25088, 4096)):
    x = torch.randn(B, I, device=dev); w = torch.randn(O, I, device=dev) * 0.01; b = torch.randn(O, device=dev)
    y = torch.empty(B, O, device=dev); dy = torch.randn(B, O, device=dev)
    dx = torch.empty(B, I, device=dev); dw = torch.empty(O, I, device=dev); db = torch.empty(O, device=dev)
    scratch = torch.empty(max(1, lib.mi355_linear_bwd_scratch(B, I, O)), device=dev)
    runs = {"linear_fwd": (lambda: lib.mi355_linear_fwd(x, w, b, y, B, I, O, 1), O * I * 4),
            "linear_bwd (dx + dW)": (lambda: lib.mi355_linear_bwd(x, w, y, dy, dx, dw, db, B, I, O, 1, 0.0, scratch), 2 * O * I * 4),
            "linear_bwd (dW only)": (lambda: lib.mi355_linear_bwd(x, w, y, dy, None, dw, db, B, I, O, 1, 0.0, scratch), O * I * 4)}
    for name, (fn, nbytes) in runs.items():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"B={B:3d} I={I:6d} O={O:5d} {name:22s} {ms * 1e3:8.1f} us  {nbytes / ms / 1e6:7.0f} GB/s", flush=True)
