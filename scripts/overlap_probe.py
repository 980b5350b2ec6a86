"""Do an MFMA-bound wgrad kernel and an HBM-bound BatchNorm-backward kernel overlap when launched on two streams?
Prints t(wgrad), t(bn), t(both, two streams)."""
import sys, os, time
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'medical-image-segmentation-and-classification_amd'))
import torch
from mi355.lib import lib, DTYPE_CODE
dev = "cuda:0"; dt = torch.bfloat16; code = DTYPE_CODE[dt]
N, H, W, C = 32, 256, 256, 64
M = N * H * W
x = torch.randn(N, H, W, C, device=dev).to(dt); dy = torch.randn(N, H, W, C, device=dev).to(dt)
splits = lib.mi355_conv2d_wgrad_splits(N, H, W, C, C, 3, 3)
ws = torch.empty(splits * C * 9 * C, device=dev)
y = torch.randn(N, H, W, C, device=dev).to(dt); g = torch.randn(N, H, W, C, device=dev).to(dt); dx = torch.empty_like(g)
gamma = torch.ones(C, device=dev); mean = torch.zeros(C, device=dev); inv = torch.ones(C, device=dev)
sc = torch.ones(C, device=dev); sh = torch.zeros(C, device=dev); sums = torch.zeros(2 * C, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def wg(s): lib.mi355_conv2d_wgrad(x, dy, ws, splits, N, H, W, C, C, H, W, C, C, 3, 3, 1, 1, 0, code, s.cuda_stream)
def bn(s): lib.mi355_bn_bwd_apply(g, C, None, 0, y, C, gamma, mean, inv, sc, sh, sums, dx, C, None, 0, None, 0, 0, None, M, C, 1, code, s.cuda_stream)
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
tw = timeit(lambda: wg(s1)); tb = timeit(lambda: bn(s1))
tboth = timeit(lambda: (wg(s1), bn(s2)))
tb3 = timeit(lambda: (wg(s1), bn(s2), bn(s2)))
print(f"wgrad {tw:.3f} ms, bn_bwd_apply {tb:.3f} ms, both on two streams {tboth:.3f} ms (sum {tw + tb:.3f}); wgrad + 2 bn: {tb3:.3f} (sum {tw + 2 * tb:.3f})")
