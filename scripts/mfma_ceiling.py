"""What the chip sustains on the conv kernels' MFMA shape with random operands and nothing else in the way
(csrc/probe.hip).  Prints TFLOP/s for: 16x16x32 bf16 from registers, the same with the halo kernel's LDS diet, and
32x32x16 bf16 from registers; 2 workgroups x 4 waves per CU, ~10 ms per launch, after a 2-second soak."""
import os, sys, time
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'medical-image-segmentation-and-classification_amd'))
import torch
import ctypes
_dll = ctypes.CDLL(os.path.join(R, 'medical-image-segmentation-and-classification_amd', 'mi355', 'libmi355probe.so'))
_dll.mi355_probe_mfma.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]


class lib:      # include/mi355probe.h
    @staticmethod
    def mi355_probe_mfma(mode, rnd, blocks, iters, sink):
        rc = _dll.mi355_probe_mfma(mode, rnd.data_ptr(), blocks, iters, sink.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
dev = "cuda:0"
rnd = (torch.randn(65536 * 8, device=dev)).to(torch.bfloat16)
sink = torch.zeros(4, device=dev)
blocks = 512
flops_per_trip = {0: 48 * 16 * 16 * 32 * 2, 1: 48 * 16 * 16 * 32 * 2, 2: 24 * 32 * 32 * 16 * 2}
names = {0: "16x16x32 bf16, operands in registers", 1: "16x16x32 bf16 + 18 ds_read_b128 per 48 MFMAs", 2: "32x32x16 bf16, operands in registers"}
for mode in (0, 1, 2):
    iters = 20000
    t_end = time.time() + 2.0
    while time.time() < t_end:                       # soak: let the clock settle under this load
        lib.mi355_probe_mfma(mode, rnd, blocks, iters, sink)
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        lib.mi355_probe_mfma(mode, rnd, blocks, iters, sink)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    tf = blocks * 4 * iters * flops_per_trip[mode] / (ms * 1e-3) / 1e12
    print(f"{names[mode]:48s}: {tf:7.1f} TFLOP/s  ({ms:.2f} ms per launch, {tf / 2500:.2f} of the 2.5 PFLOP/s dense peak)")
