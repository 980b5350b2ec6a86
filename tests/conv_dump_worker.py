"""Helper process of tests/test_gpu_conv.py::test_counted_vmcnt_matches_drained_build (not a test): runs a fixed set of forward /
data-gradient / weight-gradient launches of the LDS-DMA kernels through whatever library MI355_LIB selects and writes the raw
outputs to an .npz.      usage: python conv_dump_worker.py OUT.npz"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "medical-image-segmentation-and-classification_amd"), os.path.dirname(os.path.abspath(__file__))]

import numpy as np
import torch

from gpu_util import DEV, DTYPE_CODE, lib, pack_w, to_nhwc

CASES = [  # N, Ci, H, W, Co, k, up          (halo 8x32 / 16x16 / ping-pong-eligible / igemm-DMA 1x1 and strided shapes)
    (2, 64, 16, 32, 128, 3, 0), (1, 160, 32, 64, 64, 3, 0), (2, 96, 16, 16, 192, 3, 0), (2, 64, 8, 16, 128, 3, 1),
    (3, 256, 9, 7, 128, 1, 0), (4, 128, 24, 32, 64, 3, 0),
    # 128-channel ping-pong kernel at its default threshold (Ci >= 256): forward only / forward and data gradient / fused up-sampling
    (2, 256, 32, 64, 128, 3, 0), (2, 256, 16, 32, 256, 3, 0), (1, 256, 8, 16, 128, 3, 1),
]


# MI355_DUMP_SET=ws64: shapes of the weight-stationary kernel (Ci = 64 forward, Co = 64 data gradients), run by
# test_weight_stationary_kernel_is_bit_identical_to_the_halo_kernel under MI355_WS64=0 and under MI355_WS64_MIN_TILES=8:
# uneven tile ranges per workgroup (18 tiles on 16 groups), one / two / three channel tiles, fused up-sampling, 2x2-sum epilogue
WS64_CASES = [(3, 64, 24, 64, 64, 3, 0), (2, 64, 32, 32, 128, 3, 0), (5, 64, 16, 96, 192, 3, 0), (2, 64, 16, 32, 64, 3, 1),
              (4, 128, 24, 32, 64, 3, 0), (9, 64, 8, 32, 64, 3, 0),
              # Ci = 128 instantiation (4 x 32-pixel tiles, four slabs): uneven ranges, 1 / 2 channel tiles, H % 8 == 4, up-sampling
              (3, 128, 24, 64, 128, 3, 0), (2, 128, 20, 32, 64, 3, 0), (5, 128, 8, 32, 128, 3, 1), (3, 128, 16, 96, 64, 3, 0)]


def main_ws64():
    rec = {}
    for ci_, (n, ci, h, w_, co, k, up) in enumerate(WS64_CASES):
        for dtype in (torch.bfloat16, torch.float16):
            g = torch.Generator().manual_seed(300 + ci_)
            x = torch.randn(n, ci, h, w_, generator=g)
            w = torch.randn(co, ci, k, k, generator=g) / (ci * k * k) ** 0.5
            b = torch.randn(co, generator=g)
            ho, wo = (2 * h, 2 * w_) if up else (h, w_)
            dy = torch.randn(n, co, ho, wo, generator=g)
            code = DTYPE_CODE[dtype]
            xd, dyd = to_nhwc(x, dtype), to_nhwc(dy, dtype)
            wf, wb = pack_w(w, dtype)
            tag = f"w{ci_}_{'bf16' if dtype == torch.bfloat16 else 'fp16'}"
            rec[f"{tag}_variant_fwd"] = np.array(lib.mi355_conv2d_igemm_variant_n(n, h, w_, ci, ho, wo, co, k, k, 1, 1, -1, 1, up, code))
            rec[f"{tag}_variant_dgrad"] = np.array(lib.mi355_conv2d_igemm_variant_n(n, ho, wo, co, ho, wo, ci, k, k, 1, -1, 1, 1, 0, code))
            for rep in range(2):
                y = torch.full((n, ho, wo, co), float("nan"), dtype=dtype, device=DEV)
                lib.mi355_conv2d_igemm(xd, wf, None, y, n, h, w_, ci, ci, ho, wo, co, co, k, k, 1, 1, -1, 1, up, 0, None, code)
                rec[f"{tag}_fwd{rep}"] = y.view(torch.int16).cpu().numpy()
                # bias + ReLU + fused statistics, output into a channel slice of a wider buffer
                rows = lib.mi355_conv2d_igemm_stat_rows(n, h, w_, ci, ho, wo, co, k, k, 1, 1, -1, 1, up, code)
                part = torch.zeros(rows * 2 * co, device=DEV)
                wide = torch.zeros(n, ho, wo, co + 32, dtype=dtype, device=DEV)
                lib.mi355_conv2d_igemm(xd, wf, b.to(DEV), wide.data_ptr() + 32, n, h, w_, ci, ci, ho, wo, co, co + 32, k, k, 1, 1, -1, 1, up, 2,
                                       part, code)
                rec[f"{tag}_relu{rep}"] = wide.view(torch.int16).cpu().numpy()
                rec[f"{tag}_stats{rep}"] = part.view(rows, 2, co).sum(0).cpu().numpy()
                # data gradient (on the up-sampled grid with the 2x2-sum epilogue where the forward folded the up-sampling in), accumulating
                dx = torch.ones(n, h, w_, ci, dtype=dtype, device=DEV)
                lib.mi355_conv2d_igemm(dyd, wb, None, dx, n, ho, wo, co, co, ho, wo, ci, ci, k, k, 1, -1, 1, 1, 0, (4 if up else 0) | 1, None, code)
                rec[f"{tag}_dgrad{rep}"] = dx.view(torch.int16).cpu().numpy()
    torch.cuda.synchronize()
    np.savez(sys.argv[1], **rec)


def main():
    if os.environ.get("MI355_DUMP_SET") == "ws64":
        return main_ws64()
    rec = {}
    for ci_, case in enumerate(CASES):
        n, ci, h, w_, co, k, up = case
        for dtype in (torch.bfloat16, torch.float16):
            g = torch.Generator().manual_seed(100 + ci_)
            x = torch.randn(n, ci, h, w_, generator=g)
            w = torch.randn(co, ci, k, k, generator=g) / (ci * k * k) ** 0.5
            ho, wo = (2 * h, 2 * w_) if up else (h, w_)
            dy = torch.randn(n, co, ho, wo, generator=g)
            code, p = DTYPE_CODE[dtype], k // 2
            xd, dyd = to_nhwc(x, dtype), to_nhwc(dy, dtype)
            wf, wb = pack_w(w, dtype)
            tag = f"c{ci_}_{'bf16' if dtype == torch.bfloat16 else 'fp16'}"
            for rep in range(3):                       # the same launch three times: a race would also differ run to run
                y = torch.empty(n, ho, wo, co, dtype=dtype, device=DEV)
                lib.mi355_conv2d_igemm(xd, wf, None, y, n, h, w_, ci, ci, ho, wo, co, co, k, k, 1, 1, -p, 1, up, 0, None, code)
                rec[f"{tag}_fwd{rep}"] = y.view(torch.int16).cpu().numpy()
                if not up:
                    dx = torch.empty(n, h, w_, ci, dtype=dtype, device=DEV)
                    lib.mi355_conv2d_igemm(dyd, wb, None, dx, n, ho, wo, co, co, h, w_, ci, ci, k, k, 1, -1, p, 1, 0, 0, None, code)
                    rec[f"{tag}_dgrad{rep}"] = dx.view(torch.int16).cpu().numpy()
                sp = lib.mi355_conv2d_wgrad_splits(n, ho, wo, ci, co, k, k)
                ws = torch.empty(sp, co, k * k, ci, device=DEV)
                lib.mi355_conv2d_wgrad(xd, dyd, ws, sp, n, h, w_, ci, ci, ho, wo, co, co, k, k, 1, p, up, code)
                rec[f"{tag}_wgrad{rep}"] = ws.cpu().numpy()
    torch.cuda.synchronize()
    np.savez(sys.argv[1], **rec)


if __name__ == "__main__":
    main()
