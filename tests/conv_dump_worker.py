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


def main():
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
