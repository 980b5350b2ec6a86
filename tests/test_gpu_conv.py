"""-m gpu: implicit-GEMM conv (forward, data gradient, ConvTranspose) through the C ABI vs
torch CPU fp32 (F.conv2d / autograd).  Tolerances: fp32 1e-4 relative-to-max; bf16 / fp16 compare
against the CPU result on operands rounded to that type: 2e-2 / 3e-3 relative-to-max (output rounding 2^-8 / 2^-11)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from gpu_util import DEV, lib, to_nhwc, from_nhwc, pack_w, rel_err, q, DTYPE_CODE

pytestmark = pytest.mark.gpu
TOL = {torch.float32: 1e-4, torch.bfloat16: 2e-2, torch.float16: 3e-3}
DT = [torch.float32, torch.bfloat16, torch.float16]

CASES = [  # N, Ci, H, W, Co, K, stride, pad, up
    (2, 32, 12, 12, 64, 3, 1, 1, 0),
    (2, 64, 16, 16, 128, 3, 1, 1, 0),
    (1, 32, 9, 7, 32, 3, 1, 1, 0),
    (3, 64, 8, 8, 64, 1, 1, 0, 0),
    (2, 32, 16, 16, 64, 3, 2, 1, 0),
    (2, 32, 20, 20, 64, 7, 2, 3, 0),
    (2, 64, 6, 6, 32, 3, 1, 1, 1),
    (2, 128, 10, 10, 256, 3, 1, 1, 0),
    (1, 96, 8, 8, 96, 3, 1, 1, 0),
    # halo-patch kernel shapes: (W % 32, H % 8) and (W % 16, H % 16) tiles, fused up-sampling, Co = 64 / 128
    (2, 64, 16, 32, 128, 3, 1, 1, 0),
    (1, 32, 8, 64, 64, 3, 1, 1, 0),
    (2, 96, 16, 16, 192, 3, 1, 1, 0),
    (2, 64, 8, 16, 128, 3, 1, 1, 1),
    (1, 128, 24, 32, 64, 3, 1, 1, 0),
    # ping-pong halo kernel (conv3x3_halo_pp.hpp: H % 16 == 0, W % 32 == 0): one / three / five 32-channel slabs (odd and even
    # slab counts take different exits of the unrolled loop), several tiles per image in both directions, fused up-sampling
    (1, 32, 16, 32, 64, 3, 1, 1, 0),
    (2, 96, 32, 64, 64, 3, 1, 1, 0),
    (1, 160, 48, 32, 128, 3, 1, 1, 0),
    (3, 64, 16, 96, 192, 3, 1, 1, 0),
    (2, 64, 16, 32, 64, 3, 1, 1, 1),
    (1, 256, 16, 32, 64, 3, 1, 1, 0),
    # 128-channel ping-pong kernel (conv3x3_halo_pp128.hpp: Co % 128 == 0, Ci % 64 == 0): two / four / six slabs, one and several
    # tiles per image, two channel tiles, fused up-sampling; the data gradients of these cases run it with Ci and Co swapped
    (2, 128, 32, 64, 256, 3, 1, 1, 0),
    (1, 192, 16, 32, 128, 3, 1, 1, 0),
    (2, 64, 8, 16, 128, 3, 1, 1, 1),
    (1, 128, 48, 32, 128, 3, 1, 1, 0),
    # ... and more workgroups than CUs (36 x 4 x 2 = 288 tiles of 16 x 32 x 128: a second round of workgroups on some CUs)
    (36, 64, 32, 64, 256, 3, 1, 1, 0),
    # weight-stationary kernel (conv3x3_ws.hpp, Ci == 64 instantiation) at its default threshold of 4 x 256 tiles: one and two channel tiles
    # (the data gradient of the first case runs it too: Co = 64 there)
    (16, 64, 128, 128, 64, 3, 1, 1, 0),
    (8, 64, 128, 128, 128, 3, 1, 1, 0),
    # ... and its Ci = 128 instantiation (4 x 32-pixel tiles; threshold 24 x 256 tiles = 12 per workgroup): four channel tiles
    (12, 128, 128, 128, 256, 3, 1, 1, 0),
]


def _conv_ref(x, w, b, s, p, up):
    if up:
        x = F.interpolate(x, scale_factor=2.0, mode="nearest")
    return F.conv2d(x, w, b, stride=s, padding=p)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("case", CASES)
def test_conv_fwd(case, dtype):
    n, ci, h, w_, co, k, s, p, up = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn(n, ci, h, w_, generator=g)
    w = torch.randn(co, ci, k, k, generator=g) / (ci * k * k) ** 0.5
    b = torch.randn(co, generator=g)
    ref = _conv_ref(q(x, dtype), q(w, dtype), b, s, p, up)
    ho, wo = ref.shape[2], ref.shape[3]
    xd = to_nhwc(x, dtype)
    wf, _ = pack_w(w, dtype)
    y = torch.full((n, ho, wo, co), float("nan"), dtype=dtype, device=DEV)
    lib.mi355_conv2d_igemm(xd, wf, b.to(DEV), y, n, h, w_, ci, ci, ho, wo, co, co, k, k, s, 1, -p, 1, up, 0, None,
                           DTYPE_CODE[dtype])
    torch.cuda.synchronize()
    assert rel_err(from_nhwc(y), ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("case", [c for c in CASES if not c[8]])
def test_conv_dgrad(case, dtype):
    n, ci, h, w_, co, k, s, p, up = case
    g = torch.Generator().manual_seed(7 + hash(case) % 1000)
    x = torch.randn(n, ci, h, w_, generator=g, requires_grad=True)
    w = torch.randn(co, ci, k, k, generator=g) / (co * k * k) ** 0.5
    y = F.conv2d(x, q(w, dtype), None, stride=s, padding=p)
    dy = torch.randn(y.shape, generator=g)
    y.backward(q(dy, dtype))
    ref = x.grad
    ho, wo = y.shape[2], y.shape[3]
    dyd = to_nhwc(dy, dtype)
    _, wb = pack_w(w, dtype)
    dx = torch.full((n, h, w_, ci), float("nan"), dtype=dtype, device=DEV)
    # data gradient = gather over dy with the [Ci][tap][Co] pack: mul=1,kmul=-1,off=+p,div=s
    lib.mi355_conv2d_igemm(dyd, wb, None, dx, n, ho, wo, co, co, h, w_, ci, ci, k, k, 1, -1, p, s, 0, 0, None,
                           DTYPE_CODE[dtype])
    torch.cuda.synchronize()
    assert rel_err(from_nhwc(dx), ref) < TOL[dtype]


# (8, 8): generic kernel; (16, 32) / (32, 32): halo kernels, one and two tiles per image; 256 -> 128 channels on 32 x 64 (four tiles of
# the 128-channel ping-pong kernel per image)
@pytest.mark.parametrize("hw,ci,co,n", [((8, 8), 32, 64, 2), ((16, 32), 32, 64, 2), ((32, 32), 32, 64, 2), ((32, 64), 256, 128, 2),
                                        ((16, 32), 256, 256, 3)])
@pytest.mark.parametrize("dtype", DT)
def test_conv_strided_slices_and_accumulate(dtype, hw, ci, co, n):
    """input read from / output written into channel slices of wider buffers; accumulate flag; ReLU epilogue."""
    h, w_ = hw
    g = torch.Generator().manual_seed(3)
    x = torch.randn(n, ci, h, w_, generator=g); w = torch.randn(co, ci, 3, 3, generator=g) * (0.1 * (32 / ci) ** 0.5)
    ref = F.conv2d(q(x, dtype), q(w, dtype), None, padding=1)
    xin = torch.zeros(n, h, w_, ci + 64, dtype=dtype, device=DEV); xin[..., 64:] = to_nhwc(x, dtype)
    wf, _ = pack_w(w, dtype)
    base = torch.randn(n, h, w_, co + 96, generator=g).to(dtype)
    out = base.clone().to(DEV)
    es = out.element_size()
    lib.mi355_conv2d_igemm(xin.data_ptr() + 64 * es, wf, None, out.data_ptr() + 32 * es, n, h, w_, ci, ci + 64, h, w_, co,
                           co + 96, 3, 3, 1, 1, -1, 1, 0, 1, None, DTYPE_CODE[dtype])
    torch.cuda.synchronize()
    got = out.float().cpu()
    exp = base.float().clone(); exp[..., 32:32 + co] += ref.permute(0, 2, 3, 1)
    assert rel_err(got[..., 32:32 + co], q(exp[..., 32:32 + co], dtype)) < TOL[dtype]
    assert torch.equal(got[..., :32], base.float()[..., :32]) and torch.equal(got[..., 32 + co:], base.float()[..., 32 + co:])
    if dtype != torch.float32:                                  # ReLU in the epilogue (bit 1), with a bias
        b = torch.randn(co, generator=g)
        y = torch.empty(n, h, w_, co, dtype=dtype, device=DEV)
        lib.mi355_conv2d_igemm(xin.data_ptr() + 64 * es, wf, b.to(DEV), y, n, h, w_, ci, ci + 64, h, w_, co, co, 3, 3, 1, 1, -1, 1, 0, 2, None,
                               DTYPE_CODE[dtype])
        torch.cuda.synchronize()
        assert rel_err(from_nhwc(y), torch.relu(ref + b.view(1, -1, 1, 1))) < TOL[dtype]


@pytest.mark.parametrize("dtype", DT)
def test_conv_transpose_2x2(dtype):
    n, ci, h, w_, co = 2, 64, 5, 6, 32
    g = torch.Generator().manual_seed(11)
    x = torch.randn(n, ci, h, w_, generator=g); w = torch.randn(ci, co, 2, 2, generator=g) * 0.1
    b = torch.randn(co, generator=g)
    ref = F.conv_transpose2d(q(x, dtype), q(w, dtype), b, stride=2)
    wf, _ = pack_w(w, dtype, transposed=True)          # [Co][tap][Ci]
    y = torch.empty(n, 2 * h, 2 * w_, co, dtype=dtype, device=DEV)
    lib.mi355_conv2d_igemm(to_nhwc(x, dtype), wf, b.to(DEV), y, n, h, w_, ci, ci, 2 * h, 2 * w_, co, co, 2, 2, 1, -1, 0,
                           2, 0, 0, None, DTYPE_CODE[dtype])
    torch.cuda.synchronize()
    assert rel_err(from_nhwc(y), ref) < TOL[dtype]


GEMM_CASES = [  # kind, N, Ci, Hi, Wi, Co          (>= 128 tiles of 256 rows x 128 columns each: the launcher's threshold)
    ("1x1", 8, 128, 32, 32, 512), ("1x1", 4, 512, 64, 64, 256), ("1x1s2", 8, 256, 64, 64, 512), ("2x2s2", 8, 64, 64, 64, 512),
    ("convT", 4, 256, 16, 16, 1024), ("convT", 16, 64, 32, 32, 128),
]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", GEMM_CASES)
def test_padding_free_convolutions_as_gemms(case, dtype):
    """conv_gemm256_kernel (csrc/conv_gemm256.hpp, variant 9): 1x1 convolutions at stride 1 and 2 (ResNet-50 bottlenecks and
    shortcuts, the attention gates' wide W_g / W_x), 2x2 / stride-2 convolutions (the data gradient of ConvTranspose2d(2, 2)) and
    ConvTranspose2d(2, 2) itself as four pointwise phases (ResnetUnet.py:21,51) against torch — plain, with bias + ReLU + fused
    BatchNorm statistics into a channel slice of a wider buffer, and accumulating."""
    kind, n, ci, h, w_, co = case
    code = DTYPE_CODE[dtype]
    g = torch.Generator().manual_seed(ci + co + h)
    x = torch.randn(n, ci, h, w_, generator=g)
    b = torch.randn(co, generator=g)
    if kind == "convT":
        w = torch.randn(ci, co, 2, 2, generator=g) / (ci ** 0.5)
        ref = F.conv_transpose2d(q(x, dtype), q(w, dtype), b, stride=2)
        wf, _ = pack_w(w, dtype, transposed=True)
        k, geo = 2, (1, -1, 0, 2)                       # mul, kmul, off, div
    else:
        k, s = {"1x1": (1, 1), "1x1s2": (1, 2), "2x2s2": (2, 2)}[kind]
        w = torch.randn(co, ci, k, k, generator=g) / ((ci * k * k) ** 0.5)
        ref = F.conv2d(q(x, dtype), q(w, dtype), b, stride=s)
        wf, _ = pack_w(w, dtype)
        geo = (s, 1, 0, 1)
    ho, wo = ref.shape[2], ref.shape[3]
    assert lib.mi355_conv2d_igemm_variant_n(n, h, w_, ci, ho, wo, co, k, k, *geo, 0, code) == 9
    xd = to_nhwc(x, dtype)
    y = torch.full((n, ho, wo, co), float("nan"), dtype=dtype, device=DEV)
    lib.mi355_conv2d_igemm(xd, wf, b.to(DEV), y, n, h, w_, ci, ci, ho, wo, co, co, k, k, *geo, 0, 0, None, code)
    torch.cuda.synchronize()
    assert rel_err(from_nhwc(y), ref) < TOL[dtype]
    # accumulate (a data gradient joining an existing one)
    y2 = y.clone()
    lib.mi355_conv2d_igemm(xd, wf, b.to(DEV), y2, n, h, w_, ci, ci, ho, wo, co, co, k, k, *geo, 0, 1, None, code)
    torch.cuda.synchronize()
    assert rel_err(from_nhwc(y2), 2 * ref) < 2 * TOL[dtype]
    # bias + ReLU (+ fused statistics where the variant has them) into a channel slice of a wider buffer
    rows = lib.mi355_conv2d_igemm_stat_rows(n, h, w_, ci, ho, wo, co, k, k, *geo, 0, code)
    assert rows == (0 if kind == "convT" else n * ho * wo // 256)
    part = torch.full((max(rows, 1) * 2 * co,), float("nan"), device=DEV)
    wide = torch.zeros(n, ho, wo, co + 32, dtype=dtype, device=DEV)
    lib.mi355_conv2d_igemm(xd, wf, b.to(DEV), wide.data_ptr() + 32 * 2, n, h, w_, ci, ci, ho, wo, co, co + 32, k, k, *geo, 0, 2,
                           part if rows else None, code)
    torch.cuda.synchronize()
    got = wide[..., 32:].float().cpu().permute(0, 3, 1, 2)
    assert rel_err(got, F.relu(ref)) < TOL[dtype] and float(wide[..., :32].abs().max()) == 0.0
    if rows:
        st = part.view(rows, 2, co).sum(0).cpu()
        assert rel_err(st[0], got.double().sum((0, 2, 3)).float()) < 1e-3 and rel_err(st[1], (got.double() ** 2).sum((0, 2, 3)).float()) < 1e-3


def test_bad_args_raise():
    x = torch.zeros(1, 4, 4, 24, device=DEV)
    with pytest.raises(RuntimeError, match="Ci"):
        lib.mi355_conv2d_igemm(x, x, None, x, 1, 4, 4, 24, 24, 4, 4, 32, 32, 3, 3, 1, 1, -1, 1, 0, 0, None, 0)


WG_CASES = [  # N, Ci, H, W, Co, K, stride, pad, up
    (2, 32, 12, 12, 64, 3, 1, 1, 0),
    (2, 64, 16, 16, 128, 3, 1, 1, 0),
    (1, 32, 9, 7, 32, 3, 1, 1, 0),
    (3, 64, 8, 8, 64, 1, 1, 0, 0),
    (2, 32, 16, 16, 64, 3, 2, 1, 0),
    (2, 32, 20, 20, 64, 7, 2, 3, 0),
    (2, 64, 6, 6, 32, 3, 1, 1, 1),
    (2, 128, 10, 10, 256, 3, 1, 1, 0),
    (4, 128, 24, 24, 128, 3, 1, 1, 0),
    (1, 96, 8, 8, 160, 3, 1, 1, 0),
    # nine-tap halo kernel shapes (W % 32 == 0, H % 8 == 0), incl. channel tails and fused up-sampling
    (2, 64, 16, 32, 64, 3, 1, 1, 0),
    (1, 32, 8, 64, 96, 3, 1, 1, 0),
    (2, 64, 8, 32, 32, 3, 1, 1, 0),
    # 16-pixel-wide images, two at a time: an odd pair count, channel tails, an 8-row image, fused up-sampling (Wi = 8)
    (6, 64, 16, 16, 64, 3, 1, 1, 0),
    (4, 96, 8, 16, 160, 3, 1, 1, 0),
    (2, 64, 8, 8, 64, 3, 1, 1, 1),
    (3, 64, 16, 16, 64, 3, 1, 1, 0),      # odd batch: stays on the generic kernel
    (2, 128, 8, 16, 64, 3, 1, 1, 1),
    (1, 192, 40, 32, 128, 3, 1, 1, 0),
    (3, 64, 32, 96, 64, 3, 1, 1, 0),
    # eight-wave nine-tap kernel (wgrad3x3_halo8.hpp): rows of 64-pixel segments — two segments per row and a 32-row band, several
    # channel tiles, two bands per image, fused up-sampling — and 32-pixel-wide images two at a time: 8-row bands with a channel
    # tail, fused up-sampling (Wi = 16)
    (2, 64, 32, 128, 64, 3, 1, 1, 0),
    (1, 128, 16, 64, 192, 3, 1, 1, 0),
    (1, 64, 64, 64, 64, 3, 1, 1, 0),
    (2, 64, 16, 32, 64, 3, 1, 1, 1),
    (4, 96, 24, 32, 64, 3, 1, 1, 0),
    (4, 64, 8, 16, 128, 3, 1, 1, 1),
]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(2, 128, 8, 16, 64), (1, 64, 16, 16, 128), (2, 64, 4, 16, 64), (2, 64, 16, 32, 128)])      # n, co, h, w (input grid), ci
def test_upsampled_conv_dgrad_with_fused_2x2_sum(shape, dtype):
    """conv(Upsample(x2)(x)) (AttentionUNet.py:19-20): the data gradient on the up-sampled grid is summed over 2x2 groups
    in the kernel epilogue (accumulate bit 2) — against autograd through F.interpolate, plain and accumulating."""
    n, co, h, w_, ci = shape
    g = torch.Generator().manual_seed(co + h)
    x = torch.randn(n, ci, h, w_, generator=g, requires_grad=True)
    w = torch.randn(co, ci, 3, 3, generator=g) / (co * 9) ** 0.5
    y = F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), q(w, dtype), None, padding=1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(q(dy, dtype))
    code = DTYPE_CODE[dtype]
    assert lib.mi355_conv2d_igemm_variant(2 * h, 2 * w_, co, 2 * h, 2 * w_, ci, 3, 3, 1, -1, 1, 1, 0, code) >= 2
    _, wb = pack_w(w, dtype)
    dx = torch.full((n, h, w_, ci), float("nan"), dtype=dtype, device=DEV)
    lib.mi355_conv2d_igemm(to_nhwc(dy, dtype), wb, None, dx, n, 2 * h, 2 * w_, co, co, 2 * h, 2 * w_, ci, ci, 3, 3, 1, -1, 1, 1, 0, 4,
                           None, code)
    torch.cuda.synchronize()
    assert rel_err(from_nhwc(dx), x.grad) < TOL[dtype]
    lib.mi355_conv2d_igemm(to_nhwc(dy, dtype), wb, None, dx, n, 2 * h, 2 * w_, co, co, 2 * h, 2 * w_, ci, ci, 3, 3, 1, -1, 1, 1, 0, 5,
                           None, code)
    torch.cuda.synchronize()
    assert rel_err(from_nhwc(dx), 2 * x.grad) < 2 * TOL[dtype]
    # the kernels without that epilogue refuse the flag (fp32 / 1x1 shapes)
    with pytest.raises(RuntimeError):
        lib.mi355_conv2d_igemm(to_nhwc(dy, torch.float32), wb.float(), None, dx.float(), n, 2 * h, 2 * w_, co, co, 2 * h, 2 * w_, ci, ci,
                               3, 3, 1, -1, 1, 1, 0, 4, None, DTYPE_CODE[torch.float32])


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("case", WG_CASES)
def test_conv_wgrad(case, dtype):
    n, ci, h, w_, co, k, s, p, up = case
    g = torch.Generator().manual_seed(13 + hash(case) % 1000)
    x = torch.randn(n, ci, h, w_, generator=g)
    w = torch.zeros(co, ci, k, k, requires_grad=True)
    y = _conv_ref(q(x, dtype), w, None, s, p, up)
    dy = torch.randn(y.shape, generator=g)
    y.backward(q(dy, dtype))
    ref = w.grad
    ho, wo = y.shape[2], y.shape[3]
    cip = ci + 32 if case == WG_CASES[0] else ci       # exercise a zero-padded input (stem)
    xd = to_nhwc(x, dtype, cpad=cip)
    dyd = to_nhwc(dy, dtype)
    splits = lib.mi355_conv2d_wgrad_splits(n, ho, wo, cip, co, k, k)
    assert splits >= 1
    for sp in sorted({splits, 1, 3}):
        ws = torch.full((sp, co, k * k, cip), float("nan"), device=DEV)
        lib.mi355_conv2d_wgrad(xd, dyd, ws, sp, n, h, w_, cip, cip, ho, wo, co, co, k, k, s, p, up, DTYPE_CODE[dtype])
        dw = torch.full((co, ci, k, k), 2.0, device=DEV)
        lib.mi355_conv2d_wgrad_reduce(ws, sp, dw, co, cip, ci, k, k, 0, 0.0)
        torch.cuda.synchronize()
        assert rel_err(dw.cpu(), ref) < TOL[dtype], f"splits={sp}"
        lib.mi355_conv2d_wgrad_reduce(ws, sp, dw, co, cip, ci, k, k, 0, 1.0)   # beta = 1 accumulates
        torch.cuda.synchronize()
        assert rel_err(dw.cpu(), 2 * ref) < TOL[dtype]


@pytest.mark.parametrize("case", [(2, 64, 16, 32, 128, 3, 1, 1, 0), (2, 96, 32, 64, 64, 3, 1, 1, 0), (2, 96, 16, 16, 64, 3, 1, 1, 0), (3, 64, 9, 7, 64, 1, 1, 0, 0),
                                  (2, 128, 32, 64, 256, 3, 1, 1, 0), (36, 64, 32, 64, 256, 3, 1, 1, 0),
                                  (2, 64, 8, 16, 128, 3, 1, 1, 1), (1, 32, 5, 5, 128, 3, 2, 1, 0)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_conv_fused_bn_statistics(case, dtype):
    """bf16 / fp16 kernels fold the BatchNorm partial sums of the (rounded) outputs into their epilogue."""
    n, ci, h, w_, co, k, s, p, up = case
    g = torch.Generator().manual_seed(99)
    x = torch.randn(n, ci, h, w_, generator=g); w = torch.randn(co, ci, k, k, generator=g) / (ci * k * k) ** 0.5
    b = torch.randn(co, generator=g)
    ref = _conv_ref(q(x, dtype), q(w, dtype), b, s, p, up)
    ho, wo = ref.shape[2], ref.shape[3]
    code = DTYPE_CODE[dtype]
    rows = lib.mi355_conv2d_igemm_stat_rows(n, h, w_, ci, ho, wo, co, k, k, s, 1, -p, 1, up, code)
    assert rows > 0
    part = torch.full((rows * 2 * co,), float("nan"), device=DEV)
    y = torch.empty(n, ho, wo, co, dtype=dtype, device=DEV)
    wf, _ = pack_w(w, dtype)
    lib.mi355_conv2d_igemm(to_nhwc(x, dtype), wf, b.to(DEV), y, n, h, w_, ci, ci, ho, wo, co, co, k, k, s, 1, -p, 1, up, 0, part, code)
    sc, sh, mu, isd = (torch.empty(co, device=DEV) for _ in range(4))
    ones, zeros = torch.ones(co, device=DEV), torch.zeros(co, device=DEV)
    lib.mi355_bn_finalize(part, rows, n * ho * wo, co, ones, zeros, None, None, None, 0.1, 1e-5, sc, sh, mu, isd)
    torch.cuda.synchronize()
    yf = from_nhwc(y)
    assert rel_err(yf, ref) < TOL[dtype]
    assert rel_err(mu.cpu(), yf.mean((0, 2, 3))) < 1e-4
    assert rel_err(isd.cpu(), 1 / torch.sqrt(yf.var((0, 2, 3), unbiased=False) + 1e-5)) < 1e-4
    assert lib.mi355_conv2d_igemm_stat_rows(n, h, w_, ci, ho, wo, co, k, k, s, 1, -p, 1, up, DTYPE_CODE[torch.float32]) == 0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("ci,co", [(64, 32), (32, 64), (128, 64), (64, 128), (64, 64), (32, 32)])
@pytest.mark.parametrize("nhw", [(3, 9, 7), (2, 40, 37), (1, 3, 3), (1, 1, 2)])      # 189 / 2960 / 9 / 2 pixels (idle waves, one ragged block)
def test_pointwise_stream_kernel(nhw, ci, co, dtype):
    """Narrow 1x1 convolutions run on conv1x1_stream_kernel (variant 4): ragged pixel counts, bias + ReLU + fused statistics,
    then the same call reading / accumulating into channel slices of wider buffers (the data-gradient use)."""
    n, h, w_ = nhw
    code = DTYPE_CODE[dtype]
    assert lib.mi355_conv2d_igemm_variant(h, w_, ci, h, w_, co, 1, 1, 1, 1, 0, 1, 0, code) == 4
    g = torch.Generator().manual_seed(ci * 7 + co)
    x = torch.randn(n, ci, h, w_, generator=g); w = torch.randn(co, ci, 1, 1, generator=g) / ci ** 0.5
    b = torch.randn(co, generator=g)
    ref = F.relu(F.conv2d(q(x, dtype), q(w, dtype), b))
    rows = lib.mi355_conv2d_igemm_stat_rows(n, h, w_, ci, h, w_, co, 1, 1, 1, 1, 0, 1, 0, code)
    assert rows == min(-(-n * h * w_ // 256), 2048)
    part = torch.full((rows * 2 * co,), float("nan"), device=DEV)
    y = torch.full((n, h, w_, co), float("nan"), dtype=dtype, device=DEV)
    wf, wb = pack_w(w, dtype)
    xd = to_nhwc(x, dtype)
    lib.mi355_conv2d_igemm(xd, wf, b.to(DEV), y, n, h, w_, ci, ci, h, w_, co, co, 1, 1, 1, 1, 0, 1, 0, 2, part, code)
    sc, sh, mu, isd = (torch.empty(co, device=DEV) for _ in range(4))
    lib.mi355_bn_finalize(part, rows, n * h * w_, co, torch.ones(co, device=DEV), torch.zeros(co, device=DEV), None, None, None,
                          0.1, 1e-5, sc, sh, mu, isd)
    torch.cuda.synchronize()
    yf = from_nhwc(y)
    assert rel_err(yf, ref) < TOL[dtype]
    assert rel_err(mu.cpu(), yf.mean((0, 2, 3))) < 1e-4
    assert rel_err(isd.cpu(), 1 / torch.sqrt(yf.var((0, 2, 3), unbiased=False) + 1e-5)) < 1e-4
    # data gradient of the same layer: dy [.., co] -> dx [.., ci] with the transposed pack, read from a slice of a wider
    # buffer and ACCUMULATED into a slice of another
    dy = torch.randn(n, co, h, w_, generator=g)
    dref = F.conv_transpose2d(q(dy, dtype), q(w, dtype))
    wide_in = torch.zeros(n, h, w_, co + 32, dtype=dtype, device=DEV); wide_in[..., 32:] = to_nhwc(dy, dtype)
    base = torch.randn(n, h, w_, ci + 64, generator=g).to(dtype)
    out = base.clone().to(DEV)
    es = out.element_size()
    assert lib.mi355_conv2d_igemm_variant(h, w_, co, h, w_, ci, 1, 1, 1, -1, 0, 1, 0, code) == 4
    lib.mi355_conv2d_igemm(wide_in.data_ptr() + 32 * es, wb, None, out.data_ptr() + 32 * es, n, h, w_, co, co + 32, h, w_, ci,
                           ci + 64, 1, 1, 1, -1, 0, 1, 0, 1, None, code)
    torch.cuda.synchronize()
    got = out.float().cpu()
    exp = base.float().clone(); exp[..., 32:32 + ci] += q(dref, dtype).permute(0, 2, 3, 1)
    assert rel_err(got[..., 32:32 + ci], q(exp[..., 32:32 + ci], dtype)) < TOL[dtype]
    assert torch.equal(got[..., :32], base.float()[..., :32]) and torch.equal(got[..., 32 + ci:], base.float()[..., 32 + ci:])


def test_ping_pong_halo_variant_passes_the_same_cases():
    """conv3x3_halo_pp.hpp (512-thread workgroups, load / matrix phases half a step apart) is opt-in (MI355_HALO_PP=1; it measured
    +1-3 % on the deepest layers and -10-15 % on the 256x256 ones).  The switch is read once per process, so the forward / data
    gradient / statistics / slice / ReLU / 2x2-sum cases of this file are re-run in a child process with the variant on."""
    import os, subprocess, sys
    if os.environ.get("MI355_HALO_PP") == "1":
        pytest.skip("already the child process")
    env = dict(os.environ, MI355_HALO_PP="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-k",
                        "fwd or dgrad or statistics or slices or upsampled"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    code = DTYPE_CODE[torch.bfloat16]
    assert lib.mi355_conv2d_igemm_variant(32, 64, 96, 32, 64, 64, 3, 3, 1, 1, -1, 1, 0, code) == 2           # default: the 4-wave kernel


def test_wide_tiles_pass_the_same_cases_with_the_small_grid_rule_off():
    """The generic and the LDS-DMA ring kernel take a narrower output-channel tile when the widest one would launch at most half a
    round of workgroups (csrc/conv_igemm.hip: small_grid_tile_n, dma_tile_n) — which is every small shape of this file.  The
    128- / 64-wide instances therefore get their parity cases in a child process with MI355_DMA_SMALLGRID=0 (the switch is read
    once per process), and the rule itself is asserted through its query functions."""
    import os, subprocess, sys
    if os.environ.get("MI355_DMA_SMALLGRID") == "0":
        pytest.skip("already the child process")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-k",
                        "fwd or dgrad or statistics or slices or transpose"], env=dict(os.environ, MI355_DMA_SMALLGRID="0"),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    # ResNet-50's layer4 at batch 32 (8 x 8 x 512 channels: 16 row tiles x 4 wide tiles = 64 workgroups) -> 32-wide tiles;
    # AttentionUNet's 32 x 32 x 512 -> 256 gate projection at batch 32 (256 row tiles) keeps the 128-wide tile
    assert lib.mi355_conv2d_igemm_dma_tile(32, 8, 8, 512, 512) == 32
    assert lib.mi355_conv2d_igemm_dma_tile(32, 16, 16, 1024, 256) == 64
    assert lib.mi355_conv2d_igemm_dma_tile(32, 32, 32, 512, 256) == 128
    assert lib.mi355_conv2d_igemm_dma_tile(2, 8, 8, 96, 64) == 64            # (Ci % 64 != 0: no 32-wide instance)
    assert lib.mi355_conv2d_igemm_generic_tile(8, 8, 8, 512) == 32           # ResNet-18 layer4 at batch 8 (fp32)
    assert lib.mi355_conv2d_igemm_generic_tile(32, 256, 256, 128) == 128


def test_128_channel_ping_pong_variant_passes_the_same_cases():
    """conv3x3_halo_pp128.hpp (16 x 32 pixels x 128 channels per 512-thread workgroup; the default from Ci = 256 up) with the
    threshold lowered to every eligible shape (MI355_HALO_PP128_MINCI=64) in a child process: forward / data gradient / statistics /
    2x2-sum cases; and the default dispatch: it serves 256-deep reductions, the 4-wave kernel the shallow ones."""
    import os, subprocess, sys
    if os.environ.get("MI355_HALO_PP128_MINCI"):
        pytest.skip("already the child process")
    env = dict(os.environ, MI355_HALO_PP128="1", MI355_HALO_PP128_MINCI="64")
    here = os.path.dirname(os.path.abspath(__file__))
    pkg = os.path.join(os.path.dirname(here), "medical-image-segmentation-and-classification_amd")
    probe = ("import sys; sys.path[:0] = [%r, %r]; from gpu_util import lib, DTYPE_CODE; import torch; "
             "print(lib.mi355_conv2d_igemm_variant(32, 64, 128, 32, 64, 256, 3, 3, 1, 1, -1, 1, 0, DTYPE_CODE[torch.bfloat16]))" % (here, pkg))
    r = subprocess.run([sys.executable, "-c", probe], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("6"), (r.stdout, r.stderr[-2000:])       # IG_HALO_PP128
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-k",
                        "fwd or dgrad or statistics or upsampled or slices"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    code = DTYPE_CODE[torch.bfloat16]
    assert lib.mi355_conv2d_igemm_variant(32, 64, 256, 32, 64, 128, 3, 3, 1, 1, -1, 1, 0, code) == 6
    assert lib.mi355_conv2d_igemm_variant(32, 64, 128, 32, 64, 128, 3, 3, 1, 1, -1, 1, 0, code) == 8          # (Ci = 128: weight-stationary)
    assert lib.mi355_conv2d_igemm_variant(32, 64, 256, 32, 64, 64, 3, 3, 1, 1, -1, 1, 0, code) == 2
    # one 512-thread workgroup per CU: a batch whose grid does not fill the chip is served by the 4-wave kernel, and the statistics-
    # row query (which knows N) follows the launcher — 32 x 32 x 512 -> 512: 32 images = 256 workgroups of 16 x 32 x 128, 16 images = 128
    assert lib.mi355_conv2d_igemm_stat_rows(32, 32, 32, 512, 32, 32, 512, 3, 3, 1, 1, -1, 1, 0, code) == 32 * 2 * 1
    assert lib.mi355_conv2d_igemm_stat_rows(16, 32, 32, 512, 32, 32, 512, 3, 3, 1, 1, -1, 1, 0, code) == 16 * 4 * 1


def test_weight_stationary_kernel_is_bit_identical_to_the_halo_kernel(tmp_path):
    """conv3x3_ws.hpp (Ci = 64 / 128: weights stationary in registers, persistent workgroups, the default from two tiles per
    workgroup up) accumulates every output element in the SAME order as the 4-wave halo kernel — (slab, patch column, tap row)
    with the bias as the first C operand — so its outputs must be BIT-identical to that kernel's (MI355_WS64=0) on every case:
    forward, bias + ReLU into a channel slice, accumulating data gradients, fused up-sampling with the 2x2-sum epilogue, uneven
    tile ranges, 1 / 2 / 3 channel tiles; run to run too (a race in the persistent double buffering would show as a difference);
    the fused BatchNorm sums agree to fp32 summation order."""
    import os, subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__))
    outs = {}
    for tag, env in (("ws64", {"MI355_WS64": "1", "MI355_WS128": "1", "MI355_WS64_MIN_TILES": "1"}), ("halo", {"MI355_WS64": "0", "MI355_WS128": "0"})):
        out = str(tmp_path / f"{tag}.npz")
        r = subprocess.run([sys.executable, os.path.join(here, "conv_dump_worker.py"), out], env=dict(os.environ, MI355_DUMP_SET="ws64", **env),
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[tag] = np.load(out)
    a, b = outs["ws64"], outs["halo"]
    assert set(a.files) == set(b.files)
    n_ws = n_ws128 = 0
    for k in a.files:
        if "_variant_" in k:
            n_ws += int(a[k]) == 7
            n_ws128 += int(a[k]) == 8
            assert int(b[k]) not in (7, 8)
        elif "_stats" in k:
            assert np.allclose(a[k], b[k], rtol=2e-5, atol=1e-2), k
        else:
            # the comparison build runs the 4-wave halo kernel (variants 2 / 3: same accumulation order -> bit-identical) wherever
            # the image divides into its tiles; an image with H % 8 == 4 goes to the LDS-DMA ring kernel there (another summation
            # order): rounding-level agreement
            which = k.split("_")[0] + "_" + k.split("_")[1] + ("_variant_dgrad" if "_dgrad" in k else "_variant_fwd")
            if int(b[which]) in (2, 3):
                assert np.array_equal(a[k], b[k]), k
            else:
                dt = torch.bfloat16 if "_bf16_" in k else torch.float16
                fa, fb = (torch.from_numpy(v.copy()).view(dt).float() for v in (a[k], b[k]))
                assert float((fa - fb).abs().max()) <= TOL[dt] * float(fb.abs().max()), k
            if k.endswith("0"):
                assert np.array_equal(a[k], a[k[:-1] + "1"]), k
    assert n_ws >= 14 and n_ws128 >= 10      # forwards with Ci = 64 / 128 and data gradients with Co = 64 / 128 ran the new kernels


def test_weight_stationary_variant_dispatch():
    """Shape-level and batch-level dispatch of variant 7: Ci == 64 on 8 x 32-divisible images; at least two tiles per persistent
    workgroup (4 x CUs tiles), else the 4-wave kernel; the statistics-row count is that of 8 x 32 tiles either way."""
    code = DTYPE_CODE[torch.bfloat16]
    assert lib.mi355_conv2d_igemm_variant(256, 256, 64, 256, 256, 64, 3, 3, 1, 1, -1, 1, 0, code) == 7
    assert lib.mi355_conv2d_igemm_variant(256, 256, 64, 256, 256, 128, 3, 3, 1, -1, 1, 1, 0, code) == 7           # data gradient 64 -> 128
    assert lib.mi355_conv2d_igemm_variant(256, 256, 128, 256, 256, 64, 3, 3, 1, 1, -1, 1, 0, code) == 8            # Ci = 128: the 4-row instantiation
    assert lib.mi355_conv2d_igemm_variant(64, 64, 128, 64, 64, 256, 3, 3, 1, 1, -1, 1, 0, code) == 8
    assert lib.mi355_conv2d_igemm_variant_n(32, 128, 128, 128, 128, 128, 128, 3, 3, 1, 1, -1, 1, 0, code) == 8
    assert lib.mi355_conv2d_igemm_variant_n(2, 64, 64, 128, 64, 64, 128, 3, 3, 1, 1, -1, 1, 0, code) == 2          # 128 tiles: too few
    assert lib.mi355_conv2d_igemm_variant_n(16, 128, 128, 128, 128, 128, 128, 3, 3, 1, 1, -1, 1, 0, code) == 2     # 8 tiles per workgroup: the 4-wave kernel
    assert lib.mi355_conv2d_igemm_variant_n(32, 64, 64, 128, 64, 64, 256, 3, 3, 1, 1, -1, 1, 0, code) == 2
    # the persistent kernel leaves ONE statistics row per workgroup range (2 per CU, shared by the channel tiles), not one per tile
    assert lib.mi355_conv2d_igemm_stat_rows(32, 128, 128, 128, 128, 128, 128, 3, 3, 1, 1, -1, 1, 0, code) == 256        # of 4096 tiles
    assert lib.mi355_conv2d_igemm_variant(16, 16, 64, 16, 16, 64, 3, 3, 1, 1, -1, 1, 0, code) == 3
    assert lib.mi355_conv2d_igemm_variant_n(32, 256, 256, 64, 256, 256, 64, 3, 3, 1, 1, -1, 1, 0, code) == 7
    assert lib.mi355_conv2d_igemm_variant_n(2, 64, 64, 64, 64, 64, 64, 3, 3, 1, 1, -1, 1, 0, code) == 2           # 32 tiles: too few
    assert lib.mi355_conv2d_igemm_stat_rows(32, 256, 256, 64, 256, 256, 64, 3, 3, 1, 1, -1, 1, 0, code) == 512           # of 8192 tiles
    assert lib.mi355_conv2d_igemm_variant_n(2, 64, 64, 64, 64, 64, 64, 3, 3, 1, 1, -1, 1, 0, DTYPE_CODE[torch.float32]) == 0


def test_weight_gradient_variant_dispatch():
    """mi355_conv2d_wgrad_variant reports the launcher's choice: the eight-wave nine-tap kernel for rows of 64-pixel segments and for
    32-pixel-wide images in pairs, the four-wave kernel for the other 32-pixel-segment shapes and 16-pixel-wide pairs, the generic
    split-K kernel for everything else (fp32, strides, odd shapes)."""
    code = DTYPE_CODE[torch.bfloat16]
    v = lambda n, h, w, k=3, s=1, p=1, c=code: lib.mi355_conv2d_wgrad_variant(n, h, w, k, k, s, p, c)
    assert v(32, 256, 256) == 3 and v(32, 64, 64) == 3 and v(32, 32, 32) == 4 and v(32, 16, 16) == 2
    assert v(3, 32, 32) == 1 and v(2, 32, 96) == 1 and v(3, 16, 16) == 0 and v(2, 12, 12) == 0
    assert v(2, 64, 64, 1, 1, 0) == 0 and v(2, 64, 64, 3, 2, 1) == 0 and v(2, 64, 64, c=DTYPE_CODE[torch.float32]) == 0


def test_counted_vmcnt_matches_drained_build(tmp_path):
    """The LDS-DMA kernels order their LDS reads behind the DMA writes with hand-COUNTED ``s_waitcnt vmcnt(N)`` (the DMA is issued from
    inline asm so that hipcc does not drain the queue in front of every LDS read, dma.hpp).  ``libmi355conv_drain.so`` is the same
    source with every counted wait replaced by vmcnt(0): forward, data-gradient and weight-gradient outputs of the two builds must
    be identical bit for bit, and identical across repeated launches — a wrong count is a race, and a race shows up as a difference."""
    import os, subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__))
    so = os.path.join(os.path.dirname(here), "medical-image-segmentation-and-classification_amd", "mi355")
    outs = {}
    for tag, libname in (("counted", "libmi355conv.so"), ("drained", "libmi355conv_drain.so")):
        path = os.path.join(so, libname)
        assert os.path.exists(path), f"{libname} not built (make -C csrc)"
        out = str(tmp_path / f"{tag}.npz")
        r = subprocess.run([sys.executable, os.path.join(here, "conv_dump_worker.py"), out], env=dict(os.environ, MI355_LIB=path),
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[tag] = np.load(out)
    a, b = outs["counted"], outs["drained"]
    assert set(a.files) == set(b.files) and len(a.files) >= 90
    for k in a.files:
        assert np.array_equal(a[k], b[k]), k
        if k.endswith("0"):
            assert np.array_equal(a[k], a[k[:-1] + "1"]) and np.array_equal(a[k], a[k[:-1] + "2"]), k


def test_buffer_descriptor_dma_zero_fills_out_of_range_lanes():
    """dma16_buf (csrc/dma.hpp): the LDS-DMA pieces of the 512-thread kernels go through a buffer descriptor, and padding is the
    hardware's range check — a lane with the always-out-of-range offset, or any lane under a num_records = 0 descriptor, must
    write ZEROS to its 16-byte LDS slot (not skip it), in-range lanes their data, and the scalar offset must move the source."""
    import ctypes, os
    here = os.path.dirname(os.path.abspath(__file__))
    so = os.path.join(os.path.dirname(here), "medical-image-segmentation-and-classification_amd", "mi355", "libmi355probe.so")
    assert os.path.exists(so), "libmi355probe.so not built (make -C csrc)"
    probe = ctypes.CDLL(so)
    probe.mi355_probe_bufdma.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_ulonglong, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    src = torch.arange(1, 1025, dtype=torch.int32, device=DEV)               # 4 KiB, no zero word anywhere
    out = torch.empty(256, dtype=torch.int32, device=DEV)

    def run(soff, mask, valid):
        assert probe.mi355_probe_bufdma(src.data_ptr(), soff, mask, valid, out.data_ptr(), None) == 0
        torch.cuda.synchronize()
        return out.cpu().view(64, 4)

    got = run(0, 0, 1)
    assert torch.equal(got, src[:256].cpu().view(64, 4))
    got = run(2048, 0, 1)                                                    # scalar offset: the second half of the buffer
    assert torch.equal(got, src[512:768].cpu().view(64, 4))
    mask = 0xF0F0_0000_0000_FF01
    got = run(1024, mask, 1)
    ref = src[256:512].cpu().view(64, 4).clone()
    for lane in range(64):
        if (mask >> lane) & 1:
            ref[lane] = 0
    assert torch.equal(got, ref)
    assert int(run(0, 0, 0).abs().sum()) == 0                                # num_records = 0: the whole piece is zeros
