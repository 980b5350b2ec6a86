"""-m gpu: every non-GEMM launcher of the C ABI against torch CPU fp32 (autograd where a
backward is involved).  fp32 tolerance 1e-5..1e-4 relative-to-max; bf16 inputs are rounded to
bf16 before the CPU reference, outputs allowed 2^-7 relative (one bf16 rounding of the result)."""
import math

import pytest
import torch
import torch.nn.functional as F

from gpu_util import DEV, lib, to_nhwc, from_nhwc, pack_w, rel_err, q, DTYPE_CODE

pytestmark = pytest.mark.gpu
DT = [torch.float32, torch.bfloat16, torch.float16]
TOL = {torch.float32: 2e-5, torch.bfloat16: 1.2e-2, torch.float16: 2e-3}


def dev(t):
    return t.to(DEV)


def _partials(m, c, nq=2):
    nb = lib.mi355_rowreduce_blocks(m)
    return nb, torch.full((nb * nq * c,), float("nan"), device=DEV)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("shape", [(2, 32, 9, 7), (3, 64, 16, 16), (2, 96, 5, 5), (1, 1024, 4, 4), (4, 8, 40, 40)])
def test_bn_forward_train(shape, dtype):
    n, c, h, w = shape
    if dtype == torch.float32 and c % 4 or dtype != torch.float32 and c % 8:
        pytest.skip("chunk multiple")
    g = torch.Generator().manual_seed(c)
    x = q(torch.randn(shape, generator=g) * 2 + 0.5, dtype)
    gamma = torch.rand(c, generator=g) + 0.5; beta = torch.randn(c, generator=g)
    rm = torch.randn(c, generator=g) * 0.1; rv = torch.rand(c, generator=g) + 0.5
    rm_ref, rv_ref = rm.clone(), rv.clone()
    ref = F.relu(F.batch_norm(x, rm_ref, rv_ref, gamma, beta, True, 0.1, 1e-5))
    m = n * h * w
    xd = to_nhwc(x, dtype)
    nb, part = _partials(m, c)
    lib.mi355_bn_stats(xd, part, m, c, c, DTYPE_CODE[dtype])
    rmd, rvd, nbt = dev(rm), dev(rv), torch.zeros((), dtype=torch.int64, device=DEV)
    sc, sh, mu, isd = (torch.empty(c, device=DEV) for _ in range(4))
    lib.mi355_bn_finalize(part, nb, m, c, dev(gamma), dev(beta), rmd, rvd, nbt, 0.1, 1e-5, sc, sh, mu, isd)
    y = torch.empty_like(xd)
    lib.mi355_bn_act(xd, c, sc, sh, None, 0, None, None, None, 0, y, c, m, c, 1, DTYPE_CODE[dtype])
    torch.cuda.synchronize()
    assert rel_err(from_nhwc(y), ref) < TOL[dtype]
    assert rel_err(rmd.cpu(), rm_ref) < 1e-5 and rel_err(rvd.cpu(), rv_ref) < 1e-5 and int(nbt) == 1
    assert rel_err(mu.cpu(), x.mean((0, 2, 3))) < 1e-5
    assert rel_err(isd.cpu(), 1 / torch.sqrt(x.var((0, 2, 3), unbiased=False) + 1e-5)) < 1e-5


@pytest.mark.parametrize("rows,rowlen,nsplit", [(8192, 128, 64), (2049, 130, 64), (100, 7, 3), (5, 64, 5)])
def test_fold_rows(rows, rowlen, nsplit):
    """mi355_fold_rows: out[j] = sum of the j-th band of ceil(rows / nsplit) partial rows (a short or empty last band too)."""
    g = torch.Generator().manual_seed(rows)
    part = torch.randn(rows, rowlen, generator=g)
    out = torch.full((nsplit, rowlen), float("nan"), device=DEV)
    lib.mi355_fold_rows(dev(part), rows, rowlen, out, nsplit)
    torch.cuda.synchronize()
    per = -(-rows // nsplit)
    ref = torch.stack([part[j * per:(j + 1) * per].double().sum(0) for j in range(nsplit)]).float()
    assert torch.allclose(out.cpu(), ref, rtol=1e-6, atol=1e-6)
    assert rel_err(out.cpu().double().sum(0), part.double().sum(0)) < 1e-6


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("act,with_res", [(1, False), (0, False), (1, True)])
@pytest.mark.parametrize("shape", [(3, 64, 10, 6), (5, 32, 72, 72)])      # 180 rows; 25 920 rows: 405 partial rows from a 256-workgroup grid
def test_bn_backward(dtype, act, with_res, shape):
    n, c, h, w = shape
    g = torch.Generator().manual_seed(5 + act)
    x = q(torch.randn(n, c, h, w, generator=g) * 1.5 + 0.3, dtype).requires_grad_(True)
    res = q(torch.randn(n, c, h, w, generator=g), dtype).requires_grad_(True)
    gamma = (torch.rand(c, generator=g) + 0.5).requires_grad_(True); beta = torch.randn(c, generator=g).requires_grad_(True)
    yb = F.batch_norm(x, None, None, gamma, beta, True, 0.1, 1e-5)
    y = yb + res if with_res else yb
    y = F.relu(y) if act else y
    dy = q(torch.randn(y.shape, generator=g), dtype)
    y.backward(dy)
    m = n * h * w
    xd, dyd, yd = to_nhwc(x.detach(), dtype), to_nhwc(dy, dtype), to_nhwc(y.detach(), dtype)
    mu = x.detach().mean((0, 2, 3)); isd = 1 / torch.sqrt(x.detach().var((0, 2, 3), unbiased=False) + 1e-5)
    nb, part = _partials(m, c)
    code = DTYPE_CODE[dtype]
    lib.mi355_bn_bwd_reduce(dyd, c, yd if with_res else None, c, xd, c, dev(mu), dev(isd), dev(gamma.detach() * isd), dev(beta.detach() - mu * gamma.detach() * isd), part, m, c, act, code)
    sums = torch.empty(2 * c, device=DEV); dgam = torch.ones(c, device=DEV); dbet = torch.ones(c, device=DEV)
    lib.mi355_bn_bwd_finalize(part, nb, c, sums, dgam, dbet, 1.0)      # accumulate onto ones
    dx = torch.empty_like(xd); dres = torch.empty_like(xd)
    post0 = q(torch.randn(n, c, h, w, generator=g), dtype)      # an operand added after the activation: dpost += dy in the same pass
    dpost = to_nhwc(post0, dtype)
    nb1, p1 = _partials(m, c, 1)
    lib.mi355_bn_bwd_apply(dyd, c, yd if with_res else None, c, xd, c, dev(gamma.detach()), dev(mu), dev(isd),
                           dev(gamma.detach() * isd), dev(beta.detach() - mu * gamma.detach() * isd), sums, dx, c,
                           dres if with_res else None, c, dpost, c, 1, p1, m, c, act, code)
    dbias = torch.empty(c, device=DEV)
    lib.mi355_colsum_finalize(p1, nb1, 1, c, dbias, 0.0)
    torch.cuda.synchronize()
    tol = 5e-5 if dtype == torch.float32 else 2e-2
    assert rel_err(from_nhwc(dx), x.grad) < tol
    assert rel_err(dgam.cpu() - 1, gamma.grad) < tol and rel_err(dbet.cpu() - 1, beta.grad) < tol
    if with_res:
        assert rel_err(from_nhwc(dres), res.grad) < tol
    assert rel_err(from_nhwc(dpost), q(post0 + q(dy, dtype), dtype)) < (1e-6 if dtype == torch.float32 else 1e-2)
    assert float(dbias.abs().max()) < 1e-2 * float(x.grad.abs().sum((0, 2, 3)).max())   # ~0: dx is mean-free


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("n_ex,post_acc,with_y", [(1, 0, False), (4, 1, False), (3, 1, True), (2, 0, True)])
def test_bn_backward_apply_with_the_earlier_gradients_summed(dtype, n_ex, post_acc, with_y):
    """mi355_bn_bwd_apply_post4 (the recurrent block's last application, R2AttU_Net.py:41-44): dx bit-identical to mi355_bn_bwd_apply
    on the same operands, dpost = round(dy [+ dpost] + ex0 + .. ) summed in fp32 in that order and rounded once — also with the
    activated tensor as the mask source (`y`), fewer than four extras, 3 x 37 x 41 rows (ragged batches) and 96 channels."""
    n, c, h, w = 3, 96, 37, 41
    g = torch.Generator().manual_seed(11 + n_ex)
    m = n * h * w
    code = DTYPE_CODE[dtype]
    x = q(torch.randn(n, c, h, w, generator=g) * 1.5 + 0.3, dtype)
    dy = q(torch.randn(n, c, h, w, generator=g), dtype)
    ex = [q(torch.randn(n, c, h, w, generator=g), dtype) for _ in range(n_ex)]
    old = q(torch.randn(n, c, h, w, generator=g), dtype)
    gamma = torch.rand(c, generator=g) + 0.5; beta = torch.randn(c, generator=g)
    mu = x.mean((0, 2, 3)); isd = 1 / torch.sqrt(x.var((0, 2, 3), unbiased=False) + 1e-5)
    sc, sh = gamma * isd, beta - mu * gamma * isd
    y = q(F.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)), dtype)
    xd, dyd, yd = to_nhwc(x, dtype), to_nhwc(dy, dtype), to_nhwc(y, dtype)
    exd = [to_nhwc(e, dtype) for e in ex] + [None] * (4 - n_ex)
    nb, part = _partials(m, c)
    ya = yd if with_y else None
    lib.mi355_bn_bwd_reduce(dyd, c, ya, c, xd, c, dev(mu), dev(isd), dev(sc), dev(sh), part, m, c, 1, code)
    sums = torch.empty(2 * c, device=DEV)
    lib.mi355_bn_bwd_finalize(part, nb, c, sums, None, None, 0.0)
    dx_ref, dx = torch.empty_like(xd), torch.empty_like(xd)
    lib.mi355_bn_bwd_apply(dyd, c, ya, c, xd, c, dev(gamma), dev(mu), dev(isd), dev(sc), dev(sh), sums, dx_ref, c, None, 0, None, 0, 0,
                           None, m, c, 1, code)
    dpost = to_nhwc(old, dtype)
    lib.mi355_bn_bwd_apply_post4(dyd, c, ya, c, xd, c, dev(gamma), dev(mu), dev(isd), dev(sc), dev(sh), sums, dx, c, dpost, c, post_acc,
                                 exd[0], exd[1], exd[2], exd[3], c, m, c, 1, code)
    torch.cuda.synchronize()
    assert torch.equal(dx, dx_ref)
    want = dy.float()
    if post_acc:
        want = want + old.float()
    for e in ex:
        want = want + e.float()
    assert torch.equal(from_nhwc(dpost).float(), want.to(dtype).float())
    with pytest.raises(RuntimeError):           # a gap in the earlier gradients (ex1 without ex0 is refused at the entry point)
        lib.mi355_bn_bwd_apply_post4(dyd, c, ya, c, xd, c, dev(gamma), dev(mu), dev(isd), dev(sc), dev(sh), sums, dx, c, dpost, c, 0,
                                     None, exd[0], None, None, c, m, c, 1, code)


@pytest.mark.parametrize("dtype", DT)
def test_bn_act_two_operands_and_eval_coeffs(dtype):
    n, c, h, w = 2, 32, 6, 6
    g = torch.Generator().manual_seed(2)
    a = q(torch.randn(n, c, h, w, generator=g), dtype); b = q(torch.randn(n, c, h, w, generator=g), dtype)
    gam, bet, rm = (torch.randn(c, generator=g) for _ in range(3)); rv = torch.rand(c, generator=g) + 0.5
    sc, sh = torch.empty(c, device=DEV), torch.empty(c, device=DEV)
    lib.mi355_bn_eval_coeffs(dev(gam), dev(bet), dev(rm), dev(rv), 1e-5, c, sc, sh)
    s2, t2 = torch.randn(c, generator=g), torch.randn(c, generator=g)
    ref = F.relu(F.batch_norm(a, rm, rv, gam, bet, False, 0.1, 1e-5) + b * s2[None, :, None, None] + t2[None, :, None, None])
    y = torch.empty(n, h, w, 2 * c, dtype=dtype, device=DEV)          # write into a channel slice
    lib.mi355_bn_act(to_nhwc(a, dtype), c, sc, sh, to_nhwc(b, dtype), c, dev(s2), dev(t2), None, 0,
                     y.data_ptr() + c * y.element_size(), 2 * c, n * h * w, c, 1, DTYPE_CODE[dtype])
    torch.cuda.synchronize()
    assert rel_err(from_nhwc(y[..., c:]), ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("k,s,p,hw", [(2, 2, 0, 12), (3, 2, 1, 13), (3, 2, 1, 16)])
def test_maxpool(dtype, k, s, p, hw):
    n, c = 2, 32
    g = torch.Generator().manual_seed(k * 10 + hw)
    # coarse grid of values => many exact ties, exercising torch's first-max rule
    x = (torch.randint(-3, 4, (n, c, hw, hw), generator=g).float() / 2).requires_grad_(True)
    y = F.max_pool2d(x, k, s, p)
    dy = q(torch.randn(y.shape, generator=g), dtype)
    y.backward(dy)
    code = DTYPE_CODE[dtype]
    xd = to_nhwc(x.detach(), dtype)
    yd = torch.empty(n, y.shape[2], y.shape[3], c, dtype=dtype, device=DEV)
    lib.mi355_maxpool_fwd(xd, c, yd, c, n, hw, hw, c, k, s, p, code)
    dx = torch.full_like(xd, 1.0)
    lib.mi355_maxpool_bwd(xd, c, to_nhwc(dy, dtype), c, dx, c, n, hw, hw, c, k, s, p, 1, code)   # accumulate onto 1
    torch.cuda.synchronize()
    assert torch.equal(from_nhwc(yd), y.detach())
    assert rel_err(from_nhwc(dx) - 1, x.grad) < TOL[dtype]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("act", [0, 1])
def test_bn_act_with_fused_maxpool_is_bit_identical_to_the_two_passes(dtype, act):
    """mi355_bn_act_pool2 (BatchNorm apply + ReLU + the MaxPool2d(2, 2) of AttentionUNet.py:61,89-95 in one pass) against
    mi355_bn_act followed by mi355_maxpool_fwd: activation and pooled tensor bit for bit, both written into channel slices of
    wider buffers; and against torch on the CPU."""
    n, c, h, w = 3, 40, 10, 12
    g = torch.Generator().manual_seed(4 + act)
    x = q(torch.randn(n, c, h, w, generator=g), dtype)
    sc, sh = torch.randn(c, generator=g), torch.randn(c, generator=g)
    code = DTYPE_CODE[dtype]
    xd = to_nhwc(x, dtype)
    es = xd.element_size()
    outs = []
    for fused in (0, 1):
        y = torch.zeros(n, h, w, c + 8, dtype=dtype, device=DEV)
        p = torch.zeros(n, h // 2, w // 2, c + 16, dtype=dtype, device=DEV)
        if fused:
            lib.mi355_bn_act_pool2(xd, c, dev(sc), dev(sh), y.data_ptr() + 8 * es, c + 8, p.data_ptr() + 16 * es, c + 16, n, h, w, c, act, code)
        else:
            lib.mi355_bn_act(xd, c, dev(sc), dev(sh), None, 0, None, None, None, 0, y.data_ptr() + 8 * es, c + 8, n * h * w, c, act, code)
            lib.mi355_maxpool_fwd(y.data_ptr() + 8 * es, c + 8, p.data_ptr() + 16 * es, c + 16, n, h, w, c, 2, 2, 0, code)
        torch.cuda.synchronize()
        outs.append((y.cpu(), p.cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    ref = x * sc[None, :, None, None] + sh[None, :, None, None]
    ref = F.relu(ref) if act else ref
    assert rel_err(from_nhwc(outs[1][0][..., 8:]), ref) < TOL[dtype]
    assert rel_err(from_nhwc(outs[1][1][..., 16:]), F.max_pool2d(q(ref, dtype), 2, 2)) < TOL[dtype]
    assert float(outs[1][0][..., :8].abs().sum()) == 0 and float(outs[1][1][..., :16].abs().sum()) == 0
    # without a pooled output (p = NULL): the activation alone, the same bits
    y = torch.zeros(n, h, w, c + 8, dtype=dtype, device=DEV)
    lib.mi355_bn_act_pool2(xd, c, dev(sc), dev(sh), y.data_ptr() + 8 * es, c + 8, None, 0, n, h, w, c, act, code)
    torch.cuda.synchronize()
    assert torch.equal(y.cpu(), outs[0][0])
    # mi355_bn_act_windows: the same kernel with mi355_bn_act's residual operand (before the activation / after it: act bit 1)
    r = q(torch.randn(n, c, h, w, generator=g), dtype)
    rd = to_nhwc(r, dtype)
    for a2 in (act, act | 2):
        ya = torch.zeros(n, h, w, c, dtype=dtype, device=DEV); yb = torch.zeros_like(ya)
        lib.mi355_bn_act(xd, c, dev(sc), dev(sh), None, 0, None, None, rd, c, ya, c, n * h * w, c, a2, code)
        lib.mi355_bn_act_windows(xd, c, dev(sc), dev(sh), rd, c, yb, c, n, h, w, c, a2, code)
        torch.cuda.synchronize()
        assert torch.equal(ya, yb), a2


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("shape", [(3, 64, 8, 64), (2, 128, 6, 32)])
def test_bn_backward_when_the_pooling_is_the_only_consumer(shape, dtype):
    """dy = NULL in mi355_bn_bwd_reduce_pool2 / _apply_pool2: the activation feeds the MaxPool2d(2, 2) ONLY (VGG.py's feature
    stack), so its gradient is the routed dp alone — against mi355_maxpool_bwd (overwriting) + the row-ordered passes."""
    n, c, h, w = shape
    code = DTYPE_CODE[dtype]
    if not lib.mi355_bn_bwd_pool2_ok(h, w, c, code):
        pytest.skip("row too short for this channel count")
    g = torch.Generator().manual_seed(c)
    m = n * h * w
    x = q(torch.round(torch.randn(n, c, h, w, generator=g) * 2) / 2, dtype)
    dp = q(torch.randn(n, c, h // 2, w // 2, generator=g), dtype)
    gamma = torch.rand(c, generator=g) + 0.5
    mean = x.float().mean((0, 2, 3)); invstd = (x.float().var((0, 2, 3), unbiased=False) + 1e-5).rsqrt()
    beta = torch.randn(c, generator=g) * 0.3
    D = [dev(t.float().contiguous()) for t in (gamma, mean, invstd, gamma * invstd, beta - mean * gamma * invstd)]
    xd, dpd = to_nhwc(x, dtype), to_nhwc(dp, dtype)
    res = []
    for fused in (0, 1):
        dx = torch.zeros(n, h, w, c, dtype=dtype, device=DEV)
        nb, part = _partials(m, c)
        sums = torch.empty(2 * c, device=DEV); dg = torch.empty(c, device=DEV); db = torch.empty(c, device=DEV)
        if fused:
            lib.mi355_bn_bwd_reduce_pool2(None, 0, dpd, c, xd, c, D[1], D[2], D[3], D[4], part, n, h, w, c, code)
            lib.mi355_bn_bwd_finalize(part, min(nb, lib.mi355_bn_bwd_reduce_pool2_rows(m)), c, sums, dg, db, 0.0)
            lib.mi355_bn_bwd_apply_pool2(None, 0, dpd, c, xd, c, D[0], D[1], D[2], D[3], D[4], sums, dx, c, n, h, w, c, code)
        else:
            a = torch.empty(n, h, w, c, dtype=dtype, device=DEV); da = torch.full_like(a, float("nan"))
            lib.mi355_bn_act(xd, c, D[3], D[4], None, 0, None, None, None, 0, a, c, m, c, 1, code)
            lib.mi355_maxpool_bwd(a, c, dpd, c, da, c, n, h, w, c, 2, 2, 0, 0, code)
            lib.mi355_bn_bwd_reduce(da, c, None, 0, xd, c, D[1], D[2], D[3], D[4], part, m, c, 1, code)
            lib.mi355_bn_bwd_finalize(part, min(nb, lib.mi355_bn_bwd_reduce_rows(m)), c, sums, dg, db, 0.0)
            lib.mi355_bn_bwd_apply(da, c, None, 0, xd, c, D[0], D[1], D[2], D[3], D[4], sums, dx, c, None, 0, None, 0, 0, None, m, c, 1, code)
        torch.cuda.synchronize()
        res.append((dx.float().cpu(), sums.cpu(), dg.cpu(), db.cpu()))
    tol = 1e-5 if dtype == torch.float32 else TOL[dtype]
    for i in range(4):
        assert rel_err(res[1][i], res[0][i]) < (tol if i else 2 * tol), i


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("shape", [(3, 64, 8, 64), (2, 128, 6, 32), (2, 256, 4, 32), (1, 32, 4, 256)])
def test_bn_backward_with_lazy_maxpool_gradient(shape, dtype):
    """mi355_bn_bwd_reduce_pool2 / _apply_pool2 (the gradient of the MaxPool2d(2, 2) of AttentionUNet.py:61,89-95 added inside the
    two BatchNorm backward passes of the layer that produced the pooled activation) against the separate passes — mi355_bn_act,
    mi355_maxpool_bwd accumulating into da, mi355_bn_bwd_reduce / _apply — through the same finalize, with operands in channel
    slices of wider buffers.  The raw tensor is quantised so that windows TIE (also at zero, behind the ReLU): torch's first-maximum
    rule.  fp32: the sums of the two orders agree to rounding; 2-byte: the separate passes round da + scatter to the storage type
    first, the fused ones do not."""
    n, c, h, w = shape
    epc = 4 if dtype == torch.float32 else 8
    if not lib.mi355_bn_bwd_pool2_ok(h, w, c, DTYPE_CODE[dtype]):
        pytest.skip("row too short for this channel count")
    g = torch.Generator().manual_seed(c + w)
    code = DTYPE_CODE[dtype]
    m = n * h * w
    x = q(torch.round(torch.randn(n, c, h, w, generator=g) * 2) / 2, dtype)             # raw convolution output, many equal values
    da = q(torch.randn(n, c, h, w, generator=g), dtype)
    dp = q(torch.randn(n, c, h // 2, w // 2, generator=g), dtype)
    gamma = torch.rand(c, generator=g) + 0.5
    mean = x.float().mean((0, 2, 3)); var = x.float().var((0, 2, 3), unbiased=False)
    invstd = (var + 1e-5).rsqrt()
    beta = torch.randn(c, generator=g) * 0.3
    sc, sh = gamma * invstd, beta - mean * gamma * invstd
    # wider buffers: x at channel offset 8 of c + 8, da at 16 of c + 16, dp at 0 of c + 8, dx at 8 of c + 24
    def wide(t, off, tot):
        b = torch.zeros(t.shape[0], t.shape[2], t.shape[3], tot, dtype=dtype, device=DEV)
        b[..., off:off + c] = to_nhwc(t, dtype)
        return b
    xb, dab0, dpb = wide(x, 8, c + 8), wide(da, 16, c + 16), wide(dp, 0, c + 8)
    es = xb.element_size()
    xp, dpp = xb.data_ptr() + 8 * es, dpb.data_ptr()
    D = [dev(t.float().contiguous()) for t in (gamma, mean, invstd, sc, sh)]
    res = []
    for fused in (0, 1):
        dab = dab0.clone()
        dap = dab.data_ptr() + 16 * es
        dxb = torch.zeros(n, h, w, c + 24, dtype=dtype, device=DEV)
        dxp = dxb.data_ptr() + 8 * es
        nb, part = _partials(m, c)
        sums = torch.empty(2 * c, device=DEV); dg = torch.empty(c, device=DEV); db = torch.empty(c, device=DEV)
        if fused:
            lib.mi355_bn_bwd_reduce_pool2(dap, c + 16, dpp, c + 8, xp, c + 8, D[1], D[2], D[3], D[4], part, n, h, w, c, code)
            lib.mi355_bn_bwd_finalize(part, min(nb, lib.mi355_bn_bwd_reduce_pool2_rows(m)), c, sums, dg, db, 0.0)
            lib.mi355_bn_bwd_apply_pool2(dap, c + 16, dpp, c + 8, xp, c + 8, D[0], D[1], D[2], D[3], D[4], sums, dxp, c + 24, n, h, w, c, code)
        else:
            a = torch.empty(n, h, w, c, dtype=dtype, device=DEV)
            lib.mi355_bn_act(xp, c + 8, D[3], D[4], None, 0, None, None, None, 0, a, c, m, c, 1, code)
            lib.mi355_maxpool_bwd(a, c, dpp, c + 8, dap, c + 16, n, h, w, c, 2, 2, 0, 1, code)
            lib.mi355_bn_bwd_reduce(dap, c + 16, None, 0, xp, c + 8, D[1], D[2], D[3], D[4], part, m, c, 1, code)
            lib.mi355_bn_bwd_finalize(part, min(nb, lib.mi355_bn_bwd_reduce_rows(m)), c, sums, dg, db, 0.0)
            lib.mi355_bn_bwd_apply(dap, c + 16, None, 0, xp, c + 8, D[0], D[1], D[2], D[3], D[4], sums, dxp, c + 24, None, 0, None, 0, 0,
                                   None, m, c, 1, code)
        torch.cuda.synchronize()
        if fused:
            assert torch.equal(dab, dab0)                       # the incoming gradient is only read
        assert float(dxb[..., :8].abs().sum()) == 0 and float(dxb[..., 8 + c:].abs().sum()) == 0
        res.append((dxb[..., 8:8 + c].float().cpu(), sums.cpu(), dg.cpu(), db.cpu()))
    # torch: the composite on the CPU in fp32 (activation rounded to the storage type as the forward stores it)
    xr = x.float().clone().requires_grad_(True)
    act = F.relu(xr * sc[None, :, None, None] + sh[None, :, None, None])
    # (BatchNorm with batch statistics: differentiate through mean / var of x)
    xhat = (xr - xr.mean((0, 2, 3), keepdim=True)) * (xr.var((0, 2, 3), unbiased=False, keepdim=True) + 1e-5).rsqrt()
    act2 = F.relu(xhat * gamma[None, :, None, None] + beta[None, :, None, None])
    a_q = q(act.detach(), dtype)
    aq = a_q.clone().requires_grad_(True)
    F.max_pool2d(aq, 2, 2).backward(dp.float())
    gtot = (da.float() + aq.grad)                                 # gradient reaching the activation
    act2.backward(gtot)
    tol = TOL[dtype]
    for r in res:
        assert rel_err(from_nhwc(r[0]), xr.grad) < (tol if dtype == torch.float32 else 2 * tol)
    assert rel_err(res[1][1], res[0][1]) < (1e-5 if dtype == torch.float32 else tol)
    assert rel_err(res[1][0], res[0][0]) < (1e-5 if dtype == torch.float32 else 2 * tol)
    assert rel_err(res[1][2], res[0][2]) < (1e-5 if dtype == torch.float32 else tol)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("shape", [(2, 32, 16, 24), (1, 64, 8, 8), (3, 256, 5, 4)])
def test_gate_branches_backward_in_two_passes(shape, dtype):
    """mi355_gate_bn_bwd_reduce / _apply (+ mi355_bn_bwd_finalize_at, mi355_colsum_finalize): the backward of
    psi_in = relu(BN_g(g1) + BN_x(x1)), z = psi conv (AttentionUNet.py:32-38,48-52) without the stored gradient of psi_in, against
    the separate passes (mi355_rowdot_bwd + two BatchNorm backward chains) and against torch autograd on the CPU."""
    n, c, h, w = shape
    g = torch.Generator().manual_seed(c + h)
    code = DTYPE_CODE[dtype]
    m = n * h * w
    g1 = q(torch.randn(n, c, h, w, generator=g) * 1.5 + 0.3, dtype)
    x1 = q(torch.randn(n, c, h, w, generator=g) - 0.2, dtype)
    dz = torch.randn(m, generator=g)
    wpsi = torch.randn(c, generator=g) * 0.5
    gam = [torch.rand(c, generator=g) + 0.5 for _ in range(2)]
    bet = [torch.randn(c, generator=g) * 0.3 for _ in range(2)]
    co = []
    for t, ga, be in ((g1, gam[0], bet[0]), (x1, gam[1], bet[1])):
        mean = t.mean((0, 2, 3)); invstd = (t.var((0, 2, 3), unbiased=False) + 1e-5).rsqrt()
        co += [ga * invstd, be - mean * ga * invstd, mean, invstd]
    D = [dev(t.float().contiguous()) for t in co]
    g1d, x1d = to_nhwc(g1, dtype), to_nhwc(x1, dtype)
    dzd, wd = dev(dz), dev(wpsi)
    gd = [dev(t) for t in gam]
    nb = lib.mi355_rowreduce_blocks(m)
    res = []
    for fused in (0, 1):
        dg = torch.zeros(n, h, w, c + 8, dtype=dtype, device=DEV)
        dx = torch.zeros(n, h, w, c, dtype=dtype, device=DEV)
        es = dg.element_size()
        dw = torch.empty(c, device=DEV); db = torch.empty(1, device=DEV)
        sm = [torch.empty(2 * c, device=DEV) for _ in range(2)]
        dga = [torch.empty(c, device=DEV) for _ in range(2)]; dbe = [torch.empty(c, device=DEV) for _ in range(2)]
        zz = torch.empty(m, device=DEV); zp = torch.full((nb * 2,), float("nan"), device=DEV)
        bpsi = dev(torch.tensor([0.37]))
        if fused:
            lib.mi355_gate_psi_fwd(g1d, c, x1d, c, D[0], D[1], D[4], D[5], wd, bpsi, zz, zp, m, c, code)
            part = torch.full((nb * 5 * c,), float("nan"), device=DEV)
            lib.mi355_gate_bn_bwd_reduce(dzd, g1d, c, x1d, c, *D, wd, part, m, c, code)
            lib.mi355_colsum_finalize(part.data_ptr() + 3 * c * 4, nb, 5, c, dw, 0.0)
            lib.mi355_colsum_finalize(part.data_ptr() + 4 * c * 4, nb, 5 * c, 1, db, 0.0)
            nbf = min(nb, lib.mi355_gate_bn_bwd_reduce_rows(m))
            for i in range(2):
                lib.mi355_bn_bwd_finalize_at(part, nbf, 5, 0, 1 + i, c, sm[i], dga[i], dbe[i], 0.0)
            lib.mi355_gate_bn_bwd_apply(dzd, g1d, c, x1d, c, *D, wd, gd[0], gd[1], sm[0], sm[1], dg.data_ptr() + 8 * es, c + 8, dx, c, m, c, code)
        else:
            p = torch.empty(n, h, w, c, dtype=dtype, device=DEV)
            lib.mi355_bn_act(g1d, c, D[0], D[1], x1d, c, D[4], D[5], None, 0, p, c, m, c, 1, code)
            lib.mi355_rowdot_fwd(p, c, wd, bpsi, zz, zp, m, c, 0, 1, code)
            dp = torch.empty_like(p)
            part = torch.full((nb * 2 * c,), float("nan"), device=DEV)
            lib.mi355_rowdot_bwd(dzd, p, c, wd, dp, c, part, m, c, 1, 0, 1, 0, code)
            lib.mi355_colsum_finalize(part, nb, 2, c, dw, 0.0)
            lib.mi355_colsum_finalize(part.data_ptr() + c * 4, nb, 2 * c, 1, db, 0.0)
            for i, (t, out, ld) in enumerate(((g1d, dg.data_ptr() + 8 * es, c + 8), (x1d, dx, c))):
                part2 = torch.full((nb * 2 * c,), float("nan"), device=DEV)
                lib.mi355_bn_bwd_reduce(dp, c, None, 0, t, c, D[4 * i + 2], D[4 * i + 3], D[4 * i], D[4 * i + 1], part2, m, c, 0, code)
                lib.mi355_bn_bwd_finalize(part2, min(nb, lib.mi355_bn_bwd_reduce_rows(m)), c, sm[i], dga[i], dbe[i], 0.0)
                lib.mi355_bn_bwd_apply(dp, c, None, 0, t, c, gd[i], D[4 * i + 2], D[4 * i + 3], D[4 * i], D[4 * i + 1], sm[i], out, ld,
                                       None, 0, None, 0, 0, None, m, c, 0, code)
        torch.cuda.synchronize()
        assert float(dg[..., :8].abs().sum()) == 0
        res.append([from_nhwc(dg[..., 8:].float().cpu()), from_nhwc(dx.float().cpu()), dw.cpu(), db.cpu(), dga[0].cpu(), dbe[0].cpu(),
                    dga[1].cpu(), dbe[1].cpu(), zz.cpu(), zp.cpu()])
    # the forward: z and its (sum, sum of squares) partials from one pass as from two — to the last bits (the compiler fuses the
    # multiply-adds of the dot product in one kernel and not in the other)
    assert rel_err(res[1][8], res[0][8]) < 1e-6 and rel_err(res[1][9], res[0][9]) < 1e-6
    # torch: autograd through both BatchNorms (batch statistics), the sum, the ReLU and the one-channel convolution
    leaves = [t.clone().requires_grad_(True) for t in (g1, x1, wpsi, torch.zeros(1), gam[0], bet[0], gam[1], bet[1])]
    a = F.batch_norm(leaves[0], None, None, leaves[4], leaves[5], True, 0.1, 1e-5) + \
        F.batch_norm(leaves[1], None, None, leaves[6], leaves[7], True, 0.1, 1e-5)
    pr = F.relu(a)
    z = (q(pr.detach(), dtype) + (pr - pr.detach())).permute(0, 2, 3, 1).reshape(m, c) @ leaves[2] + leaves[3]
    (z * dz).sum().backward()
    assert rel_err(res[1][8], z.detach() + 0.37) < 2 * (TOL[dtype] if dtype != torch.float32 else 1e-5)
    ref = [l.grad for l in leaves]
    tol = TOL[dtype] if dtype != torch.float32 else 1e-4
    for i in range(8):
        for r in res:
            assert rel_err(r[i], ref[i]) < 2 * tol, (i, rel_err(r[i], ref[i]))
        # the gradient of psi_in is rounded to the storage type exactly as the separate pass stores it: the branch gradients and
        # the BatchNorm parameter gradients are the SAME numbers (same row order, same arithmetic); the psi weight's gradient is
        # summed over another grid
        # (fp32 and bf16; the fp16 instantiations of the two kernel families contract the apply arithmetic differently)
        if i in (0, 1, 4, 5, 6, 7) and dtype != torch.float16:
            assert torch.equal(res[1][i], res[0][i]), i
        else:
            assert rel_err(res[1][i], res[0][i]) < (2e-5 if i in (2, 3) or dtype == torch.float32 else tol), i


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("shape", [(2, 64, 16, 24), (3, 32, 5, 4)])
def test_logit_head_behind_batchnorm_relu_in_one_pass_each_way(shape, dtype):
    """The same kernels with ONE normalised operand (x1 = NULL): a = relu(BN(y)) in front of the one-channel logit convolution
    (AttentionUNet.py:84) — forward z = w . a + b without storing a, backward of the BatchNorm and of the convolution's weight
    and bias without storing a's gradient — against mi355_bn_act + mi355_rowdot_fwd / mi355_rowdot_bwd + mi355_bn_bwd_reduce /
    _apply, and against torch autograd."""
    n, c, h, w = shape
    g = torch.Generator().manual_seed(c + w)
    code = DTYPE_CODE[dtype]
    m = n * h * w
    y = q(torch.round(torch.randn(n, c, h, w, generator=g) * 4) / 4 + 0.125, dtype)
    dz = torch.randn(m, generator=g)
    wh = torch.randn(c, generator=g) * 0.5
    gam = torch.rand(c, generator=g) + 0.5; bet = torch.randn(c, generator=g) * 0.3
    mean = y.mean((0, 2, 3)); invstd = (y.var((0, 2, 3), unbiased=False) + 1e-5).rsqrt()
    D = [dev(t.float().contiguous()) for t in (gam * invstd, bet - mean * gam * invstd, mean, invstd)]
    yd, dzd, wd, gd, bh = to_nhwc(y, dtype), dev(dz), dev(wh), dev(gam), dev(torch.tensor([-0.21]))
    nb = lib.mi355_rowreduce_blocks(m)
    res = []
    for fused in (0, 1):
        dy = torch.zeros(n, h, w, c, dtype=dtype, device=DEV)
        zz = torch.empty(m, device=DEV)
        dw = torch.empty(c, device=DEV); db = torch.empty(1, device=DEV); sm = torch.empty(2 * c, device=DEV)
        dga = torch.empty(c, device=DEV); dbe = torch.empty(c, device=DEV)
        if fused:
            lib.mi355_gate_psi_fwd(yd, c, None, 0, D[0], D[1], None, None, wd, bh, zz, None, m, c, code)
            part = torch.full((nb * 5 * c,), float("nan"), device=DEV)
            lib.mi355_gate_bn_bwd_reduce(dzd, yd, c, None, 0, D[0], D[1], D[2], D[3], None, None, None, None, wd, part, m, c, code)
            lib.mi355_colsum_finalize(part.data_ptr() + 3 * c * 4, nb, 5, c, dw, 0.0)
            lib.mi355_colsum_finalize(part.data_ptr() + 4 * c * 4, nb, 5 * c, 1, db, 0.0)
            lib.mi355_bn_bwd_finalize_at(part, min(nb, lib.mi355_gate_bn_bwd_reduce_rows(m)), 5, 0, 1, c, sm, dga, dbe, 0.0)
            lib.mi355_gate_bn_bwd_apply(dzd, yd, c, None, 0, D[0], D[1], D[2], D[3], None, None, None, None, wd, gd, None, sm, None,
                                        dy, c, None, 0, m, c, code)
        else:
            a = torch.empty(n, h, w, c, dtype=dtype, device=DEV)
            lib.mi355_bn_act(yd, c, D[0], D[1], None, 0, None, None, None, 0, a, c, m, c, 1, code)
            lib.mi355_rowdot_fwd(a, c, wd, bh, zz, None, m, c, 0, 1, code)
            da = torch.empty_like(a)
            part = torch.full((nb * 2 * c,), float("nan"), device=DEV)
            lib.mi355_rowdot_bwd(dzd, a, c, wd, da, c, part, m, c, 0, 0, 1, 0, code)
            lib.mi355_colsum_finalize(part, nb, 2, c, dw, 0.0)
            lib.mi355_colsum_finalize(part.data_ptr() + c * 4, nb, 2 * c, 1, db, 0.0)
            part2 = torch.full((nb * 2 * c,), float("nan"), device=DEV)
            lib.mi355_bn_bwd_reduce(da, c, None, 0, yd, c, D[2], D[3], D[0], D[1], part2, m, c, 1, code)
            lib.mi355_bn_bwd_finalize(part2, min(nb, lib.mi355_bn_bwd_reduce_rows(m)), c, sm, dga, dbe, 0.0)
            lib.mi355_bn_bwd_apply(da, c, None, 0, yd, c, gd, D[2], D[3], D[0], D[1], sm, dy, c, None, 0, None, 0, 0, None, m, c, 1, code)
        torch.cuda.synchronize()
        res.append([from_nhwc(dy.float().cpu()), dw.cpu(), db.cpu(), dga.cpu(), dbe.cpu(), zz.cpu()])
    leaves = [t.clone().requires_grad_(True) for t in (y, wh, torch.zeros(1), gam, bet)]
    pr = F.relu(F.batch_norm(leaves[0], None, None, leaves[3], leaves[4], True, 0.1, 1e-5))
    z = (q(pr.detach(), dtype) + (pr - pr.detach())).permute(0, 2, 3, 1).reshape(m, c) @ leaves[1] + leaves[2]
    (z * dz).sum().backward()
    tol = TOL[dtype] if dtype != torch.float32 else 1e-4
    assert rel_err(res[1][5], z.detach() - 0.21) < 2 * (TOL[dtype] if dtype != torch.float32 else 1e-5)
    for i in range(5):
        for r in res:
            assert rel_err(r[i], leaves[i].grad) < 2 * tol, (i, rel_err(r[i], leaves[i].grad))
        assert rel_err(res[1][i], res[0][i]) < (2e-5 if dtype == torch.float32 else tol), i
    assert rel_err(res[1][5], res[0][5]) < 1e-6


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("shape", [(2, 3, 9, 7), (1, 1, 4, 16), (3, 2, 16, 8)])
def test_input_im2col_and_the_stem_as_a_pointwise_convolution(shape, dtype):
    """mi355_pack_input_im2col3 against F.unfold (channel c * 9 + kh * 3 + kw, zero outside the image, zero beyond 9 C), and the
    stem Conv2d(C, Co, 3, 1, 1) computed as the pointwise convolution of that tensor with the SAME weight memory
    ([Co][C][3][3] read as [Co][9 C][1][1]) against F.conv2d (AttentionUNet.py:6,60)."""
    n, c, h, w = shape
    g = torch.Generator().manual_seed(h * w)
    x = torch.randn(n, c, h, w, generator=g)
    code = DTYPE_CODE[dtype]
    y = torch.full((n, h, w, 32), float("nan"), dtype=dtype, device=DEV)
    lib.mi355_pack_input_im2col3(dev(x), y, n, c, h, w, code)
    torch.cuda.synchronize()
    ref = F.unfold(x, 3, padding=1).view(n, c * 9, h, w)                      # channel order (c, kh, kw)
    got = y.float().cpu().permute(0, 3, 1, 2)
    assert torch.equal(got[:, :c * 9], q(ref, dtype)) and float(got[:, c * 9:].abs().sum()) == 0
    if dtype == torch.float32:
        return
    co = 64
    wt = torch.randn(co, c, 3, 3, generator=g) / (c * 9) ** 0.5
    b = torch.randn(co, generator=g)
    wf, _ = pack_w(wt.view(co, c * 9, 1, 1), dtype, cip=32)
    out = torch.empty(n, h, w, co, dtype=dtype, device=DEV)
    lib.mi355_conv2d_igemm(y, wf, dev(b), out, n, h, w, 32, 32, h, w, co, co, 1, 1, 1, 1, 0, 1, 0, 0, None, code)
    torch.cuda.synchronize()
    assert rel_err(from_nhwc(out), F.conv2d(q(x, dtype), q(wt, dtype), b, padding=1)) < TOL[dtype]


@pytest.mark.parametrize("dtype", DT)
def test_upsample_bwd_add_relu(dtype):
    n, c, h, w = 2, 32, 5, 7
    g = torch.Generator().manual_seed(9)
    code = DTYPE_CODE[dtype]
    x = torch.randn(n, c, h, w, generator=g, requires_grad=True)
    dy = q(torch.randn(n, c, 2 * h, 2 * w, generator=g), dtype)
    F.interpolate(x, scale_factor=2.0, mode="nearest").backward(dy)
    dx = torch.empty(n, h, w, c, dtype=dtype, device=DEV)
    lib.mi355_upsample2_bwd(to_nhwc(dy, dtype), c, dx, c, n, h, w, c, 0, code)
    a, b = q(torch.randn(n, c, h, w, generator=g), dtype), q(torch.randn(n, c, h, w, generator=g), dtype)
    ad, bd = to_nhwc(a, dtype), to_nhwc(b, dtype)
    s_ = torch.empty_like(ad); r_ = torch.empty_like(ad); rb = torch.empty_like(ad)
    lib.mi355_add(ad, c, bd, c, s_, c, n * h * w, c, code)
    lib.mi355_relu_fwd(ad, c, r_, c, n * h * w, c, code)
    lib.mi355_relu_bwd(bd, c, r_, c, rb, c, n * h * w, c, code)
    torch.cuda.synchronize()
    assert rel_err(from_nhwc(dx), x.grad) < TOL[dtype]
    assert rel_err(from_nhwc(s_), a + b) < TOL[dtype]
    assert torch.equal(from_nhwc(r_), F.relu(a)) and torch.equal(from_nhwc(rb), b * (a > 0))


@pytest.mark.parametrize("dtype", DT)
def test_pack_unpack_and_weight_pack(dtype):
    code = DTYPE_CODE[dtype]
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 3, 9, 11, generator=g)
    y = torch.full((2, 9, 11, 32), float("nan"), dtype=dtype, device=DEV)
    lib.mi355_pack_input_nchw(dev(x), y, 2, 3, 9, 11, 32, code)
    torch.cuda.synchronize()
    assert torch.equal(y.float().cpu()[..., :3].permute(0, 3, 1, 2), q(x, dtype)) and float(y[..., 3:].abs().max()) == 0
    x2 = torch.randn(2, 16, 5, 6, generator=g)
    y2 = torch.zeros(2, 5, 6, 24, dtype=dtype, device=DEV)
    lib.mi355_pack_nchw(dev(x2), y2, 2, 16, 5, 6, 24, code)
    back = torch.empty(2, 16, 5, 6, device=DEV)
    lib.mi355_unpack_output_nchw(y2, back, 2, 16, 5, 6, 24, code)
    torch.cuda.synchronize()
    assert torch.equal(back.cpu(), q(x2, dtype))
    for transposed in (0, 1):
        w = torch.randn(40, 3, 3, 3, generator=g) if not transposed else torch.randn(3, 40, 2, 2, generator=g)
        co, ci = (40, 3)
        kk = w.shape[2]
        wf_ref, wb_ref = pack_w(w, dtype, cip=32, transposed=bool(transposed))
        wf = torch.empty_like(wf_ref); wb = torch.empty_like(wb_ref)
        lib.mi355_pack_conv_weight(dev(w), wf, wb, co, ci, 32, kk, kk, transposed, code)
        torch.cuda.synchronize()
        assert torch.equal(wf, wf_ref) and torch.equal(wb, wb_ref)
    # every pack of a plan in one launch (descriptor table), ragged tiles, 7x7 stem, missing wb
    cases = [((40, 3, 3, 3), 32, 0, True), ((3, 40, 2, 2), 32, 1, True), ((64, 3, 7, 7), 32, 0, False),
             ((96, 160, 3, 3), 160, 0, True), ((37, 64, 1, 1), 64, 0, True), ((64, 32, 2, 2), 64, 1, True)]
    rows, outs = [], []
    for i, (shape, cip, tr, has_wb) in enumerate(cases):
        w = torch.randn(*shape, generator=g)
        co, ci = (shape[1], shape[0]) if tr else (shape[0], shape[1])
        wf_ref, wb_ref = pack_w(w, dtype, cip=cip, transposed=bool(tr))
        wd = dev(w)
        wf = torch.full_like(wf_ref, float("nan")); wb = torch.full_like(wb_ref, float("nan")) if has_wb else None
        sc = (torch.rand(co, generator=g) + 0.5) if (i % 2 == 0 and not tr) else None      # folded per-output-channel scale
        if sc is not None:
            wf_ref, wb_ref = pack_w(w * sc.view(-1, 1, 1, 1), dtype, cip=cip)
        scd = dev(sc) if sc is not None else None
        rows.append([wd.data_ptr(), wf.data_ptr(), wb.data_ptr() if has_wb else 0, co, ci, cip, shape[2] * shape[3], tr,
                     scd.data_ptr() if scd is not None else 0])
        outs.append((wd, wf, wb, wf_ref, wb_ref, scd))
    table = torch.tensor(rows, dtype=torch.int64).to(DEV)
    with pytest.raises(RuntimeError):
        lib.mi355_pack_conv_weights_batched(table, len(rows), 8, code)        # a stale descriptor layout is refused on the host
    lib.mi355_pack_conv_weights_batched(table, len(rows), 9, code)
    torch.cuda.synchronize()
    for i, (wd, wf, wb, wf_ref, wb_ref, _) in enumerate(outs):
        assert torch.equal(wf, wf_ref), cases[i]
        assert wb is None or torch.equal(wb, wb_ref), cases[i]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("c", [32, 64, 256])
def test_attention_gate_tail(dtype, c):
    """p -> z = psi-conv(p) -> BN(1) -> sigmoid -> x*psi, forward and full backward vs autograd."""
    n, h, w, f = 2, 6, 5, 2 * c
    code = DTYPE_CODE[dtype]
    g = torch.Generator().manual_seed(c)
    p = q(F.relu(torch.randn(n, c, h, w, generator=g)), dtype).requires_grad_(True)
    x = q(torch.randn(n, f, h, w, generator=g), dtype).requires_grad_(True)
    wp = (torch.randn(1, c, 1, 1, generator=g) / math.sqrt(c)).requires_grad_(True)
    bp = torch.randn(1, generator=g).requires_grad_(True)
    gam = torch.tensor([1.3], requires_grad=True); bet = torch.tensor([-0.2], requires_grad=True)
    z = F.conv2d(p, wp, bp)
    psi = torch.sigmoid(F.batch_norm(z, None, None, gam, bet, True, 0.1, 1e-5))
    out = x * psi
    dout = q(torch.randn(out.shape, generator=g), dtype)
    out.backward(dout)
    m = n * h * w
    pd, xd = to_nhwc(p.detach(), dtype), to_nhwc(x.detach(), dtype)
    zd = torch.empty(m, device=DEV)
    nb, part = _partials(m, 1)
    lib.mi355_rowdot_fwd(pd, c, dev(wp.detach().flatten()), dev(bp.detach()), zd, part, m, c, 0, 1, code)
    sc, sh, mu, isd = (torch.empty(1, device=DEV) for _ in range(4))
    lib.mi355_bn_finalize(part, nb, m, 1, dev(gam.detach()), dev(bet.detach()), None, None, None, 0.1, 1e-5, sc, sh, mu, isd)
    od = torch.empty_like(xd)
    lib.mi355_gate_mul_fwd(xd, f, zd, sc, sh, od, f, m, f, code)
    # backward
    dxd = torch.empty_like(xd); dzn = torch.empty(m, device=DEV)
    nb2, part2 = _partials(m, 1)
    lib.mi355_gate_mul_bwd(to_nhwc(dout, dtype), f, xd, f, zd, sc, sh, mu, isd, dxd, f, 0, dzn, part2, m, f, code)
    sums = torch.empty(2, device=DEV); dgam = torch.zeros(1, device=DEV); dbet = torch.zeros(1, device=DEV)
    lib.mi355_bn_bwd_finalize(part2, nb2, 1, sums, dgam, dbet, 0.0)
    dz = torch.empty(m, device=DEV)
    lib.mi355_bn1_bwd_apply(dzn, zd, dev(gam.detach()), mu, isd, sums, dz, m)
    dpd = torch.empty_like(pd)
    nb3, part3 = _partials(m, c)
    lib.mi355_rowdot_bwd(dz, pd, c, dev(wp.detach().flatten()), dpd, c, part3, m, c, 0, 0, 1, 0, code)
    dw = torch.empty(c, device=DEV); db = torch.empty(c, device=DEV)
    lib.mi355_colsum_finalize(part3, nb3, 2, c, dw, 0.0)
    lib.mi355_colsum_finalize(part3[c:], nb3, 2, c, db, 0.0)
    torch.cuda.synchronize()
    tol = 1e-4 if dtype == torch.float32 else 2e-2
    assert rel_err(from_nhwc(od), out.detach()) < TOL[dtype]
    assert rel_err(from_nhwc(dxd), x.grad) < tol
    assert rel_err(dgam.cpu(), gam.grad) < tol and rel_err(dbet.cpu(), bet.grad) < tol
    assert rel_err(from_nhwc(dpd), p.grad) < tol
    assert rel_err(dw.cpu(), wp.grad.flatten()) < tol
    assert abs(float(db[0]) - float(bp.grad)) < 1e-3 * float(wp.grad.abs().max()) + 1e-5   # ~0 (BN follows)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("nhw,f", [((2, 37, 41), 64), ((1, 5, 7), 96), ((2, 37, 41), 96), ((3, 16, 16), 128)])
def test_gate_multiply_backward_accumulating_ragged_and_odd_chunk_counts(nhw, f, dtype):
    """mi355_gate_mul_fwd / _bwd on the paths the attention-gate test above does not reach: accumulate = 1 into a pre-filled
    gradient (the skip tensor has other consumers, AttentionUNet.py:101-116), row counts that are no multiple of the kernel's
    row groups (2 x 37 x 41, 35 rows: clamped tail loads), channel counts whose 16-byte chunk count is not a power of two
    (96 channels: idle lanes in every row group), also through rowdot_fwd's clamped chunk loads."""
    n, h, w = nhw
    code = DTYPE_CODE[dtype]
    g = torch.Generator().manual_seed(f + h)
    m = n * h * w
    x = q(torch.randn(n, f, h, w, generator=g), dtype)
    z = torch.randn(m, generator=g)
    dout = q(torch.randn(n, f, h, w, generator=g), dtype)
    pre = q(torch.randn(n, f, h, w, generator=g), dtype)
    gam, bet = 1.3, -0.2
    mu, var = z.mean(), z.var(unbiased=False)
    isd = 1.0 / torch.sqrt(var + 1e-5)
    sc_, sh_ = gam * isd, bet - gam * isd * mu
    psi = torch.sigmoid(z * sc_ + sh_).view(n, 1, h, w)
    xd, dyd = to_nhwc(x, dtype), to_nhwc(dout, dtype)
    zd, scd, shd, mud, isdd = dev(z), dev(sc_.view(1)), dev(sh_.view(1)), dev(mu.view(1)), dev(isd.view(1))
    od = torch.empty_like(xd)
    lib.mi355_gate_mul_fwd(xd, f, zd, scd, shd, od, f, m, f, code)
    dzn = torch.empty(m, device=DEV)
    nb, part = _partials(m, 1)
    for acc in (0, 1):
        dxd = to_nhwc(pre, dtype).clone()
        lib.mi355_gate_mul_bwd(dyd, f, xd, f, zd, scd, shd, mud, isdd, dxd, f, acc, dzn, part, m, f, code)
        torch.cuda.synchronize()
        want = dout * psi + (pre if acc else 0)
        assert rel_err(from_nhwc(dxd), want) < TOL[dtype], acc
    ref_dzn = ((dout * x).sum(1, keepdim=True) * psi * (1 - psi)).flatten()
    assert rel_err(from_nhwc(od), x * psi) < TOL[dtype]
    assert rel_err(dzn.cpu(), ref_dzn) < (1e-4 if dtype == torch.float32 else 2e-2)
    sums = torch.empty(2, device=DEV); dg = torch.zeros(1, device=DEV); db = torch.zeros(1, device=DEV)
    lib.mi355_bn_bwd_finalize(part, nb, 1, sums, dg, db, 0.0)
    torch.cuda.synchronize()
    zhat = (z - mu) * isd
    assert abs(float(db) - float(ref_dzn.sum())) <= 2e-2 * float(ref_dzn.abs().sum()) + 1e-5
    assert abs(float(dg) - float((ref_dzn * zhat).sum())) <= 2e-2 * float((ref_dzn * zhat).abs().sum()) + 1e-5
    # rowdot_fwd with the same ragged / odd-chunk geometry (its chunk loads are clamped the same way)
    wp = torch.randn(f, generator=g) / math.sqrt(f)
    zz = torch.empty(m, device=DEV)
    nb2, part2 = _partials(m, 1)
    lib.mi355_rowdot_fwd(xd, f, dev(wp), dev(torch.tensor([0.25])), zz, part2, m, f, 0, 1, code)
    torch.cuda.synchronize()
    ref = (x * wp.view(1, f, 1, 1)).sum(1).flatten() + 0.25
    assert rel_err(zz.cpu(), ref) < (1e-4 if dtype == torch.float32 else 2e-2)


def test_heads_and_losses():
    g = torch.Generator().manual_seed(8)
    n, c, hw = 3, 96, 20
    for dtype in DT:
        x = q(torch.randn(n, c, 4, 5, generator=g), dtype)
        xd = to_nhwc(x, dtype)
        for is_max in (1, 0):
            y = torch.empty(n, c, device=DEV); am = torch.empty(n, c, dtype=torch.int32, device=DEV)
            lib.mi355_global_pool_fwd(xd, c, y, am, n, hw, c, is_max, DTYPE_CODE[dtype])
            xr = x.clone().requires_grad_(True)
            ref = (F.adaptive_max_pool2d(xr, 1) if is_max else F.adaptive_avg_pool2d(xr, 1)).flatten(1)
            dy = torch.randn(n, c, generator=g); ref.backward(dy)
            dx = torch.empty_like(xd)
            lib.mi355_global_pool_bwd(dev(dy), am, dx, c, n, hw, c, is_max, DTYPE_CODE[dtype])
            torch.cuda.synchronize()
            assert rel_err(y.cpu(), ref.detach()) < 1e-6 and rel_err(from_nhwc(dx), xr.grad) < TOL[dtype]
    # linear fwd/bwd (+relu)
    b, i, o = 4, 200, 7
    x = torch.randn(b, i, generator=g, requires_grad=True); w = torch.randn(o, i, generator=g, requires_grad=True)
    bias = torch.randn(o, generator=g, requires_grad=True)
    y = F.relu(F.linear(x, w, bias)); dy = torch.randn(b, o, generator=g); y.backward(dy)
    yd = torch.empty(b, o, device=DEV)
    lib.mi355_linear_fwd(dev(x.detach()), dev(w.detach()), dev(bias.detach()), yd, b, i, o, 1)
    dx, dw, db = torch.empty(b, i, device=DEV), torch.zeros(o, i, device=DEV), torch.zeros(o, device=DEV)
    assert lib.mi355_linear_bwd_scratch(b, i, o) == 0
    lib.mi355_linear_bwd(dev(x.detach()), dev(w.detach()), yd, dev(dy), dx, dw, db, b, i, o, 1, 0.0, None)
    torch.cuda.synchronize()
    assert rel_err(yd.cpu(), y.detach()) < 1e-5 and rel_err(dx.cpu(), x.grad) < 1e-5
    assert rel_err(dw.cpu(), w.grad) < 1e-5 and rel_err(db.cpu(), bias.grad) < 1e-5
    # wide layers (torchvision VGG head): streaming kernels; ragged batch (19 = 16 + 3), O not a multiple of the tiles,
    # beta = 1 accumulation
    for b, i, o, relu in ((19, 2052, 517, 1), (2, 4096, 300, 0)):
        x = torch.randn(b, i, generator=g, requires_grad=True); w = (torch.randn(o, i, generator=g) / i ** 0.5).requires_grad_(True)
        bias = torch.randn(o, generator=g, requires_grad=True)
        lin = F.linear(x, w, bias)
        y = F.relu(lin) if relu else lin
        dy = torch.randn(b, o, generator=g); y.backward(dy)
        yd = torch.empty(b, o, device=DEV)
        lib.mi355_linear_fwd(dev(x.detach()), dev(w.detach()), dev(bias.detach()), yd, b, i, o, relu)
        need = lib.mi355_linear_bwd_scratch(b, i, o)
        assert need % (b * i) == 0 and need // (b * i) >= 4          # one partial per output-row range (enough ranges to fill the chip)
        scratch = torch.empty(need, device=DEV)
        dx = torch.full((b, i), float("nan"), device=DEV); dw = torch.ones(o, i, device=DEV); db = torch.ones(o, device=DEV)
        lib.mi355_linear_bwd(dev(x.detach()), dev(w.detach()), yd, dev(dy), dx, dw, db, b, i, o, relu, 1.0, scratch)
        torch.cuda.synchronize()
        assert rel_err(yd.cpu(), y.detach()) < 1e-5 and rel_err(dx.cpu(), x.grad) < 1e-5
        assert rel_err(dw.cpu(), w.grad + 1) < 1e-5 and rel_err(db.cpu(), bias.grad + 1) < 1e-5
        with pytest.raises(RuntimeError):
            lib.mi355_linear_bwd(dev(x.detach()), dev(w.detach()), yd, dev(dy), dx, dw, db, b, i, o, relu, 0.0, None)
    # dropout: keep-rate and scaling
    nn_ = 1 << 18
    xx = torch.ones(nn_, device=DEV); yy = torch.empty(nn_, device=DEV); mk = torch.empty(nn_, dtype=torch.uint8, device=DEV)
    lib.mi355_dropout_fwd(xx, yy, mk, nn_, 0.3, 1234, None)
    dd = torch.empty(nn_, device=DEV)
    lib.mi355_dropout_bwd(xx, mk, dd, nn_, 0.3)
    torch.cuda.synchronize()
    keep = float(mk.float().mean())
    assert abs(keep - 0.7) < 5e-3 and torch.equal(yy, dd) and abs(float(yy.max()) - 1 / 0.7) < 1e-6
    # losses
    z = torch.randn(4, 1, 16, 16, generator=g) * 3; t = (torch.rand(4, 1, 16, 16, generator=g) > 0.5).float()
    zr = z.clone().requires_grad_(True); l = F.binary_cross_entropy_with_logits(zr, t); l.backward()
    loss = torch.empty(1, device=DEV); dz = torch.empty(z.numel(), device=DEV)
    lib.mi355_bce_logits(dev(z), dev(t), loss, dz, None, z.numel())
    torch.cuda.synchronize()
    assert abs(float(loss) - float(l)) < 1e-5 and rel_err(dz.cpu().view_as(z), zr.grad) < 1e-5
    lg = torch.randn(6, 3, generator=g); yy_ = torch.randint(0, 3, (6,), generator=g)
    lr_ = lg.clone().requires_grad_(True); l2 = F.cross_entropy(lr_, yy_, label_smoothing=0.1); l2.backward()
    dl = torch.empty(6, 3, device=DEV)
    lib.mi355_ce_smooth(dev(lg), dev(yy_), loss, dl, None, 6, 3, 0.1)
    torch.cuda.synchronize()
    assert abs(float(loss) - float(l2)) < 1e-5 and rel_err(dl.cpu(), lr_.grad) < 1e-5


def test_clip_and_adamw_match_torch():
    g = torch.Generator().manual_seed(21)
    n = 100_003
    p0 = torch.randn(n, generator=g); p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([p], lr=1e-3, weight_decay=5e-4)
    pd = dev(p0.clone()); pad = (-n) % 4
    m = torch.zeros(n, device=DEV); v = torch.zeros(n, device=DEV)
    lr = torch.tensor([1e-3], device=DEV); step = torch.zeros(1, dtype=torch.int32, device=DEV)
    norm, coef, finf = (torch.empty(1, device=DEV) for _ in range(3))
    nb = lib.mi355_rowreduce_blocks(n); part = torch.empty(nb, device=DEV)
    for it in range(3):
        gr = torch.randn(n, generator=g) * (10 if it == 0 else 0.001)
        p.grad = gr.clone()
        tn = torch.nn.utils.clip_grad_norm_([p], 1.0)
        opt.step()
        gd = dev(gr)
        lib.mi355_sumsq_partial(gd, part, n)
        lib.mi355_clip_coef(part, nb, 1.0, 1.0, None, norm, coef, finf, step)
        lib.mi355_adamw(pd, gd, m, v, n, lr, 0.9, 0.999, 1e-8, 5e-4, coef, 1.0, None, finf, step)
        torch.cuda.synchronize()
        assert abs(float(norm) - float(tn)) < 1e-4 * float(tn)
        assert rel_err(pd.cpu(), p.detach()) < 1e-6
    assert int(step) == 3
    # non-finite gradient => step skipped, counter unchanged
    gd = dev(torch.full((n,), float("inf")))
    before = pd.clone()
    lib.mi355_sumsq_partial(gd, part, n)
    lib.mi355_clip_coef(part, nb, 1.0, 1.0, None, norm, coef, finf, step)
    lib.mi355_adamw(pd, gd, m, v, n, lr, 0.9, 0.999, 1e-8, 5e-4, coef, 1.0, None, finf, step)
    torch.cuda.synchronize()
    assert torch.equal(pd, before) and int(step) == 3 and float(finf) == 1.0
    # device-side loss scale: gradients carry a factor 1024, dev_scale = 1/1024 restores the step and the norm
    gr = torch.randn(n, generator=g)
    p.grad = gr.clone()
    tn = torch.nn.utils.clip_grad_norm_([p], 1.0)
    opt.step()
    gd = dev(gr * 1024.0); inv = torch.tensor([1.0 / 1024.0], device=DEV)
    lib.mi355_sumsq_partial(gd, part, n)
    lib.mi355_clip_coef(part, nb, 1.0, 1.0, inv, norm, coef, finf, None)
    lib.mi355_step_tick(step, finf)
    lib.mi355_adamw(pd, gd, m, v, n, lr, 0.9, 0.999, 1e-8, 5e-4, coef, 1.0, inv, finf, step)
    torch.cuda.synchronize()
    assert int(step) == 4 and abs(float(norm) - float(tn)) < 1e-4 * float(tn) and rel_err(pd.cpu(), p.detach()) < 1e-6
    # scale update rule (torch.amp.GradScaler.update)
    scale = torch.tensor([65536.0], device=DEV); iv = torch.tensor([1 / 65536.0], device=DEV)
    tr = torch.zeros(1, dtype=torch.int32, device=DEV); one = torch.ones(1, device=DEV); zero = torch.zeros(1, device=DEV)
    lib.mi355_amp_update(scale, iv, tr, one, 2.0, 0.5, 3)
    assert float(scale) == 32768.0 and int(tr) == 0 and float(iv) == 1 / 32768.0
    for k in range(3):
        lib.mi355_amp_update(scale, iv, tr, zero, 2.0, 0.5, 3)
        assert float(scale) == (65536.0 if k == 2 else 32768.0) and int(tr) == (0 if k == 2 else k + 1)


def test_seg_counts():
    g = torch.Generator().manual_seed(2)
    logit = torch.randn(3, 1, 33, 17, generator=g); t = (torch.rand(3, 1, 33, 17, generator=g) > 0.6).float()
    cnt = torch.empty(3, 4, device=DEV)
    lib.mi355_seg_counts(dev(logit), dev(t), cnt, 3, 33 * 17, 1, 0.5)
    torch.cuda.synchronize()
    pb = torch.sigmoid(logit) > 0.5; tb = t > 0.5
    exp = torch.stack([(pb & tb).flatten(1).sum(1), pb.flatten(1).sum(1), tb.flatten(1).sum(1),
                       (pb == tb).flatten(1).sum(1)], 1).float()
    assert torch.equal(cnt.cpu(), exp)


def test_adamw_checkpoint_interoperates_with_torch():
    import copy
    """mi355.optim.AdamW.state_dict() has torch.optim.AdamW's layout: a torch optimiser continues from our checkpoint
    and we continue from torch's, bit-compatibly with an uninterrupted run (resume, SURVEY.md 8f N3)."""
    from mi355 import optim as moptim
    from models.classification_models.ResNet import ResNet18
    torch.manual_seed(0)
    m = ResNet18(num_classes=3)
    m.compute_dtype = torch.float32
    m = m.to(DEV).train()
    m.engine._check_storage()
    ps = list(m.parameters())
    g = torch.Generator().manual_seed(3)
    grads = [[torch.randn(p.shape, generator=g) * 0.01 for p in ps] for _ in range(4)]

    def run(opt, k0, k1, params):
        for k in range(k0, k1):
            for p, gr in zip(params, grads[k]):
                p.grad = gr.to(p.device)
            if isinstance(opt, moptim.AdamW):
                m.engine.flat_g.zero_()
                for p, gr in zip(params, grads[k]):
                    m.engine.grad_view(p).copy_(gr.to(DEV))
            opt.step()

    p0 = [p.detach().clone() for p in ps]
    ours = moptim.AdamW(ps, lr=1e-3, weight_decay=5e-4)
    run(ours, 0, 2, ps)
    sd = ours.state_dict()
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"} and float(sd["state"][0]["step"]) == 2.0
    # torch continues from our checkpoint
    tp = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    topt = torch.optim.AdamW(tp, lr=1e-3, weight_decay=5e-4)
    topt.load_state_dict(copy.deepcopy(sd))        # (torch keeps the `step` tensors by reference and bumps them in place)
    run(topt, 2, 4, tp)
    # we continue from our own checkpoint in a fresh optimiser
    fresh = moptim.AdamW(ps, lr=1e-3, weight_decay=5e-4)
    fresh.load_state_dict(sd)
    run(fresh, 2, 4, ps)
    torch.cuda.synchronize()
    for a, b in zip(ps, tp):
        assert rel_err(a.detach().cpu(), b.detach().cpu()) < 1e-5
    # and from torch's checkpoint: restart both from the initial weights
    with torch.no_grad():
        for p, q0 in zip(ps, p0):
            p.copy_(q0)
    tp2 = [torch.nn.Parameter(q0.clone()) for q0 in p0]
    t2 = torch.optim.AdamW(tp2, lr=1e-3, weight_decay=5e-4)
    run(t2, 0, 2, tp2)
    with torch.no_grad():
        for p, q in zip(ps, tp2):
            p.copy_(q)
    back = moptim.AdamW(ps, lr=1e-3, weight_decay=5e-4)
    back.load_state_dict(copy.deepcopy(t2.state_dict()))
    run(back, 2, 4, ps)
    run(t2, 2, 4, tp2)
    torch.cuda.synchronize()
    for a, b in zip(ps, tp2):
        assert rel_err(a.detach().cpu(), b.detach().cpu()) < 1e-5
