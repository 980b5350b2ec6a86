"""Helpers shared by the -m gpu parity tests (torch is used here only to move data and to
compute the fp32 CPU reference of each op)."""
import torch

from mi355.lib import lib, F32, BF16, DTYPE_CODE  # noqa: F401

DEV = "cuda:0"


def to_nhwc(x, dtype=torch.float32, cpad=None, dev=DEV):
    """NCHW fp32 CPU -> contiguous [N,H,W,C(pad)] device tensor."""
    n, c, h, w = x.shape
    cp = cpad or c
    y = torch.zeros(n, h, w, cp, dtype=dtype)
    y[..., :c] = x.permute(0, 2, 3, 1).to(dtype)
    return y.to(dev)


def from_nhwc(y, c=None):
    y = y.float().cpu()
    if c is not None:
        y = y[..., :c]
    return y.permute(0, 3, 1, 2).contiguous()


def pack_w(w, dtype, cip=None, transposed=False, dev=DEV):
    """[Co,Ci,KH,KW] -> Wf [Co][taps][Cip], Wb [Cip][taps][Co] (torch reference of the pack kernel)."""
    if transposed:
        w = w.permute(1, 0, 2, 3)
    co, ci, kh, kw = w.shape
    cip = cip or ci
    wf = torch.zeros(co, kh * kw, cip, dtype=dtype)
    wf[..., :ci] = w.permute(0, 2, 3, 1).reshape(co, kh * kw, ci).to(dtype)
    wb = wf.permute(2, 1, 0).contiguous()
    return wf.to(dev), wb.to(dev)


def rel_err(a, b):
    a = a.double(); b = b.double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def q(x, dtype):
    """Round-trip through the storage dtype (what the kernel actually sees)."""
    return x.to(dtype).float()


def _fma32(a, b, c):
    """fmaf(a, b, c) of fp32 tensors: the product is exact in fp64, the sum is rounded once to fp64 and once to fp32 — the same
    value as the fused operation except in double-rounding ties, and always the same SIGN (what a ReLU mask needs)."""
    return (a.double() * b.double() + c.double()).float()


def gpu_kinks(plan):
    """ReLU masks (NCHW bool) and max-pool arg-max indices (torch layout) of the forward the plan just ran — the decisions the
    GPU actually took, read back from the plan's activation buffers (Plan.acts), for replay inside the oracle (oracle.nets.Kinks)."""
    import torch.nn.functional as F
    relu, pool = [], []
    for a in plan.acts:
        if a[0] == "relu":
            relu.append((a[1].torch_view().float() > 0).permute(0, 3, 1, 2).contiguous().cpu())
        elif a[0] == "relu_pre":                  # recurrent block: only x + relu(.) is stored; the mask is that of fmaf(y, scale, shift)
            y, sc, sh = a[1], a[2], a[3]
            v = _fma32(y.torch_view().float(), sc[: y.C].view(1, 1, 1, -1), sh[: y.C].view(1, 1, 1, -1))
            relu.append((v > 0).permute(0, 3, 1, 2).contiguous().cpu())
        elif a[0] == "relu_pre2":                 # attention gate: relu(bn(g1) + bn(x1)) is never stored (mi355_gate_psi_fwd)
            # the kernels' own chain (gate.hip): sh = shift_g + shift_x, f = fmaf(g1, scale_g, sh), f = fmaf(x1, scale_x, f) — an
            # un-fused a * b + c puts a few of the 10^7 near-zero elements on the other side, and a replay with THOSE masks is 2-4e-3
            # off in the gate's gradients at the 32 x 32 level (measured at the benchmark shape)
            g1, sg, tg, x1, sx, tx = a[1:]
            c = g1.C
            v = _fma32(g1.torch_view().float(), sg[:c].view(1, 1, 1, -1), (tg[:c] + tx[:c]).view(1, 1, 1, -1))
            v = _fma32(x1.torch_view().float(), sx[:c].view(1, 1, 1, -1), v)
            relu.append((v > 0).permute(0, 3, 1, 2).contiguous().cpu())
        elif a[0] == "relu_v":                    # ReLU behind a Linear of a classifier head: fp32 [B, F]
            y = a[1]
            relu.append((y.buf[: y.B * y.F].view(y.B, y.F) > 0).cpu())
        elif a[0] == "gmax":                      # AdaptiveMaxPool2d((1, 1)): the kernel's arg-max pixel per (n, c)
            x, am = a[1], a[2]
            pool.append(am[: x.N * x.C].view(x.N, x.C).long().cpu())
        elif a[0] == "pool":
            x, k, s, p = a[1], a[3], a[4], a[5]
            xv = x.torch_view().float().permute(0, 3, 1, 2).contiguous().cpu()
            pool.append(F.max_pool2d(xv, k, s, p, return_indices=True)[1])          # first maximum wins, as in the kernel
    return relu, pool


def replayed_oracle(name, sd, x, y, relu, pool, seg=True, **net_kw):
    """fp64 oracle forward + backward with the given kink decisions replayed -> (loss, logits, grads)."""
    from oracle import nets, train as otrain
    sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    nets.Kinks.start("replay", relu, pool)
    try:
        res = otrain.forward_backward(name, sd64, x.double(), y.double() if seg else y, seg, **net_kw)
    finally:
        _, _, used = nets.Kinks.stop()
    assert used == (len(relu), len(pool)), (used, len(relu), len(pool))
    return res
