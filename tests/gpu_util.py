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
