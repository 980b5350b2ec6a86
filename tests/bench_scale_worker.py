"""Helper process of tests/test_gpu_bench_scale.py (not a test): Attention U-Net at the BENCHMARK shape (256x256, batch 32) on
the HIP path in fp32, fp16 and bf16 from the same weights and batch — one evaluation and one train step each — and the
differences of the 2-byte runs from the fp32 run, layer by layer.  A process of its own because the kernel selection switches
(MI355_WGRAD_HALO, MI355_IGEMM_VARIANT) are read once per process.      usage: python bench_scale_worker.py OUT.npz"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "medical-image-segmentation-and-classification_amd")]

import numpy as np
import torch

import bench
from oracle import nets


def he_state():
    sd = nets.default_init_state("AttentionUNet", seed=0)
    for v in sd.values():
        if v.dim() == 4:
            v.mul_(6 ** 0.5)          # eval-mode BN with fresh running statistics is the identity: keep activations O(1)
    return sd


def l2rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def run(dtype, x, y):
    from mi355 import nn as mnn
    from models.segmentation_models.AttentionUNet import AttentionUNet
    m = AttentionUNet()
    m.load_state_dict(he_state())
    m.compute_dtype = dtype
    m = m.to("cuda:0")
    m.eval()
    with torch.no_grad():
        ev = m(x).float()
    m.train()
    out = m(x)
    loss = mnn.BCEWithLogitsLoss()(out, y)
    # fp16 activation gradients need the reference's loss scaling (helpers.py:285,329): d loss / d logit = 1 / (32 * 65536) is
    # below fp16's normal range.  A power of two, divided out of the flat gradient buffer below.
    scale = 65536.0 if dtype == torch.float16 else 1.0
    (loss * scale).backward()
    torch.cuda.synchronize()
    m.engine.flat_g.mul_(1.0 / scale)
    plan = out._mi355_plan
    acts = [a[1].torch_view().float().clone() for a in plan.acts if a[0] == "relu"]
    return {"eval": ev, "logits": out.detach().float().clone(), "loss": float(loss.detach()), "acts": acts,
            "grad_norm": np.array([float(p.grad.double().norm()) for _, p in m.named_parameters()]),
            "flat_g": m.engine.flat_g.clone(), "tags": sorted({l.tag for l in plan.fwd + plan.bwd if l.tag})}


def main():
    x, y = bench.make_batch(32, 256, seed=0, device="cuda:0")
    ref = run(torch.float32, x, y)
    rec = {"eval_first": ref["eval"][0].cpu().numpy(), "eval_last": ref["eval"][31].cpu().numpy(), "loss_fp32": ref["loss"],
           "grad_norm_fp32": ref["grad_norm"], "finite_fp32": bool(torch.isfinite(ref["flat_g"]).all())}
    for tag, dt in (("fp16", torch.float16), ("bf16", torch.bfloat16)):
        r = run(dt, x, y)
        rec[f"tags_{tag}"] = np.array(r["tags"])
        rec[f"loss_{tag}"] = r["loss"]
        rec[f"eval_err_{tag}"] = l2rel(r["eval"], ref["eval"])
        rec[f"logit_err_{tag}"] = l2rel(r["logits"], ref["logits"])
        rec[f"act_err_{tag}"] = np.array([l2rel(a, b) for a, b in zip(r["acts"], ref["acts"])])
        rec[f"grad_norm_{tag}"] = r["grad_norm"]
        g, g0 = r["flat_g"].double(), ref["flat_g"].double()
        rec[f"grad_cos_{tag}"] = float((g * g0).sum() / (g.norm() * g0.norm()))
        rec[f"grad_total_{tag}"] = float(g.norm() / g0.norm())
        rec[f"finite_{tag}"] = bool(torch.isfinite(r["flat_g"]).all())
        rec[f"grad_sample_{tag}"] = r["flat_g"][::997].cpu().numpy()
        rec[f"logits_{tag}"] = r["logits"][:2].cpu().numpy()
        rec[f"mask_agree_{tag}"] = float(((r["logits"] > 0) == (ref["logits"] > 0)).float().mean())
    np.savez(sys.argv[1], **rec)
    for k, v in rec.items():
        if np.ndim(v) == 0 or (np.ndim(v) == 1 and len(v) <= 30 and v.dtype.kind == "f"):
            print(k, v)


if __name__ == "__main__":
    main()
