"""Helper process of tests/test_gpu_bench_scale.py (not a test): one model at its BENCHMARK shape on the HIP path in fp32, fp16 and
bf16 from the same weights and batch — one evaluation and one train step each — and the differences of the 2-byte runs from the
fp32 run, layer by layer.  A process of its own because the kernel selection switches (MI355_WGRAD_HALO, MI355_IGEMM_VARIANT,
MI355_WS64, MI355_HALO_PP128) are read once per process.

    usage: [MI355_SCALE_CONFIG=C3|C4|C5seg|C5cls|C2] python bench_scale_worker.py OUT.npz

    C3     AttentionUNet 256 x 256, batch 32           (BASELINE.json configs[2]; the default)
    C4     R2AttU_Net    256 x 256, batch 16           (configs[3]: the 128-channel ping-pong kernel's grid-fill fall-back, the
                                                         weight-stationary kernel and the multi-application weight gradient at 256 x 256)
    C5seg  AttentionUNet 512 x 512, batch 16           (configs[4]'s segmenter)
    C5cls  vgg16_bn      512 x 512, batch 16           (configs[4]'s classifier, CrossEntropy(label_smoothing 0.1))
    C2     ResNetUnet    256 x 256, batch 32, frozen encoder   (configs[1])"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "medical-image-segmentation-and-classification_amd")]

import numpy as np
import torch

import bench
from oracle import nets


def he_state(name="AttentionUNet"):
    sd = nets.default_init_state(name, seed=0)
    for v in sd.values():
        if v.dim() == 4:
            v.mul_(6 ** 0.5)          # eval-mode BN with fresh running statistics is the identity: keep activations O(1)
    return sd


def _attention_unet():
    from models.segmentation_models.AttentionUNet import AttentionUNet
    return AttentionUNet()


def _r2attunet():
    from models.segmentation_models.R2AttU_Net import R2AttU_Net
    return R2AttU_Net()


def _resnet_unet():
    from models.segmentation_models.ResnetUnet import ResNetUnet
    return ResNetUnet(freeze=True)


def _vgg16_bn():
    from utils.helpers import get_class_model
    m, _ = get_class_model("vgg16_bn")
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0               # dropout streams cannot be matched across precisions
    return m


# name -> (oracle net, constructor, state, image size, batch, segmentation?, images whose eval logits are kept for the CPU oracle)
CONFIGS = {
    "C3": ("AttentionUNet", _attention_unet, he_state, 256, 32, True, (0, 31)),
    "C4": ("R2AttU_Net", _r2attunet, lambda: nets.closed_form_state("R2AttU_Net"), 256, 16, True, (0, 15)),
    "C5seg": ("AttentionUNet", _attention_unet, he_state, 512, 16, True, (0, 15)),
    "C5cls": ("VGG16_BN", _vgg16_bn, lambda: nets.closed_form_state("VGG16_BN", num_classes=3, head_dropout=True), 512, 16, False, (0, 15)),
    "C2": ("ResNetUnet", _resnet_unet, lambda: nets.closed_form_state("ResNetUnet"), 256, 32, True, (0, 31)),
}
MAX_ACTS = 40          # layer-wise activations kept per run (R2AttU_Net has 108 + : every third one)


def config():
    return CONFIGS[os.environ.get("MI355_SCALE_CONFIG", "C3")]


def make_batch(cfg, device):
    _, _, _, hw, b, seg, _ = cfg
    x, y = bench.make_batch(b, hw, seed=0, device=device)
    if not seg:
        y = torch.randint(0, 3, (b,), generator=torch.Generator().manual_seed(1)).to(device)
    return x, y


def l2rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def run(cfg, dtype, x, y):
    from mi355 import nn as mnn
    _, ctor, state, _, _, seg, _ = cfg
    m = ctor()
    m.load_state_dict(state())
    m.compute_dtype = dtype
    m = m.to("cuda:0")
    m.eval()
    with torch.no_grad():
        ev = m(x).float()
    m.train()
    out = m(x)
    loss = (mnn.BCEWithLogitsLoss() if seg else mnn.CrossEntropyLoss(label_smoothing=0.1))(out, y)
    # fp16 activation gradients need the reference's loss scaling (helpers.py:285,329): d loss / d logit = 1 / (32 * 65536) is
    # below fp16's normal range.  A power of two, divided out of the flat gradient buffer below.
    scale = 65536.0 if dtype == torch.float16 else 1.0
    (loss * scale).backward()
    torch.cuda.synchronize()
    m.engine.flat_g.mul_(1.0 / scale)
    plan = out._mi355_plan
    # (relu_pre: the raw convolution output of a recurrent application; relu_pre2: the raw W_g output of an attention gate)
    kept = [a[1] for a in plan.acts if a[0] in ("relu", "relu_pre", "relu_pre2")]
    kept = kept[::max(1, -(-len(kept) // MAX_ACTS))]
    acts = [a.torch_view().float().clone() for a in kept]
    trainable = [p for _, p in m.named_parameters() if p.requires_grad]
    return {"eval": ev, "logits": out.detach().float().clone(), "loss": float(loss.detach()), "acts": acts,
            "grad_norm": np.array([float(p.grad.double().norm()) for p in trainable]),
            "flat_g": m.engine.flat_g.clone(), "tags": sorted({l.tag for l in plan.fwd + plan.bwd if l.tag})}


def main():
    cfg = config()
    keep = cfg[6]
    x, y = make_batch(cfg, "cuda:0")
    ref = run(cfg, torch.float32, x, y)
    rec = {"eval_first": ref["eval"][keep[0]].cpu().numpy(), "eval_last": ref["eval"][keep[1]].cpu().numpy(), "loss_fp32": ref["loss"],
           "grad_norm_fp32": ref["grad_norm"], "finite_fp32": bool(torch.isfinite(ref["flat_g"]).all())}
    for tag, dt in (("fp16", torch.float16), ("bf16", torch.bfloat16)):
        r = run(cfg, dt, x, y)
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
        if cfg[5]:
            rec[f"mask_agree_{tag}"] = float(((r["logits"] > 0) == (ref["logits"] > 0)).float().mean())
        del r
        torch.cuda.empty_cache()
    np.savez(sys.argv[1], **rec)
    for k, v in rec.items():
        if np.ndim(v) == 0 or (np.ndim(v) == 1 and len(v) <= 40 and v.dtype.kind == "f"):
            print(k, v)


if __name__ == "__main__":
    main()
