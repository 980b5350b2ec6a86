"""GPU diagnostic (not a test): where does the bf16 train-mode forward of AttentionUNet drift from the fp32 HIP forward?
Relative L2 difference of every post-ReLU activation (Plan.acts), in forward order.  usage: diag_bf16_depth.py [batch] [size]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "medical-image-segmentation-and-classification_amd"), os.path.join(ROOT, "tests")]
import torch
import bench
import bench_scale_worker as w
from models.segmentation_models.AttentionUNet import AttentionUNet
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
hw = int(sys.argv[2]) if len(sys.argv) > 2 else 256
x, _ = bench.make_batch(bs, hw, seed=0, device="cuda:0")
sd = w.he_state()
acts, outs = {}, {}
for dt in (torch.float32, torch.bfloat16, torch.float16):
    m = AttentionUNet(); m.load_state_dict(sd); m.compute_dtype = dt; m = m.to("cuda:0").train()
    with torch.no_grad():
        outs[dt] = m(x).float().clone()
    plan = [p for p in m.engine.plans.values() if p.training][0]
    acts[dt] = [(a[0], a[1].torch_view().float().clone()) for a in plan.acts if a[0] == "relu"]
    names = [l.name for l in plan.fwd]
l2 = lambda a, b: float((a - b).norm() / b.norm())
for i, ((k, a32), (_, a16), (_, ah)) in enumerate(zip(acts[torch.float32], acts[torch.bfloat16], acts[torch.float16])):
    print(f"relu #{i:2d} shape {tuple(a32.shape)}  bf16 l2rel {l2(a16, a32):.3e}   fp16 l2rel {l2(ah, a32):.3e}   mean|a| {float(a32.abs().mean()):.3f}")
print("logits: bf16", l2(outs[torch.bfloat16], outs[torch.float32]), "fp16", l2(outs[torch.float16], outs[torch.float32]),
      " logit std", float(outs[torch.float32].std()), "mean", float(outs[torch.float32].mean()))
