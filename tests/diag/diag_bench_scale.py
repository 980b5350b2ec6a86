"""GPU diagnostic (not a test): train-mode forward of AttentionUNet 256x256 batch 32 — HIP fp32 and bf16 against the CPU fp32
oracle on the SAME full batch (train-mode BatchNorm couples the samples, so the oracle must see all 32 images; ~1-2 min)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "medical-image-segmentation-and-classification_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import bench
import bench_scale_worker as w
from oracle import nets
from models.segmentation_models.AttentionUNet import AttentionUNet
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
x, _ = bench.make_batch(bs, 256, seed=0, device="cpu")
sd = w.he_state()
t0 = time.time()
with torch.no_grad():
    ref = nets.attention_unet({k: v.clone() for k, v in sd.items()}, x, True).numpy()
print("oracle train-mode forward", time.time() - t0, "s", flush=True)
l2 = lambda a, b: float(np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b))
for dt in (torch.float32, torch.bfloat16):
    m = AttentionUNet(); m.load_state_dict(sd); m.compute_dtype = dt; m = m.to("cuda:0").train()
    with torch.no_grad():
        out = m(x.cuda()).float().cpu().numpy()
    print(dt, "train logits l2rel vs oracle", l2(out, ref), "per-image max", max(l2(out[i], ref[i]) for i in range(bs)),
          "sign agreement", float(((out > 0) == (ref > 0)).mean()), flush=True)
    msd = m.state_dict()
