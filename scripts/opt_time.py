import sys, os
sys.path[:0] = ["/root/repo", "/root/repo/medical-image-segmentation-and-classification_amd"]
import torch, bench
from mi355 import nn as mnn, optim as moptim
from utils.helpers import get_seg_model
m = get_seg_model("attentionunet"); m.compute_dtype = torch.bfloat16; m = m.cuda().train()
x, y = bench.make_batch(32, 256, 0, "cuda")
crit = mnn.BCEWithLogitsLoss(); opt = moptim.AdamW(m.parameters(), lr=1e-6, weight_decay=5e-4)
for _ in range(3):
    opt.zero_grad(set_to_none=True); crit(m(x), y).backward(); moptim.clip_grad_norm_(m.parameters(), 1.0); opt.step()
torch.cuda.synchronize()
def t(fn, n=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
print("zero_grad      %.3f ms" % t(lambda: opt.zero_grad(set_to_none=True)))
print("clip_grad_norm %.3f ms" % t(lambda: moptim.clip_grad_norm_(m.parameters(), 1.0)))
print("adamw step     %.3f ms" % t(lambda: opt.step()))
print("numel", sum(p.numel() for p in m.parameters()))
