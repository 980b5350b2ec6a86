"""GPU diagnostic (not a test): the "60 ms first launch" of VERDICT r3 weak #9 (commit 97bc0ed warmed it away in bench.py).

bench.py's per-launch profile replays the plan on ONE stream; in round 3 the first such replay charged one launch of vgg16_bn
512 x 512 60 ms (0.19 ms ever after).  This script reproduces the replay WITHOUT the warm-up and times the first and the second
execution of every launch, on the stream the plan normally uses for it (production binding) and on the main stream (the
profile's binding), so that what differs is visible: a kernel's first execution EVER, its first execution on a given stream,
or its first execution with another stream's bound arguments.

    python tests/diag/diag_first_launch.py [model] [batch] [size] [dtype]      (default: vgg16_bn 16 512 fp16)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "medical-image-segmentation-and-classification_amd")]
import torch

import bench
from mi355 import amp as mamp, nn as mnn, optim as moptim
from utils.helpers import get_class_model, get_seg_model

model = sys.argv[1] if len(sys.argv) > 1 else "vgg16_bn"
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 16
hw = int(sys.argv[3]) if len(sys.argv) > 3 else 512
dt = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[sys.argv[4] if len(sys.argv) > 4 else "fp16"]
seg = model in ("attentionunet", "r2attunet", "r2unet", "resnetunet")
m = get_seg_model(model) if seg else get_class_model(model)[0]
m.compute_dtype = dt
m = m.cuda().train()
x, y = bench.make_batch(bs, hw, 0, "cuda")
if not seg:
    y = torch.randint(0, 3, (bs,)).cuda()
crit = mnn.BCEWithLogitsLoss() if seg else mnn.CrossEntropyLoss(label_smoothing=0.1)
opt = moptim.AdamW(m.parameters(), lr=1e-6)
scaler = mamp.GradScaler(enabled=dt == torch.float16)
for _ in range(3):                                   # production steps: every kernel has run, on its production stream
    opt.zero_grad(set_to_none=True)
    scaler.scale(crit(m(x), y)).backward()
    scaler.step(opt); scaler.update()
torch.cuda.synchronize()
plan = [p for p in m.engine.plans.values() if p.dout is not None][0]
main = torch.cuda.current_stream().cuda_stream
side_launches = [i for i, l in enumerate(plan.bwd) if l.side]
print(f"{model} {bs}x{hw}x{hw} {dt}: {len(plan.bwd)} backward launches, {len(side_launches)} of them on the side stream in production")


def replay(calls, tag):
    rows = []
    for rep in range(3):
        recs = []
        for i, (fn, args, name, l) in enumerate(calls):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); rc = fn(*args); e1.record()
            assert rc == 0, name
            recs.append((i, name, l, e0, e1))
        torch.cuda.synchronize()
        rows.append([(i, name, l, e0.elapsed_time(e1)) for i, name, l, e0, e1 in recs])
    slow = [(a[3], b[3], c[3], a[0], a[1], a[2].tag, a[2].side) for a, b, c in zip(*rows) if a[3] > 1.0 and a[3] > 5 * min(b[3], c[3])]
    print(f"[{tag}] sum of launch times per replay: " + " / ".join(f"{sum(r[3] for r in rep):.2f} ms" for rep in rows))
    for first, second, third, i, name, ktag, side in sorted(slow, reverse=True)[:8]:
        print(f"   launch {i:4d} {name:28s} {ktag:32s} side={side}: first {first:8.3f} ms, then {second:.3f} / {third:.3f}")
    if not slow:
        print("   no launch was more than 5x slower the first time")


bwd_main = plan._resolve(plan.bwd, main)             # every launch bound to the main stream: what bench.py's profile replays
replay(list(bwd_main), "backward launches, all bound to the MAIN stream, first replay ever")
replay(list(bwd_main), "the same binding again")
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    bwd_side = plan._resolve(plan.bwd, side.cuda_stream)
    replay(list(bwd_side), "all bound to a NEW stream, first replay on it")
    replay(list(bwd_side), "the same binding again")
