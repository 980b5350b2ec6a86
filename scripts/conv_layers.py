"""Per-launch timing of every conv launch in a segmentation model's train plan (HIP events); default AttentionUNet 256^2 bs 32."""
import sys, os
R = os.path.join(os.path.dirname(__file__), '..')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'medical-image-segmentation-and-classification_amd'))
import torch
import bench
from mi355 import nn as mnn, optim as moptim
from utils.helpers import get_seg_model
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
name = sys.argv[2] if len(sys.argv) > 2 else "attentionunet"          # conv_layers.py [batch [model [size]]]
hw = int(sys.argv[3]) if len(sys.argv) > 3 else 256
m = get_seg_model(name); m.compute_dtype = torch.bfloat16; m = m.cuda().train()
x, y = bench.make_batch(bs, hw, 0, "cuda")
crit = mnn.BCEWithLogitsLoss(); opt = moptim.AdamW(m.parameters(), lr=1e-6)
for _ in range(2):
    out = m(x); crit(out, y).backward()
torch.cuda.synchronize()
plan = [p for p in m.engine.plans.values() if p.dout is not None][0]
_s = torch.cuda.current_stream().cuda_stream
fwd, bwd = plan._resolve(plan.pre + plan.fwd, _s), plan._resolve(plan.bwd, _s)   # single stream: events must bracket the side-stream launches too
rows = {}
for rep in range(3):
    recs = []
    for i, (fn, args, name, l) in enumerate(list(fwd) + list(bwd)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(x.data_ptr(), *args[1:]) if i == 0 else fn(*args); e1.record()
        if l.flops: recs.append((i, l, e0, e1))
    torch.cuda.synchronize()
    for i, l, e0, e1 in recs:
        rows.setdefault(i, [l, 1e9])
        rows[i][1] = min(rows[i][1], e0.elapsed_time(e1))
print(f"{'idx':>4s} {'kernel':34s} {'N Hi Wi Ci -> Ho Wo Co k':40s} {'ms':>8s} {'TFLOP/s':>8s}")
tot = {}
for i, (l, ms) in sorted(rows.items()):
    a = l.args
    if l.name == 'mi355_conv2d_igemm':
        shp = f"{a[4]} {a[5]}x{a[6]}x{a[7]} -> {a[9]}x{a[10]}x{a[11]} k{a[13]} up{a[19]} {'dgrad' if a[16] < 0 else 'fwd'}"
    else:
        shp = f"{a[4]} {a[5]}x{a[6]}x{a[7]} -> {a[9]}x{a[10]}x{a[11]} k{a[13]} splits{a[3]} wgrad"
    print(f"{i:4d} {l.tag:34s} {shp:40s} {ms:8.3f} {l.flops/ms/1e9:8.1f}")
