import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'medical-image-segmentation-and-classification_amd'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import numpy as np, torch
from oracle import nets, train as otrain
import test_gpu_models as tg
name = sys.argv[1] if len(sys.argv) > 1 else 'AttentionUNet'
from mi355 import nn as mnn
z = np.load(os.path.join(tg.G, f"model_{name}.npz"))
hw, seg = int(z["hw"]), bool(z["seg"])
m, sd, kw = tg._build(name, torch.float32)
x, mask = otrain.closed_form_input(2, hw)
y = mask if seg else torch.tensor([1, 2])
m.train()
crit = mnn.BCEWithLogitsLoss() if seg else mnn.CrossEntropyLoss(label_smoothing=0.1)
out = m(x.cuda()); loss = crit(out, y.cuda()); loss.backward(); torch.cuda.synchronize()
_, oo, og = otrain.forward_backward(name, {k: v.clone() for k, v in sd.items()}, x, y, seg)
print('logit err', float((out.detach().cpu()-oo).abs().max()))
for k, p in m.named_parameters():
    ref = og[k]; got = p.grad.cpu()
    err = float((got-ref).abs().max()); mx = float(ref.abs().max())
    flag = '  <<<<' if err > 2e-3*mx + 1e-7 else ''
    print(f"{k:40s} max|ref| {mx:.3e} err {err:.3e} rel {err/(mx+1e-30):.2e}{flag}")
