"""GPU diagnostic (not a test): per-tensor gradient error of the fp32 HIP path vs the fp64 oracle, next to the CPU fp32 oracle's
own error, in parameter order.  usage: python tests/diag/diag_grads.py AttentionUNet 2 32 [out_channels]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "medical-image-segmentation-and-classification_amd")]
import torch
from oracle import nets, train as otrain
from mi355 import nn as mnn

name, bs, hw = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
K = int(sys.argv[4]) if len(sys.argv) > 4 else 1
if name == "AttentionUNet":
    from models.segmentation_models.AttentionUNet import AttentionUNet
    m = AttentionUNet(out_channel=K); okw = {}
else:
    from models.segmentation_models.R2U_Net import R2U_Net
    m = R2U_Net(out_channels=K, t=2); okw = {"t": 2}
sd = nets.closed_form_state(name, out_channels=K)
m.load_state_dict(sd); m.compute_dtype = torch.float32; m = m.to("cuda:0").train()
x, mask = otrain.closed_form_input(bs, hw)
y = torch.cat([mask.roll(3 * k, dims=3) for k in range(K)], 1)
out = m(x.cuda()); loss = mnn.BCEWithLogitsLoss()(out, y.cuda()); loss.backward(); torch.cuda.synchronize()
sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
l64, o64, g64 = otrain.forward_backward(name, sd64, x.double(), y.double(), True, **okw)
l32, o32, g32 = otrain.forward_backward(name, {k: v.clone() for k, v in sd.items()}, x, y, True, **okw)
print("loss", float(loss), l64, "logits rel", float((out.detach().cpu().double() - o64).abs().max() / o64.abs().max()))
gmax = max(float(v.abs().max()) for v in g64.values())
for k, p in m.named_parameters():
    ref = g64[k]; sc = float(ref.abs().max())
    if sc < 1e-6 * gmax: continue
    g = p.grad.cpu().double()
    eg = float((g - ref).abs().max()) / sc; ec = float((g32[k].double() - ref).abs().max()) / sc
    ratio = float((g * ref).sum() / (ref * ref).sum())
    print(f"{k:34s} gpu {eg:.2e} cpu {ec:.2e}  projection {ratio:.6f}")
