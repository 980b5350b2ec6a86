"""Where does the HIP fp32 R2AttU_Net trajectory leave the CPU oracle's?  Protocol of tests/golden/train_traj_R2AttU_Net.npz
(closed-form weights, 32 x 32, batch 4, lr 1e-4, two batches per epoch): after every optimiser step, the largest relative
deviation of parameters / running statistics between HIP and oracle, the eval-mode validation loss of each, and the CROSS
evaluations (oracle state through the HIP eval forward and vice versa) that separate a state difference from a forward one."""
import os
import sys

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "medical-image-segmentation-and-classification_amd"))
import numpy as np
import torch
from oracle import nets, train as otrain
from mi355 import nn as mnn, optim as moptim

name = sys.argv[1] if len(sys.argv) > 1 else "R2AttU_Net"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
lr, hw = 1e-4, 32
if name == "R2AttU_Net":
    from models.segmentation_models.R2AttU_Net import R2AttU_Net as C
elif name == "R2U_Net":
    from models.segmentation_models.R2U_Net import R2U_Net as C
else:
    from models.segmentation_models.AttentionUNet import AttentionUNet as C
b = [otrain.synthetic_batch(4, hw, seed=s) for s in (0, 1, 2)]
sd = nets.closed_form_state(name)
m = C(); m.load_state_dict(sd); m.compute_dtype = torch.float32; m = m.cuda()
opt = moptim.AdamW(m.parameters(), lr=lr, weight_decay=5e-4); crit = mnn.BCEWithLogitsLoss()
oopt = otrain.AdamW(nets.param_keys(sd), lr)
fwd = nets.NETS[name]


def rel(a, b_):
    a = a.double(); b_ = b_.double()
    return float((a - b_).abs().max() / (b_.abs().max() + 1e-30))


def val_hip(model):
    model.eval()
    with torch.no_grad():
        z = model(b[2][0].cuda()).float().cpu()
    model.train()
    return float(otrain.bce_with_logits(z, b[2][1])), z


def val_oracle(state):
    with torch.no_grad():
        z = fwd({k: v.clone() for k, v in state.items()}, b[2][0], False)
    return float(otrain.bce_with_logits(z, b[2][1])), z


for i in range(steps):
    x, y = b[i % 2]
    m.train()
    opt.zero_grad(); loss = crit(m(x.cuda()), y.cuda()); loss.backward(); moptim.clip_grad_norm_(m.parameters(), 1.0); opt.step()
    oloss, _, _ = otrain.train_step(name, sd, x, y, oopt, True)
    msd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    dev = {k: rel(msd[k], sd[k]) for k in sd if sd[k].is_floating_point()}
    wp = max((k for k in dev if not nets.is_buffer(k)), key=dev.get)
    wm = max((k for k in dev if k.endswith("running_mean")), key=dev.get)
    wv = max((k for k in dev if k.endswith("running_var")), key=dev.get)
    vh, zh = val_hip(m)
    vo, zo = val_oracle(sd)
    # cross: oracle state through the HIP eval forward
    m2 = C(); m2.load_state_dict(sd); m2.compute_dtype = torch.float32; m2 = m2.cuda()
    vx, zx = val_hip(m2)
    vy, zy = val_oracle(msd)
    print(f"step {i + 1}: loss HIP {float(loss):.6f} oracle {oloss:.6f} | worst param {wp} {dev[wp]:.1e}  mean {wm} {dev[wm]:.1e}  var {wv} {dev[wv]:.1e}")
    print(f"   eval val loss: HIP {vh:.4f}  oracle {vo:.4f}  oracle-state@HIP-fwd {vx:.4f} (logits rel {rel(zx, zo):.1e})  HIP-state@oracle-fwd {vy:.4f} (logits rel {rel(zy, zh):.1e})", flush=True)
    del m2
