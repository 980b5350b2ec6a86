import sys, os, time
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'medical-image-segmentation-and-classification_amd'))
import torch
from oracle import nets, train as otrain
from mi355 import nn as mnn, optim as moptim
from models.segmentation_models.AttentionUNet import AttentionUNet

def task(b, hw, seed):
    x, m = otrain.synthetic_batch(b, hw, seed=seed)
    x = 0.6 * x + m * torch.tensor([1.0, -0.7, 0.4]).view(1, 3, 1, 1)     # ellipse visible in the image
    return x, m

def dice(logit, m):
    p = (torch.sigmoid(logit) > 0.5).float(); t = (m > 0.5).float()
    return float((2 * (p * t).sum() + 1e-7) / (p.sum() + t.sum() + 1e-7))

hw, b, steps, lr = 64, 4, int(sys.argv[1]) if len(sys.argv) > 1 else 40, 1e-3
batches = [task(b, hw, s) for s in range(4)]
xv, mv = task(8, hw, 99)
sd0 = nets.default_init_state("AttentionUNet", seed=0)
res = {}
for mode in ("bf16", "fp32"):
    m = AttentionUNet(); m.load_state_dict(sd0); m.compute_dtype = torch.bfloat16 if mode == "bf16" else torch.float32
    m = m.cuda().train(); opt = moptim.AdamW(m.parameters(), lr=lr, weight_decay=5e-4); crit = mnn.BCEWithLogitsLoss()
    for i in range(steps):
        x, y = batches[i % 4]
        opt.zero_grad(); out = m(x.cuda()); loss = crit(out, y.cuda()); loss.backward(); moptim.clip_grad_norm_(m.parameters(), 1.0); opt.step()
    with torch.no_grad():
        d_tr = dice(m(xv.cuda()).float().cpu(), mv)
        m.eval(); d_ev = dice(m(xv.cuda()).float().cpu(), mv)
    res[mode] = (float(loss.detach()), d_tr, d_ev)
    print(mode, 'final loss %.4f dice(train-mode BN) %.4f dice(eval) %.4f' % res[mode], flush=True)
t0 = time.time()
sd = {k: v.clone() for k, v in sd0.items()}
opt = otrain.AdamW(nets.param_keys(sd), lr)
for i in range(steps):
    x, y = batches[i % 4]
    loss, _, _ = otrain.train_step("AttentionUNet", sd, x, y, opt, True)
with torch.no_grad():
    d_tr = dice(nets.attention_unet({k: v.clone() for k, v in sd.items()}, xv, True), mv)
    d_ev = dice(nets.attention_unet(sd, xv, False), mv)
print('oracle fp32 final loss %.4f dice(train-mode BN) %.4f dice(eval) %.4f  (%.1fs)' % (loss, d_tr, d_ev, time.time() - t0))
