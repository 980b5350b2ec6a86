"""Generate tests/golden/*.npz by running the REFERENCE's own classes and its own
``train()`` (imported from /root/reference) on closed-form weights and inputs.

Run only in the build container (the reference never travels):

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py

The fixtures are data (inputs are regenerated from closed forms, expected outputs
are stored); no reference source text is stored.  ``seaborn`` (used only by the
reference's EDA plots, helpers.py:14,52-118) is absent from the image and is
replaced by an empty module object so that ``utils.helpers`` imports.
"""
from __future__ import annotations

import contextlib
import io
import os
import re
import sys
import tempfile
import types

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, REF)
sys.path.insert(1, REPO)

import numpy as np
import torch

from oracle import nets
from oracle import train as otrain

OUT = os.path.join(REPO, "tests", "golden")
torch.set_num_threads(8)


def _ref_classes():
    from models.segmentation_models.AttentionUNet import AttentionUNet, AttentionGate, UpConv, basic_block
    from models.segmentation_models.R2AttU_Net import R2AttU_Net, Recurrent_block, RRCNN_block
    from models.segmentation_models.R2U_Net import R2U_Net
    from models.classification_models.ResNet import ResNet18, ResNet50, BasicBlock
    from models.classification_models.VGG import VGG16, VGG19
    return locals()


def _ref_helpers():
    sys.modules.setdefault("seaborn", types.ModuleType("seaborn"))
    import utils.helpers as H
    return H


def _zero_dropout(m):
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0


def model_fixture(name, ctor, hw, seg, lr=1e-3, head_dropout=False, H=None):
    """logits (eval + train), loss, per-parameter grad norms, BN buffers after one
    train-mode forward, parameter checksums after clip(1.0)+AdamW(lr, wd 5e-4)."""
    m = ctor()
    if head_dropout:
        H.add_dropout_to_fc(m, p=0.0)
    _zero_dropout(m)
    sd = nets.closed_form_state(name, head_dropout=head_dropout) if not seg else nets.closed_form_state(name)
    m.load_state_dict(sd)
    x, mask = otrain.closed_form_input(2, hw)
    y = mask if seg else torch.tensor([1, 2])
    m.eval()
    with torch.no_grad():
        logits_eval = m(x).clone()
    m.train()
    crit = torch.nn.BCEWithLogitsLoss() if seg else torch.nn.CrossEntropyLoss(label_smoothing=0.1)
    opt = torch.optim.AdamW(m.parameters(), lr=lr, weight_decay=5e-4)
    opt.zero_grad(set_to_none=True)
    out = m(x)
    loss = crit(out, y)
    loss.backward()
    names = [k for k, _ in m.named_parameters()]
    gnorm = np.array([float(p.grad.double().norm()) for _, p in m.named_parameters()])
    gsum = np.array([float(p.grad.double().sum()) for _, p in m.named_parameters()])
    gfull_name = names[0]
    gfull = dict(m.named_parameters())[gfull_name].grad.clone().numpy()
    total = float(torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0))
    opt.step()
    after = m.state_dict()
    bufs = [k for k in after if nets.is_buffer(k)]
    rec = {
        "hw": hw, "seg": int(seg), "lr": lr, "head_dropout": int(head_dropout),
        "logits_eval": logits_eval.numpy(), "logits_train": out.detach().numpy(),
        "loss": float(loss), "total_grad_norm": total,
        "param_names": np.array(names), "grad_norm": gnorm, "grad_sum": gsum,
        "param_l2_after": np.array([float(after[k].double().norm()) for k in names]),
        "param_sum_after": np.array([float(after[k].double().sum()) for k in names]),
        "buffer_names": np.array(bufs),
        "buffer_l2_after": np.array([float(after[k].double().norm()) for k in bufs]),
    }
    # a few full tensors for sharper pins
    if bufs:
        first_bn = [k for k in bufs if k.endswith("running_mean")][0]
        last_bn = [k for k in bufs if k.endswith("running_var")][-1]
        rec["full_" + first_bn] = after[first_bn].numpy()
        rec["full_" + last_bn] = after[last_bn].numpy()
    rec["gradfull_" + gfull_name] = gfull
    np.savez_compressed(os.path.join(OUT, f"model_{name}.npz"), **rec)
    print(f"model_{name}: loss {float(loss):.6f} |g| {total:.4f}")


def block_fixtures(C):
    """Block-level outputs (+ input gradients) of the reference's building blocks."""
    torch.manual_seed(1234)
    rec = {}

    def run(tag, mod, inputs):
        g = torch.Generator().manual_seed(7)
        with torch.no_grad():
            for i, p in enumerate(mod.parameters()):
                p.copy_(torch.randn(p.shape, generator=g) * (0.3 if p.dim() > 1 else 0.1))
            for mm in mod.modules():
                if isinstance(mm, torch.nn.BatchNorm2d):
                    mm.weight.copy_(1.0 + 0.2 * torch.randn(mm.weight.shape, generator=g))
        mod.train()
        ins = [t.clone().requires_grad_(True) for t in inputs]
        out = mod(*ins)
        w = torch.randn(out.shape, generator=g)
        (out * w).sum().backward()
        rec[tag + "/out"] = out.detach().numpy()
        rec[tag + "/wout"] = w.numpy()
        for i, t in enumerate(ins):
            rec[f"{tag}/in{i}"] = inputs[i].numpy()
            rec[f"{tag}/din{i}"] = t.grad.numpy()
        for k, v in mod.state_dict().items():
            rec[f"{tag}/sd/{k}"] = v.numpy()
        for k, p in mod.named_parameters():
            rec[f"{tag}/grad/{k}"] = p.grad.numpy()

    g = torch.Generator().manual_seed(3)
    run("basic_block", C["basic_block"](8, 16), [torch.randn(2, 8, 12, 12, generator=g)])
    run("UpConv", C["UpConv"](16, 8), [torch.randn(2, 16, 6, 6, generator=g)])
    run("AttentionGate", C["AttentionGate"](16, 16, 8),
        [torch.randn(2, 16, 8, 8, generator=g), torch.randn(2, 16, 8, 8, generator=g)])
    run("Recurrent_block", C["Recurrent_block"](8, 8, t=5), [torch.randn(2, 8, 8, 8, generator=g)])
    run("RRCNN_block", C["RRCNN_block"](4, 8, t=2), [torch.randn(2, 4, 8, 8, generator=g)])
    run("BasicBlock_s2", C["BasicBlock"](8, 16, stride=2), [torch.randn(2, 8, 8, 8, generator=g)])
    np.savez_compressed(os.path.join(OUT, "blocks.npz"), **rec)
    print("blocks:", len(rec), "arrays")


def train_traj_seg(C, H):
    """Reference train() (helpers.py:231-412), segmentation branch, on a fixed synthetic
    loader (shuffle off): parsed per-epoch log lines, best score, final checksums."""
    from torch.utils.data import DataLoader, TensorDataset
    hw, epochs, lr = 32, 3, 1e-5
    xs, ys = zip(*[otrain.synthetic_batch(4, hw, seed=s) for s in (0, 1, 2)])
    tr = DataLoader(TensorDataset(torch.cat(xs[:2]), torch.cat(ys[:2])), batch_size=4, shuffle=False)
    va = DataLoader(TensorDataset(xs[2], ys[2]), batch_size=4, shuffle=False)
    m = C["AttentionUNet"]()
    m.load_state_dict(nets.closed_form_state("AttentionUNet"))
    buf = io.StringIO()
    with tempfile.TemporaryDirectory() as d, contextlib.redirect_stdout(buf):
        best = H.train(m, tr, va, torch.device("cpu"), epochs, lr, "AttentionUNet", d, seg=True)
        saved = torch.load(os.path.join(d, "AttentionUNet_best_loss.pt"))
    lines = re.findall(r"Ep(\d+): TrainLoss ([\d.]+) \| ValLoss ([\d.]+) \| IoU ([\d.]+)", buf.getvalue())
    sd = m.state_dict()
    names = list(sd.keys())
    np.savez_compressed(
        os.path.join(OUT, "train_traj_AttentionUNet.npz"),
        hw=hw, epochs=epochs, lr=lr, best=float(best),
        log=np.array([[float(v) for v in l] for l in lines]),
        names=np.array(names),
        final_l2=np.array([float(sd[k].double().norm()) for k in names]),
        final_sum=np.array([float(sd[k].double().sum()) for k in names]),
        saved_l2=np.array([float(saved[k].double().norm()) for k in names]))
    print("train_traj seg:", lines, "best", best)


def train_traj_cls(C, H):
    """Reference train(), classification branch, across the stage-1 -> stage-2 switch
    (helpers.py:258-312) with the local ResNet18 + add_dropout_to_fc head (p forced to 0)."""
    from torch.utils.data import DataLoader, TensorDataset
    hw, epochs, lr = 32, 7, 1e-4
    xs, ys = zip(*[otrain.synthetic_batch(4, hw, seed=10 + s, classes=3) for s in (0, 1, 2)])
    tr = DataLoader(TensorDataset(torch.cat(xs[:2]), torch.cat(ys[:2])), batch_size=4, shuffle=False)
    va = DataLoader(TensorDataset(xs[2], ys[2]), batch_size=4, shuffle=False)
    m = C["ResNet18"](num_classes=1000)
    head = H.add_dropout_to_fc(m, p=0.0)
    m.load_state_dict(nets.closed_form_state("ResNet18", head_dropout=True))
    buf = io.StringIO()
    with tempfile.TemporaryDirectory() as d, contextlib.redirect_stdout(buf):
        best = H.train(m, tr, va, torch.device("cpu"), epochs, lr, "ResNet18", d, seg=False,
                       cls_head_name=head)
    lines = re.findall(r"Ep(\d+): TrainLoss ([\d.]+) \(Acc ([\d.]+)%\) \| ValLoss ([\d.]+) \| ValAcc ([\d.]+)%",
                       buf.getvalue())
    sd = m.state_dict()
    names = list(sd.keys())
    np.savez_compressed(
        os.path.join(OUT, "train_traj_ResNet18.npz"),
        hw=hw, epochs=epochs, lr=lr, best=float(best), head=head,
        log=np.array([[float(v) for v in l] for l in lines]),
        names=np.array(names),
        final_l2=np.array([float(sd[k].double().norm()) for k in names]),
        final_sum=np.array([float(sd[k].double().sum()) for k in names]))
    print("train_traj cls:", lines, "best", best)


def metric_fixture(H):
    """helpers.iou / helpers.acc on fixed tensors."""
    g = torch.Generator().manual_seed(5)
    pred = torch.rand(3, 1, 16, 16, generator=g)
    mask = (torch.rand(3, 1, 16, 16, generator=g) > 0.6).float()
    logits = torch.randn(6, 3, generator=g)
    y = torch.randint(0, 3, (6,), generator=g)
    c, n = H.acc(logits, y)
    np.savez_compressed(os.path.join(OUT, "metrics.npz"), pred=pred.numpy(), mask=mask.numpy(),
                        iou=H.iou(pred, mask), logits=logits.numpy(), y=y.numpy(), acc=np.array([c, n]))
    print("metrics: iou", H.iou(pred, mask), "acc", c, n)


def main():
    os.makedirs(OUT, exist_ok=True)
    C = _ref_classes()
    H = _ref_helpers()
    block_fixtures(C)
    metric_fixture(H)
    model_fixture("AttentionUNet", C["AttentionUNet"], 64, True)
    model_fixture("R2AttU_Net", C["R2AttU_Net"], 32, True)
    model_fixture("R2U_Net", C["R2U_Net"], 32, True)
    model_fixture("ResNet18", lambda: C["ResNet18"](num_classes=1000), 64, False, head_dropout=True, H=H)
    model_fixture("ResNet50", lambda: C["ResNet50"](num_classes=1000), 64, False, head_dropout=True, H=H)
    model_fixture("VGG16", lambda: C["VGG16"](num_classes=1000), 32, False, head_dropout=True, H=H)
    train_traj_seg(C, H)
    train_traj_cls(C, H)


if __name__ == "__main__":
    main()
