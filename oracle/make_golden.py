"""Generate tests/golden/*.npz by running the REFERENCE's own classes and its own
``train()`` (imported from /root/reference) on closed-form weights and inputs.

Run only in the build container (the reference never travels):

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py

The fixtures are data (inputs are regenerated from closed forms, expected outputs
are stored); no reference source text is stored.  Third-party modules the image
lacks and the functions under test never touch are replaced by EMPTY module
objects so that the reference's modules import: ``seaborn`` (EDA plots only,
helpers.py:14,52-118) for ``utils.helpers``; ``cv2``, ``albumentations``
(+ ``.pytorch`` with a dummy ``ToTensorV2`` name), ``torchvision``
(+ ``.transforms`` / ``.models``) for ``utils.tester``, ``utils.pipeline`` and
``models.segmentation_models.ResnetUnet`` — of which only the metric functions,
the two eval loops, the report writers (tester.py:92-312, 738-876), the two
``Pipeline._predict_*`` methods (pipeline.py:324-357) and ``DecoderBlock``
(ResnetUnet.py:17-27) are run: none of them reaches a stubbed module.
``ResNetUnet`` itself (ResnetUnet.py:29-83) runs with ``torchvision.models.resnet50``
answered by a plain-torch container of torchvision's public layout
(``_TvResNet50`` below): its constructor, ``_freeze_backbone`` and forward wiring
are the reference's own code, the ENCODER is ours and stays declared unpinned, as
do the hub models' layouts and the Albumentations transforms (they need the real
libraries).
``utils.tester`` creates ``./results`` at import time (tester.py:38): the
generator changes into a scratch directory first.
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


def _empty_module(name, **attrs):
    if name not in sys.modules:
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
    return sys.modules[name]


_SCRATCH = None


def _ref_tester_pipeline():
    """(utils.tester, utils.pipeline, DecoderBlock) of the reference, imported behind empty stand-in modules for the
    third-party packages this image lacks (module docstring) from a scratch working directory."""
    global _SCRATCH
    import transformers  # noqa: F401  (models/classification_models/CLIP.py imports it; present in the image)
    _empty_module("seaborn")
    _empty_module("cv2")
    a = _empty_module("albumentations")
    a.pytorch = _empty_module("albumentations.pytorch", ToTensorV2=type("ToTensorV2", (), {}))
    tv = _empty_module("torchvision")
    tv.transforms = _empty_module("torchvision.transforms")
    tv.models = _empty_module("torchvision.models")
    if _SCRATCH is None:
        _SCRATCH = tempfile.TemporaryDirectory()
        os.chdir(_SCRATCH.name)
    import utils.tester as T
    import utils.pipeline as P
    from models.segmentation_models.ResnetUnet import DecoderBlock
    return T, P, DecoderBlock


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


BLOCK_CASES = {   # tag -> (constructor args, input shapes); channel counts are MFMA-friendly multiples of 32
    "basic_block": ((32, 64), [(2, 32, 12, 12)]),
    "UpConv": ((64, 32), [(2, 64, 6, 6)]),
    "AttentionGate": ((64, 64, 32), [(2, 64, 8, 8), (2, 64, 8, 8)]),
    "Recurrent_block": ((32, 32, 5), [(2, 32, 8, 8)]),
    "RRCNN_block": ((32, 64, 2), [(2, 32, 8, 8)]),
    "BasicBlock_s2": ((32, 64, 2), [(2, 32, 8, 8)]),
    "DecoderBlock": ((96, 32), [(2, 64, 4, 4), (2, 32, 8, 8)]),      # ResnetUnet.py:17-27: ConvT(64,64,2,2) -> cat skip -> basic_block(96,32)
}


def block_fixtures(C):
    """Block-level outputs, input gradients, parameter gradients and post-forward BN buffers of the
    reference's building blocks.  Weights and inputs are closed-form (oracle.nets.closed_form_fill /
    closed_form_tensor), so only results are stored."""
    rec = {}
    ctor = {"basic_block": C["basic_block"], "UpConv": C["UpConv"], "AttentionGate": C["AttentionGate"],
            "Recurrent_block": lambda a, b, t: C["Recurrent_block"](a, b, t=t),
            "RRCNN_block": lambda a, b, t: C["RRCNN_block"](a, b, t=t),
            "BasicBlock_s2": lambda a, b, s: C["BasicBlock"](a, b, stride=s),
            "DecoderBlock": lambda a, b: _ref_tester_pipeline()[2](a, b)}
    for ti, (tag, (args, in_shapes)) in enumerate(BLOCK_CASES.items()):
        mod = ctor[tag](*args)
        mod.load_state_dict(nets.closed_form_fill(mod.state_dict(), salt=10.0 * ti))
        mod.train()
        ins = [nets.closed_form_tensor(s, 100.0 + 10 * ti + j).requires_grad_(True) for j, s in enumerate(in_shapes)]
        out = mod(*ins)
        w = nets.closed_form_tensor(tuple(out.shape), 200.0 + ti)
        (out * w).sum().backward()
        rec[tag + "/out"] = out.detach().numpy()
        for i, t in enumerate(ins):
            rec[f"{tag}/din{i}"] = t.grad.numpy()
        for k, v in mod.state_dict().items():
            if nets.is_buffer(k):
                rec[f"{tag}/buf/{k}"] = v.numpy()
        for k, p in mod.named_parameters():
            g = p.grad.reshape(-1)
            rec[f"{tag}/gradnorm/{k}"] = float(g.double().norm())
            rec[f"{tag}/grad/{k}"] = (g if g.numel() <= 4096 else g[::7]).numpy()     # large tensors: every 7th element
    np.savez_compressed(os.path.join(OUT, "blocks.npz"), **rec)
    print("blocks:", len(rec), "arrays")


def train_traj_seg(C, H, name="AttentionUNet", lr=1e-5, hw=32):
    """Reference train() (helpers.py:231-412), segmentation branch, on a fixed synthetic
    loader (shuffle off): parsed per-epoch log lines, best score, final checksums.
    The recurrent nets (R2AttU_Net.py:88-158, R2U_Net.py:50-111) run the same protocol at a 10x larger learning rate
    — six optimiser steps that visibly move the shared-weight convolutions and accumulate 36 running-statistics updates per
    recurrent BatchNorm — and additionally store a few FULL tensors of the final state."""
    from torch.utils.data import DataLoader, TensorDataset
    epochs = 3
    xs, ys = zip(*[otrain.synthetic_batch(4, hw, seed=s) for s in (0, 1, 2)])
    tr = DataLoader(TensorDataset(torch.cat(xs[:2]), torch.cat(ys[:2])), batch_size=4, shuffle=False)
    va = DataLoader(TensorDataset(xs[2], ys[2]), batch_size=4, shuffle=False)
    m = C[name]()
    m.load_state_dict(nets.closed_form_state(name))
    buf = io.StringIO()
    with tempfile.TemporaryDirectory() as d, contextlib.redirect_stdout(buf):
        best = H.train(m, tr, va, torch.device("cpu"), epochs, lr, name, d, seg=True)
        saved = torch.load(os.path.join(d, f"{name}_best_loss.pt"))
    lines = re.findall(r"Ep(\d+): TrainLoss ([\d.]+) \| ValLoss ([\d.]+) \| IoU ([\d.]+)", buf.getvalue())
    sd = m.state_dict()
    names = list(sd.keys())
    rec = dict(
        hw=hw, epochs=epochs, lr=lr, best=float(best),
        log=np.array([[float(v) for v in l] for l in lines]),
        names=np.array(names),
        final_l2=np.array([float(sd[k].double().norm()) for k in names]),
        final_sum=np.array([float(sd[k].double().sum()) for k in names]),
        saved_l2=np.array([float(saved[k].double().norm()) for k in names]))
    if name != "AttentionUNet":
        initial = nets.closed_form_state(name)
        rec["moved_l2"] = np.array([float((sd[k].double() - initial[k].double()).norm()) for k in names])
        for k in FULL_TENSORS[name]:
            rec["full_" + k] = sd[k].numpy()
        with torch.no_grad():                       # the final state's eval-mode logits on the validation batch
            m.eval()
            rec["final_val_logits"] = m(xs[2]).numpy()
    np.savez_compressed(os.path.join(OUT, f"train_traj_{name}.npz"), **rec)
    print(f"train_traj seg {name}:", lines, "best", best)


FULL_TENSORS = {n: ("RRCNN1.RCNN.0.conv.0.weight", "RRCNN1.RCNN.0.conv.1.running_mean", "RRCNN1.RCNN.0.conv.1.running_var",
                    "RRCNN3.RCNN.1.conv.1.running_var", "RRCNN5.RCNN.1.conv.1.running_mean", "up_RRCNN2.RCNN.1.conv.0.bias",
                    "up_RRCNN2.RCNN.1.conv.1.weight", "up_RRCNN2.RCNN.1.conv.1.running_var", "conv_1x1.weight")
                for n in ("R2AttU_Net", "R2U_Net")}


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


SEG_KEYS = ("iou", "dice", "pixel_accuracy", "precision", "recall", "f1")
CLS_SCALARS = ("accuracy", "precision", "recall", "f1")
CLS_ARRAYS = ("precision_per_class", "recall_per_class", "f1_per_class", "confusion_matrix")


def _centred_resnet18(C, H, x):
    """The reference's local ResNet18 + `add_dropout_to_fc` head at closed-form weights; the head bias is shifted by the
    batch mean of the logits so that all three classes occur (closed-form weights put every image in class 0 otherwise).
    Returns (model, shift): the shift is stored in the fixture as data."""
    m = C["ResNet18"](num_classes=1000)
    H.add_dropout_to_fc(m, p=0.0)
    m.load_state_dict(nets.closed_form_state("ResNet18", head_dropout=True))
    m.eval()
    with torch.no_grad():
        shift = m(x).mean(0)
        m.fc[1].bias -= shift
    return m, shift


def tester_fixture(C, H):
    """utils/tester.py run as it is (module docstring: empty stand-ins for cv2 / albumentations / torchvision):
    calculate_iou / _dice / _pixel_accuracy / _segmentation_metrics (:92-193) on random and degenerate masks,
    test_segmentation_model / test_classification_model (:197-312; return dicts AND printed text) over TensorDataset
    loaders with the reference's own AttentionUNet / ResNet18 at closed-form weights, print_summary (:738-805) and
    save_results_to_csv (:808-876) on the dictionaries those loops returned plus two hand-made rows."""
    from torch.utils.data import DataLoader, TensorDataset
    T, _, _ = _ref_tester_pipeline()
    g = torch.Generator().manual_seed(4)
    rnd = lambda *s: torch.rand(*s, generator=g)
    one, zero = torch.ones(1, 12, 12), torch.zeros(1, 12, 12)
    cases = {
        "random": (rnd(1, 40, 40), (rnd(1, 40, 40) > 0.6).float()),
        "random_soft_target": (rnd(1, 24, 24), rnd(1, 24, 24)),             # targets are thresholded too (:105)
        "empty_prediction": (zero + 0.2, (rnd(1, 12, 12) > 0.5).float()),
        "empty_target": (rnd(1, 12, 12), zero),
        "both_empty": (zero + 0.1, zero),
        "both_full": (one * 0.9, one),
        "at_threshold": (zero + 0.5, one),                                  # 0.5 is NOT above the threshold
        "perfect": ((rnd(1, 12, 12) > 0.5).float(),) * 2,
        "batch_of_images": (rnd(3, 1, 16, 16), (rnd(3, 1, 16, 16) > 0.5).float()),   # the functions sum over whatever they get
    }
    rec = {"seg/names": np.array(list(cases)), "seg_keys": np.array(SEG_KEYS)}
    for tag, (p, t) in cases.items():
        t = t.clone()
        rec[f"seg/{tag}/pred"], rec[f"seg/{tag}/target"] = p.numpy(), t.numpy()
        rec[f"seg/{tag}/iou"] = T.calculate_iou(p, t)
        rec[f"seg/{tag}/dice"] = T.calculate_dice(p, t)
        rec[f"seg/{tag}/pixel_accuracy"] = T.calculate_pixel_accuracy(p, t)
        m = T.calculate_segmentation_metrics(p, t)
        assert tuple(m) == SEG_KEYS
        rec[f"seg/{tag}/metrics"] = np.array([m[k] for k in SEG_KEYS])
    m7 = T.calculate_segmentation_metrics(cases["random"][0], cases["random"][1], threshold=0.7)
    rec["seg/random/metrics_t0.7"] = np.array([m7[k] for k in SEG_KEYS])

    # --- eval loops -------------------------------------------------------------------------------------------------
    seg = C["AttentionUNet"]()
    seg.load_state_dict(nets.closed_form_state("AttentionUNet"))
    xs, ms = zip(*[otrain.synthetic_batch(3, 32, seed=s) for s in (5, 6)])
    dl = DataLoader(TensorDataset(torch.cat(xs), torch.cat(ms)), batch_size=3)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(io.StringIO()):
        seg_res = T.test_segmentation_model(seg, dl, torch.device("cpu"), "AttentionUNet")
    rec["segloop/metrics"] = np.array([seg_res[k] for k in SEG_KEYS])
    rec["segloop/stdout"] = np.array(buf.getvalue())

    x, y = otrain.synthetic_batch(12, 64, seed=20, classes=3)
    cls, shift = _centred_resnet18(C, H, x)
    dl = DataLoader(TensorDataset(x, y), batch_size=5)                      # ragged last batch
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(io.StringIO()):
        cls_res = T.test_classification_model(cls, dl, torch.device("cpu"), "ResNet18")
    rec["clsloop/bias_shift"] = shift.numpy()
    for k in CLS_SCALARS:
        rec[f"clsloop/{k}"] = cls_res[k]
    for k in CLS_ARRAYS:
        rec[f"clsloop/{k}"] = np.asarray(cls_res[k])
    rec["clsloop/stdout"] = np.array(buf.getvalue())
    with torch.no_grad():
        z = cls(x)
    top = z.sort(1, descending=True).values
    rec["clsloop/logit_margin"] = (top[:, 0] - top[:, 1]).numpy()
    rec["clsloop/pred"] = z.argmax(1).numpy()

    # --- reports ----------------------------------------------------------------------------------------------------
    results = {"ResNet18": cls_res,
               "VGG16": {"accuracy": 93.0, "precision": 92.0, "recall": 93.0, "f1": 92.5},
               "AttentionUNet": seg_res,
               "R2Unet": {"iou": 79.0, "dice": 87.25, "pixel_accuracy": 96.5, "precision": 88.0, "recall": 86.5, "f1": 87.24}}
    def said(fn, *a):
        b = io.StringIO()
        with contextlib.redirect_stdout(b):
            fn(*a)
        return np.array(b.getvalue())
    rec["report/summary"] = said(T.print_summary, results)
    rec["report/summary_empty"] = said(T.print_summary, {})
    rec["report/summary_seg_only"] = said(T.print_summary, {"AttentionUNet": seg_res})
    with tempfile.TemporaryDirectory() as d:
        c, s_ = os.path.join(d, "c.csv"), os.path.join(d, "s.csv")
        rec["report/csv_stdout"] = np.array(str(said(T.save_results_to_csv, results, c, s_)).replace(d, "<dir>"))
        rec["report/csv_cls"], rec["report/csv_seg"] = np.array(open(c).read()), np.array(open(s_).read())
        rec["report/csv_stdout_seg_only"] = np.array(
            str(said(T.save_results_to_csv, {"AttentionUNet": seg_res}, c, s_)).replace(d, "<dir>"))
        rec["report/csv_stdout_empty"] = said(T.save_results_to_csv, {}, c, s_)
    np.savez_compressed(os.path.join(OUT, "tester.npz"), **rec)
    print("tester:", len(cases), "metric cases; seg loop", seg_res, "; cls loop acc", cls_res["accuracy"], "pred", rec["clsloop/pred"])


def pipeline_fixture(C, H):
    """utils/pipeline.py:324-357 `Pipeline._predict_classification` / `_predict_segmentation` run as they are, on a
    Pipeline object whose two models are set by hand (the reference's own ResNet18 / AttentionUNet classes at closed-form
    weights; `_load_models` would fetch from the hub): per image the class string, the confidence in percent and the
    uint8 mask.  Both calls are recorded for EVERY image; `process_image` (:359-418) only segments the "COVID" ones."""
    _, P, _ = _ref_tester_pipeline()
    x, _ = otrain.synthetic_batch(12, 64, seed=21, classes=3)
    cls, shift = _centred_resnet18(C, H, x)
    seg = C["AttentionUNet"]()
    seg.load_state_dict(nets.closed_form_state("AttentionUNet"))
    seg.eval()
    pipe = P.Pipeline.__new__(P.Pipeline)
    pipe.classification_model, pipe.segmentation_model = cls, seg
    preds, confs, masks = [], [], []
    for i in range(x.shape[0]):
        p, c = pipe._predict_classification(x[i:i + 1])
        preds.append(P.CLASSES.index(p)); confs.append(c)
        mk = pipe._predict_segmentation(x[i:i + 1])
        assert mk.dtype == np.uint8 and mk.shape == (64, 64) and set(np.unique(mk)) <= {0, 255}
        masks.append(mk)
    with torch.no_grad():
        z = cls(x)
        zs = seg(x)
    top = z.sort(1, descending=True).values
    pipe.segmentation_model = None
    assert pipe._predict_segmentation(x[:1]) is None
    pipe.segmentation_model = P.PlaceholderModel()
    assert pipe._predict_segmentation(x[:1]) is None
    pipe.classification_model = None
    none_cls = pipe._predict_classification(x[:1])
    np.savez_compressed(
        os.path.join(OUT, "pipeline.npz"), classes=np.array(P.CLASSES), bias_shift=shift.numpy(), pred=np.array(preds),
        confidence=np.array(confs), masks=np.packbits(np.stack(masks) == 255, axis=-1), hw=64,
        logit_margin=(top[:, 0] - top[:, 1]).numpy(), seg_logit_near_zero=(zs.abs() < 1e-3).sum(dim=(1, 2, 3)).numpy(),
        no_cls_model=np.array([str(none_cls[0]), str(none_cls[1])]))
    print("pipeline: pred", preds, "conf", [round(c, 2) for c in confs], "mask px", [int((m > 0).sum()) for m in masks])


class _TvBottleneck(torch.nn.Module):
    """Stand-in for torchvision's ResNet ``Bottleneck`` (v1.5: the stride sits on the 3x3) — a plain torch.nn container with
    torchvision's PUBLIC attribute names (conv1 / bn1 / conv2 / bn2 / conv3 / bn3 / relu / downsample), written from the
    published architecture because torchvision is absent here.  Test infrastructure of resnet_unet_fixture() only: it lets the
    REFERENCE's ResNetUnet.__init__ / forward / _freeze_backbone (ResnetUnet.py:29-83) run as they are.  The encoder itself
    stays declared unpinned (this class is ours); what the fixture pins is the reference's wiring around it."""

    def __init__(self, inplanes, planes, stride, downsample):
        super().__init__()
        nn = torch.nn
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        idn = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        return self.relu(self.bn3(self.conv3(y)) + idn)


class _TvResNet50(torch.nn.Module):
    """... and of ``torchvision.models.resnet50``'s trunk: conv1 / bn1 / relu / maxpool / layer1..4 (the attributes
    ResnetUnet.py:34-43 takes; avgpool / fc are never touched there)."""

    def __init__(self):
        super().__init__()
        nn = torch.nn
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        inplanes = 64
        for i, (planes, blocks, stride) in enumerate(((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)), start=1):
            layers = []
            for b in range(blocks):
                s = stride if b == 0 else 1
                ds = None
                if b == 0 and (s != 1 or inplanes != planes * 4):
                    ds = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride=s, bias=False), nn.BatchNorm2d(planes * 4))
                layers.append(_TvBottleneck(inplanes, planes, s, ds))
                inplanes = planes * 4
            setattr(self, f"layer{i}", nn.Sequential(*layers))


def resnet_unet_fixture():
    """The REFERENCE's ``ResNetUnet`` (ResnetUnet.py:29-83) run as it is — constructor, ``_freeze_backbone``, forward wiring (skip
    order, DecoderBlock inputs, decoder1, the 1x1 head) — with ``torchvision.models.resnet50(weights=ResNet50_Weights.DEFAULT)``
    answered by the plain-torch container above (no download, closed-form weights loaded afterwards).  Stored, for freeze=True and
    freeze=False: eval and train logits, the loss, the ``requires_grad`` pattern, which parameters received a gradient, every
    BatchNorm buffer after ONE train-mode forward (a frozen encoder still updates its running statistics: model.train() is
    global, helpers.py:315), per-parameter gradient norms, and the parameters after clip(1.0) + AdamW over ALL parameters
    (helpers.py:251,333: frozen ones have no .grad and are skipped by torch)."""
    _ref_tester_pipeline()                                # installs the empty torchvision stand-ins and imports ResnetUnet
    tvm = sys.modules["torchvision.models"]
    calls = []
    tvm.ResNet50_Weights = type("ResNet50_Weights", (), {"DEFAULT": "IMAGENET1K_V2"})

    def resnet50(weights=None, **kw):
        calls.append(weights)
        return _TvResNet50()
    tvm.resnet50 = resnet50
    from models.segmentation_models.ResnetUnet import ResNetUnet
    rec = {}
    x, mask = otrain.closed_form_input(2, 64)
    for freeze in (True, False):
        tag = "frozen" if freeze else "unfrozen"
        m = ResNetUnet(n_classes=1, freeze=freeze)
        assert calls[-1] == "IMAGENET1K_V2"               # the reference asked for the default pretrained weights
        sd = nets.closed_form_state("ResNetUnet")
        missing = m.load_state_dict(sd, strict=True)
        names = [k for k, _ in m.named_parameters()]
        m.eval()
        with torch.no_grad():
            rec[f"{tag}/logits_eval"] = m(x).numpy()
        m.train()
        opt = torch.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=5e-4)
        opt.zero_grad(set_to_none=True)
        out = m(x)
        loss = torch.nn.BCEWithLogitsLoss()(out, mask)
        loss.backward()
        params = dict(m.named_parameters())
        rec[f"{tag}/logits_train"] = out.detach().numpy()
        rec[f"{tag}/loss"] = float(loss)
        rec[f"{tag}/param_names"] = np.array(names)
        rec[f"{tag}/requires_grad"] = np.array([params[k].requires_grad for k in names])
        rec[f"{tag}/has_grad"] = np.array([params[k].grad is not None for k in names])
        rec[f"{tag}/grad_norm"] = np.array([float(params[k].grad.double().norm()) if params[k].grad is not None else -1.0 for k in names])
        rec[f"{tag}/gradfull/out.weight"] = params["out.weight"].grad.clone().numpy()                  # (before the clip scales them)
        rec[f"{tag}/gradfull/decoder2.up_sample.weight"] = params["decoder2.up_sample.weight"].grad[::4, ::4].clone().numpy()
        rec[f"{tag}/total_grad_norm"] = float(torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0))
        opt.step()
        after = m.state_dict()
        bufs = [k for k in after if nets.is_buffer(k)]
        rec[f"{tag}/buffer_names"] = np.array(bufs)
        rec[f"{tag}/buffer_l2"] = np.array([float(after[k].double().norm()) for k in bufs])
        rec[f"{tag}/param_l2_after"] = np.array([float(after[k].double().norm()) for k in names])
        rec[f"{tag}/param_moved"] = np.array([float((after[k].double() - sd[k].double()).abs().max()) for k in names])
        for k in ("encoder1.1.running_mean", "encoder5.2.bn3.running_var", "decoder5.basic_block.1.running_mean", "decoder1.2.running_var"):
            rec[f"{tag}/full/{k}"] = after[k].numpy()
        print(f"ResNetUnet {tag}: loss {float(loss):.6f} |g| {rec[f'{tag}/total_grad_norm']:.4f} frozen {int((~rec[f'{tag}/requires_grad']).sum())} / {len(names)}")
    rec["state_keys"] = np.array(list(m.state_dict().keys()))
    np.savez_compressed(os.path.join(OUT, "model_ResNetUnet.npz"), **rec)


def cls_metric_fixture():
    """utils/tester.py:49-88 ``calculate_classification_metrics`` is six scikit-learn calls (accuracy_score; precision_ /
    recall_ / f1_score with average="weighted" and average=None, zero_division=0; confusion_matrix).  tester.py itself
    cannot be imported here (cv2 / albumentations are absent), so the SAME calls with the SAME keyword arguments are
    made on fixed label vectors with the scikit-learn of this container (1.7.2; the algorithm is sklearn's published
    support-weighted average of per-class precision / recall / F1 with 0 for empty denominators) and stored."""
    from sklearn.metrics import accuracy_score, confusion_matrix, f1_score, precision_score, recall_score
    g = np.random.RandomState(7)
    cases = {
        "mixed": (g.randint(0, 3, 200), g.randint(0, 3, 200)),
        "skewed": (np.where(g.rand(300) < 0.8, 1, g.randint(0, 3, 300)), np.where(g.rand(300) < 0.7, 1, g.randint(0, 3, 300))),
        "class_never_predicted": (np.array([0, 0, 1, 1, 0, 1, 0]), np.array([0, 1, 2, 1, 2, 1, 0])),
        "class_never_true": (np.array([0, 2, 1, 1, 2, 1, 0]), np.array([0, 1, 1, 1, 0, 1, 0])),
        "perfect": (np.array([2, 0, 1, 1, 2]), np.array([2, 0, 1, 1, 2])),
        "single_class": (np.array([1, 1, 1]), np.array([1, 1, 1])),
        "all_wrong": (np.array([1, 2, 0, 1]), np.array([0, 0, 1, 2])),
    }
    rec = {"names": np.array(list(cases))}
    for tag, (pred, lab) in cases.items():
        kw = dict(zero_division=0)
        rec[f"{tag}/pred"], rec[f"{tag}/label"] = pred.astype(np.int64), lab.astype(np.int64)
        rec[f"{tag}/accuracy"] = accuracy_score(lab, pred) * 100
        rec[f"{tag}/precision"] = precision_score(lab, pred, average="weighted", **kw) * 100
        rec[f"{tag}/recall"] = recall_score(lab, pred, average="weighted", **kw) * 100
        rec[f"{tag}/f1"] = f1_score(lab, pred, average="weighted", **kw) * 100
        rec[f"{tag}/precision_per_class"] = precision_score(lab, pred, average=None, **kw) * 100
        rec[f"{tag}/recall_per_class"] = recall_score(lab, pred, average=None, **kw) * 100
        rec[f"{tag}/f1_per_class"] = f1_score(lab, pred, average=None, **kw) * 100
        rec[f"{tag}/confusion_matrix"] = confusion_matrix(lab, pred)
    np.savez_compressed(os.path.join(OUT, "cls_metrics.npz"), **rec)
    print("cls_metrics:", len(cases), "cases")


def png_fixture():
    """PNG byte streams written by PIL and the arrays PIL's ``Image.open(...).convert("RGB" / "L")`` returns for them — the
    reference's decode path (utils/dataset.py:55,101-102).  Pillow is a third-party dependency of the reference
    (requirements.txt: pillow==10.4.0; this container: see PIL.__version__ in the file), not reference code."""
    import io
    import PIL
    from PIL import Image
    g = np.random.RandomState(11)
    yy, xx = np.mgrid[0:299, 0:299]
    xray = (127 + 90 * np.sin(xx / 37.0) * np.cos(yy / 53.0) + g.randint(-3, 4, (299, 299))).clip(0, 255).astype(np.uint8)
    mask = ((((yy - 128) / 90.0) ** 2 + ((xx - 120) / 70.0) ** 2) <= 1).astype(np.uint8)[:256, :256] * 255
    pal = Image.fromarray((g.randint(0, 256, (45, 67)) % 23).astype(np.uint8), "P")
    pal.putpalette(g.randint(0, 256, 3 * 23).astype(np.uint8).tobytes())
    pal4 = Image.fromarray((g.randint(0, 256, (20, 21)) % 11).astype(np.uint8), "P")
    pal4.putpalette(g.randint(0, 256, 3 * 11).astype(np.uint8).tobytes())
    cases = {
        "gray_299": Image.fromarray(xray, "L"),                                   # what the dataset's images look like
        "mask_256": Image.fromarray(mask, "L"),                                   # ... and its masks
        "rgb_odd": Image.fromarray(g.randint(0, 256, (31, 45, 3)).astype(np.uint8), "RGB"),
        "rgba": Image.fromarray(g.randint(0, 256, (17, 19, 4)).astype(np.uint8), "RGBA"),
        "gray_alpha": Image.fromarray(g.randint(0, 256, (13, 29, 2)).astype(np.uint8), "LA"),
        "palette": pal,
        "palette_small": pal4,                                                     # PIL writes 4-bit indices
        "bilevel": Image.fromarray((g.randint(0, 2, (23, 37)) * 255).astype(np.uint8)).convert("1"),
        "smooth_rgb": Image.fromarray(np.stack([xray[:64, :80], xray[10:74, 5:85], 255 - xray[:64, :80]], -1), "RGB"),
        "one_pixel": Image.fromarray(np.array([[200]], dtype=np.uint8), "L"),
        "one_row": Image.fromarray(g.randint(0, 256, (1, 70, 3)).astype(np.uint8), "RGB"),
        "one_col": Image.fromarray(g.randint(0, 256, (70, 1)).astype(np.uint8), "L"),
    }
    rec = {"pillow": np.array(PIL.__version__)}
    names = []
    for tag, im in cases.items():
        for lvl in ((6,) if im.size[0] * im.size[1] > 10000 else (6, 1)):        # (large images once: fixture size)
            name = tag if lvl == 6 else tag + "_fast"
            bio = io.BytesIO()
            im.save(bio, format="PNG", compress_level=lvl)
            data = bio.getvalue()
            ref = Image.open(io.BytesIO(data))
            rec[f"{name}/png"] = np.frombuffer(data, dtype=np.uint8)
            rec[f"{name}/rgb"] = np.array(ref.convert("RGB"))
            rec[f"{name}/l"] = np.array(ref.convert("L"))
            names.append(name)
    # ---- streams PIL cannot WRITE (16-bit colour, Adam7) but reads: assembled here by hand, every PNG filter type in use, and
    # decoded by PIL for the expected arrays.  16-bit gray opens as mode I;16, whose convert("L" / "RGB") saturates at 255.
    import struct
    import zlib

    def paeth(a, b, c):
        p = a + b - c
        pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
        return a if pa <= pb and pa <= pc else (b if pb <= pc else c)

    def filt(ft, row, prev, bpp):
        out = bytearray(len(row))
        for i, v in enumerate(row):
            a = row[i - bpp] if i >= bpp else 0
            b = prev[i] if prev is not None else 0
            c = prev[i - bpp] if prev is not None and i >= bpp else 0
            pred = (0, a, b, (a + b) >> 1, paeth(a, b, c))[ft]
            out[i] = (v - pred) & 255
        return bytes(out)

    def pack_rows(samples, depth):
        """samples: uint array [h][w * channels] -> list of packed row byte strings"""
        rows = []
        for r in samples:
            if depth == 16:
                rows.append(r.astype(">u2").tobytes())
            elif depth == 8:
                rows.append(r.astype(np.uint8).tobytes())
            else:
                bits = np.unpackbits(r.astype(np.uint8)[:, None], axis=1)[:, 8 - depth:].reshape(-1)
                rows.append(np.packbits(bits).tobytes())
        return rows

    def make_png(samples, color, depth, interlace, plte=None):
        """samples [h][w][channels] unsigned ints -> PNG byte stream (filter type cycles 0..4 over the rows of every pass)"""
        h, w, ch = samples.shape
        bpp = max(1, depth * ch // 8)
        passes = [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)] if interlace else [(0, 0, 1, 1)]
        raw, k = b"", 0
        for x0, y0, dx, dy in passes:
            sub = samples[y0::dy, x0::dx]
            if sub.shape[0] == 0 or sub.shape[1] == 0:
                continue
            prev = None
            for row in pack_rows(sub.reshape(sub.shape[0], -1), depth):
                ft = k % 5
                raw += bytes([ft]) + filt(ft, row, prev, bpp)
                prev, k = row, k + 1

        def chunk(t, d):
            return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
        body = chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, color, 0, 0, 1 if interlace else 0))
        if plte is not None:
            body += chunk(b"PLTE", plte.astype(np.uint8).tobytes())
        z = zlib.compress(raw, 6)
        body += chunk(b"IDAT", z[:len(z) // 2]) + chunk(b"IDAT", z[len(z) // 2:])      # (two IDAT chunks: the payload is their concatenation)
        return b"\x89PNG\r\n\x1a\n" + body + chunk(b"IEND", b"")

    u16 = lambda *s_: g.randint(0, 65536, s_)
    low16 = np.where(g.rand(9, 13, 1) < 0.5, g.randint(0, 256, (9, 13, 1)), u16(9, 13, 1))      # half the pixels below 256: not saturated
    hand = {
        "gray16": make_png(low16, 0, 16, 0), "gray16_adam7": make_png(low16, 0, 16, 1),
        "rgb16": make_png(u16(7, 11, 3), 2, 16, 0), "rgb16_adam7": make_png(u16(7, 11, 3), 2, 16, 1),
        "gray_alpha16": make_png(u16(6, 9, 2), 4, 16, 0), "rgba16_adam7": make_png(u16(10, 9, 4), 6, 16, 1),
        "gray8_adam7": make_png(g.randint(0, 256, (33, 29, 1)), 0, 8, 1), "rgb8_adam7": make_png(g.randint(0, 256, (17, 40, 3)), 2, 8, 1),
        "rgba8_adam7_tiny": make_png(g.randint(0, 256, (3, 2, 4)), 6, 8, 1),                     # several empty passes
        "gray1_adam7": make_png(g.randint(0, 2, (19, 23, 1)), 0, 1, 1), "gray4_adam7": make_png(g.randint(0, 16, (12, 21, 1)), 0, 4, 1),
        "palette2_adam7": make_png(g.randint(0, 4, (14, 15, 1)), 3, 2, 1, plte=g.randint(0, 256, (4, 3))),
        "palette8_adam7": make_png(g.randint(0, 200, (16, 16, 1)), 3, 8, 1, plte=g.randint(0, 256, (200, 3))),
        "one_pixel_adam7": make_png(g.randint(0, 256, (1, 1, 3)), 2, 8, 1),
    }
    for name, data in hand.items():
        ref = Image.open(io.BytesIO(data))
        ref.load()
        rec[f"{name}/png"] = np.frombuffer(data, dtype=np.uint8)
        rec[f"{name}/rgb"] = np.array(ref.convert("RGB"))
        rec[f"{name}/l"] = np.array(Image.open(io.BytesIO(data)).convert("L"))
        names.append(name)
    rec["names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "png_cases.npz"), **rec)
    print("png_cases:", len(names), "streams, pillow", PIL.__version__)


def main(only=()):
    os.makedirs(OUT, exist_ok=True)
    C = _ref_classes()
    H = _ref_helpers()
    jobs = {
        "blocks": lambda: block_fixtures(C),
        "metrics": lambda: metric_fixture(H),
        "tester": lambda: tester_fixture(C, H),
        "pipeline": lambda: pipeline_fixture(C, H),
        "cls_metrics": cls_metric_fixture,
        "png": png_fixture,
        "AttentionUNet": lambda: model_fixture("AttentionUNet", C["AttentionUNet"], 64, True),
        "R2AttU_Net": lambda: model_fixture("R2AttU_Net", C["R2AttU_Net"], 32, True),
        "R2U_Net": lambda: model_fixture("R2U_Net", C["R2U_Net"], 32, True),
        "ResNet18": lambda: model_fixture("ResNet18", lambda: C["ResNet18"](num_classes=1000), 64, False, head_dropout=True, H=H),
        "ResNet50": lambda: model_fixture("ResNet50", lambda: C["ResNet50"](num_classes=1000), 64, False, head_dropout=True, H=H),
        "VGG16": lambda: model_fixture("VGG16", lambda: C["VGG16"](num_classes=1000), 32, False, head_dropout=True, H=H),
        "VGG19": lambda: model_fixture("VGG19", lambda: C["VGG19"](num_classes=1000), 32, False, head_dropout=True, H=H),
        "ResNetUnet": resnet_unet_fixture,
        "train_traj_seg": lambda: train_traj_seg(C, H),
        "train_traj_R2AttU_Net": lambda: train_traj_seg(C, H, "R2AttU_Net", lr=1e-5, hw=64),
        "train_traj_R2U_Net": lambda: train_traj_seg(C, H, "R2U_Net", lr=1e-5, hw=64),
        "train_traj_cls": lambda: train_traj_cls(C, H),
    }
    for k in (only or jobs):
        jobs[k]()


if __name__ == "__main__":
    main(sys.argv[1:])       # no arguments: every fixture; otherwise the named ones (e.g. `make_golden.py VGG19 cls_metrics`)
