"""-m gpu: whole-model parity of the MI355X launch-plan path.

fp32 mode is checked (a) against the golden vectors produced by the REFERENCE classes
(tests/golden/model_*.npz: eval/train logits, loss, per-parameter gradient norms, BN buffers,
post-AdamW parameter norms) and (b) element-wise against the CPU oracle run on this box.
Tolerance: 1e-3 relative (north-star bound for fp32); observed errors are ~1e-5.
bf16 mode is judged on segmentation agreement (Dice of the binarised masks vs the fp32 oracle)."""
import os

import numpy as np
import pytest
import torch

from oracle import nets
from oracle import train as otrain

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda:0"
RTOL = 1e-3


def _rel(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


def _build(name, dtype):
    from mi355 import engine
    from utils import helpers
    if name == "AttentionUNet":
        from models.segmentation_models.AttentionUNet import AttentionUNet as C
        m, kw = C(), {}
    elif name == "R2AttU_Net":
        from models.segmentation_models.R2AttU_Net import R2AttU_Net as C
        m, kw = C(), {}
    elif name == "R2U_Net":
        from models.segmentation_models.R2U_Net import R2U_Net as C
        m, kw = C(), {}
    elif name in ("ResNet18", "ResNet50"):
        from models.classification_models import ResNet
        m = getattr(ResNet, name)(num_classes=1000)
        helpers.add_dropout_to_fc(m, p=0.0)
        kw = {"head_dropout": True}
    elif name in ("VGG16", "VGG19"):
        from models.classification_models import VGG
        m = getattr(VGG, name)(num_classes=1000)
        helpers.add_dropout_to_fc(m, p=0.0)
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        kw = {"head_dropout": True}
    sd = nets.closed_form_state(name, **kw)
    m.load_state_dict(sd)
    m.compute_dtype = dtype
    return m.to(DEV), sd, kw


# (median, maximum) deviation of the per-tensor gradient NORMS from the reference's, per model: twice what the HIP path measures on
# these fixtures (printed by the test); two fp32 evaluations of a network with ReLU / max-pool kinks differ by this much whatever
# the implementation — the CPU oracle with the GPU's masks replayed agrees to 1e-3 everywhere (tests/test_gpu_kinks.py)
GRAD_NORM_BOUNDS = {   # measured (r03): median / max
    "AttentionUNet": (1e-3, 8e-2),      # 3.6e-4 / 4.6e-2 (103 tensors; the maximum is one small tensor behind a handful of flipped masks)
    "R2AttU_Net": (2e-3, 8e-2),         # 6.7e-4 / 6.1e-2
    "R2U_Net": (1e-4, 4e-3),            # 1.6e-5 / 1.3e-3
    "ResNet18": (1e-5, 2e-5),           # 1.0e-6 / 2.6e-6: no mask flips on this fixture, the kernels' fp32 rounding alone
    "ResNet50": (2e-3, 3e-2),           # 5.9e-4 / 1.0e-2
    "VGG16": (2e-6, 2e-6),              # 1.2e-7 / 2.1e-7
    "VGG19": (2e-6, 3e-6),              # 4.3e-7 / 6.4e-7
}


@pytest.mark.parametrize("name", ["AttentionUNet", "R2AttU_Net", "R2U_Net", "ResNet18", "ResNet50", "VGG16", "VGG19"])
def test_fp32_model_matches_reference_golden_and_oracle(name):
    from mi355 import nn as mnn, optim as moptim
    z = np.load(os.path.join(G, f"model_{name}.npz"))
    hw, seg, lr = int(z["hw"]), bool(z["seg"]), float(z["lr"])
    m, sd, kw = _build(name, torch.float32)
    x, mask = otrain.closed_form_input(2, hw)
    y = mask if seg else torch.tensor([1, 2])
    xd, yd = x.to(DEV), y.to(DEV)

    m.eval()
    with torch.no_grad():
        ev = m(xd)
    assert _rel(ev.cpu().numpy(), z["logits_eval"]) < RTOL

    m.train()
    crit = mnn.BCEWithLogitsLoss() if seg else mnn.CrossEntropyLoss(label_smoothing=0.1)
    opt = moptim.AdamW(m.parameters(), lr=lr, weight_decay=5e-4)
    opt.zero_grad(set_to_none=True)
    out = m(xd)
    loss = crit(out, yd)
    loss.backward()
    torch.cuda.synchronize()
    assert _rel(out.detach().cpu().numpy(), z["logits_train"]) < 1e-4      # measured 1.4e-5 (R2AttU_Net) .. 5e-5 (AttentionUNet) vs fp64
    assert abs(float(loss.detach()) - float(z["loss"])) < RTOL * max(1.0, abs(float(z["loss"])))

    # Gradients against the REFERENCE's values.  Both sides are fp32 evaluations of a discontinuous function (DESIGN.md section 5:
    # a dozen ReLU masks differ between any two evaluations and move every gradient by ~1e-3..1e-2), so the golden gradient
    # norms are matched to 8 % here; the sharp statement — every gradient tensor to rounding accuracy against the fp64 oracle
    # on the GPU's own masks — is tests/test_gpu_kinks.py, on these same fixtures.
    names = [str(s) for s in z["param_names"]]
    params = dict(m.named_parameters())
    assert names == list(params.keys())
    gn = np.array([float(params[k].grad.double().norm()) for k in names])
    big = z["grad_norm"] > 1e-5 * z["grad_norm"].max()
    dev = np.abs(gn[big] / z["grad_norm"][big] - 1)
    print(f"golden gradient norms {name}: median deviation {np.median(dev):.2e}, max {dev.max():.2e} over {int(big.sum())} tensors")
    med_tol, max_tol = GRAD_NORM_BOUNDS[name]
    assert np.median(dev) <= med_tol and dev.max() <= max_tol, (np.median(dev), dev.max())
    # BN buffers after one train-mode forward
    msd = m.state_dict()
    for k, l2 in zip([str(s) for s in z["buffer_names"]], z["buffer_l2_after"]):
        assert abs(float(msd[k].double().norm()) - l2) < RTOL * max(l2, 1e-3), k
    for k in z.files:
        if k.startswith("full_"):
            assert _rel(msd[k[5:]].cpu().numpy(), z[k]) < RTOL

    # clip + AdamW
    total = moptim.clip_grad_norm_(m.parameters(), 1.0)
    opt.step()
    torch.cuda.synchronize()
    assert abs(float(total) - float(z["total_grad_norm"])) < 3e-2 * float(z["total_grad_norm"])
    numel = np.array([params[k].numel() for k in names])
    l2 = np.array([float(params[k].detach().double().norm()) for k in names])
    assert np.all(np.abs(l2 - z["param_l2_after"]) <= 1e-5 * l2 + 0.3 * lr * np.sqrt(numel))


@pytest.mark.parametrize("name", ["AttentionUNet", "R2AttU_Net", "R2U_Net"])
def test_fp32_train_loop_matches_reference_trajectory(name, tmp_path, capsys):
    """utils.helpers.train (the drop-in driver) on the fixed synthetic loader vs the per-epoch
    log of the reference's own train() (tests/golden/train_traj_*.npz, lr 1e-5), for the recurrent nets (64 x 64) with full
    tensors of the final state — six optimiser steps through the shared-weight convolutions (one mi355_conv2d_wgrad_multi launch
    per convolution and step) and 36 ordered running-statistics updates per recurrent BatchNorm (R2AttU_Net.py:41-44).  The
    bounds are those of tests/test_oracle_pins.py::test_train_trajectory_seg: what two CPU evaluations of the protocol differ by."""
    from torch.utils.data import DataLoader, TensorDataset
    from utils import helpers
    z = np.load(os.path.join(G, f"train_traj_{name}.npz"))
    hw, epochs, lr = int(z["hw"]), int(z["epochs"]), float(z["lr"])
    m, sd, _ = _build(name, torch.float32)
    b = [otrain.synthetic_batch(4, hw, seed=s) for s in (0, 1, 2)]
    tr = DataLoader(TensorDataset(torch.cat([b[0][0], b[1][0]]), torch.cat([b[0][1], b[1][1]])), batch_size=4, shuffle=False)
    va = DataLoader(TensorDataset(b[2][0], b[2][1]), batch_size=4, shuffle=False)
    best = helpers.train(m, tr, va, torch.device(DEV), epochs, lr, name, str(tmp_path), seg=True)
    text = capsys.readouterr().out
    import re
    rows = re.findall(r"Ep(\d+): TrainLoss ([\d.]+) \| ValLoss ([\d.]+) \| IoU ([\d.]+)", text)
    assert len(rows) == epochs
    for row, ref in zip(rows, z["log"]):
        got = [float(v) for v in row]
        assert abs(got[1] - ref[1]) <= 2e-3 and abs(got[2] - ref[2]) <= 2e-3 * max(1, ref[2]) and abs(got[3] - ref[3]) <= 5e-3, (got, ref)
    assert abs(best - float(z["best"])) < 2e-3 * float(z["best"])
    saved = torch.load(os.path.join(str(tmp_path), f"{name}_best_loss.pt"))
    assert list(saved.keys()) == [str(s) for s in z["names"]]
    names = [str(s) for s in z["names"]]
    msd = m.state_dict()
    l2 = np.array([float(msd[k].double().norm()) for k in names])
    assert np.allclose(l2, z["final_l2"], rtol=5e-3)
    worst = {}
    for k in z.files:
        if k.startswith("full_"):
            worst[k[5:]] = _rel(msd[k[5:]].cpu().numpy(), z[k])
    print("trajectory", name, "full-tensor deviations", {k: f"{v:.1e}" for k, v in worst.items()})
    from test_oracle_pins import FULL_TOL
    for k, v in worst.items():
        assert v < FULL_TOL(k), (k, v)
    if "final_val_logits" in z.files:
        m.eval()
        with torch.no_grad():
            ev = m(b[2][0].to(DEV)).float().cpu().numpy()
        assert _rel(ev, z["final_val_logits"]) < 3e-2


def test_fp32_cls_two_stage_train_matches_reference(tmp_path, capsys):
    """Stage-1 (head only) -> stage-2 switch of helpers.train on ResNet18 vs the reference log."""
    from torch.utils.data import DataLoader, TensorDataset
    from utils import helpers
    z = np.load(os.path.join(G, "train_traj_ResNet18.npz"))
    hw, epochs, lr = int(z["hw"]), int(z["epochs"]), float(z["lr"])
    m, sd, _ = _build("ResNet18", torch.float32)
    b = [otrain.synthetic_batch(4, hw, seed=10 + s, classes=3) for s in (0, 1, 2)]
    tr = DataLoader(TensorDataset(torch.cat([b[0][0], b[1][0]]), torch.cat([b[0][1], b[1][1]])), batch_size=4, shuffle=False)
    va = DataLoader(TensorDataset(b[2][0], b[2][1]), batch_size=4, shuffle=False)
    best = helpers.train(m, tr, va, torch.device(DEV), epochs, lr, "ResNet18", str(tmp_path), seg=False, cls_head_name="fc")
    text = capsys.readouterr().out
    import re
    rows = re.findall(r"Ep(\d+): TrainLoss ([\d.]+) \(Acc ([\d.]+)%\) \| ValLoss ([\d.]+) \| ValAcc ([\d.]+)%", text)
    assert len(rows) == len(z["log"])
    for row, ref in zip(rows, z["log"]):
        got = [float(v) for v in row]
        # 14 optimisation steps of a chaotic fp32 trajectory: losses drift apart slowly, accuracies are exact
        assert abs(got[1] - ref[1]) <= 2e-2 * max(1, ref[1]) and abs(got[2] - ref[2]) < 1e-6
        assert abs(got[3] - ref[3]) <= 2e-2 * max(1, ref[3]) and abs(got[4] - ref[4]) < 1e-6
    assert abs(best - float(z["best"])) < 1e-6


def test_fp32_well_conditioned_resnet18_is_elementwise_exact():
    """ResNet18 at batch 8 is well conditioned (reference fp32 vs fp64: 5e-6), so every gradient tensor
    of the HIP fp32 path must agree with the fp64 oracle to 1e-4 — 10x inside the north-star bound."""
    from mi355 import nn as mnn
    m, sd, kw = _build("ResNet18", torch.float32)
    x, y = otrain.synthetic_batch(8, 64, seed=3, classes=3)
    m.train()
    out = m(x.to(DEV))
    mnn.CrossEntropyLoss(label_smoothing=0.1)(out, y.to(DEV)).backward()
    torch.cuda.synchronize()
    sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    _, o64, g64 = otrain.forward_backward("ResNet18", sd64, x.double(), y, False)
    assert _rel(out.detach().cpu().numpy(), o64.numpy()) < 1e-4
    gmax = max(float(v.abs().max()) for v in g64.values())
    for k, p in m.named_parameters():
        ref = g64[k].numpy()
        assert np.abs(p.grad.cpu().numpy() - ref).max() < 1e-4 * np.abs(ref).max() + 1e-6 * gmax, k


def test_state_dict_roundtrip_and_no_cpu_fallback():
    from models.segmentation_models.AttentionUNet import AttentionUNet
    m = AttentionUNet()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 3, 32, 32))
    m = m.to(DEV)
    m.eval()
    x = torch.randn(1, 3, 32, 32, device=DEV)
    with torch.no_grad():
        a = m(x)
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    m2 = AttentionUNet()
    m2.load_state_dict(sd)
    m2 = m2.to(DEV).eval()
    with torch.no_grad():
        b = m2(x)
    assert torch.equal(a, b)


def test_resnet_unet_matches_oracle_fp32():
    """ResNetUnet (config 2).  The reference class needs torchvision (absent), so this model is checked
    against the oracle only (parity unpinned at the torchvision-encoder boundary): logits,
    loss, frozen-encoder semantics, BN buffers and — against the fp64 oracle on the GPU's own ReLU / max-pool decisions — the
    decoder gradients, including both ConvTranspose2d backward paths."""
    from mi355 import nn as mnn
    from models.segmentation_models.ResnetUnet import ResNetUnet
    name = "ResNetUnet"
    sd = nets.closed_form_state(name)
    m = ResNetUnet()
    m.load_state_dict(sd)
    m.compute_dtype = torch.float32
    m = m.to(DEV)
    x, mask = otrain.closed_form_input(2, 64)
    m.eval()
    with torch.no_grad():
        ev = m(x.to(DEV)).cpu()
        ev_ref = nets.resnet_unet({k: v.clone() for k, v in sd.items()}, x, False)
    assert _rel(ev.numpy(), ev_ref.numpy()) < RTOL
    m.train()
    out = m(x.to(DEV))
    loss = mnn.BCEWithLogitsLoss()(out, mask.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    frozen = {k for k, p in m.named_parameters() if not p.requires_grad}
    assert frozen and all(k.startswith("encoder") for k in frozen)
    assert all(p.grad is None for k, p in m.named_parameters() if k in frozen)

    # the fp64 oracle on the GPU's own ReLU / max-pool decisions (tests/test_gpu_kinks.py): the two evaluate the same smooth function
    from gpu_util import gpu_kinks
    relu, pool = gpu_kinks(out._mi355_plan)
    s64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    pk = [k for k in nets.param_keys(s64) if k not in frozen]
    for k in pk:
        s64[k].requires_grad_(True)
    nets.Kinks.start("replay", relu, pool)
    try:
        o64 = nets.resnet_unet(s64, x.double(), True)
        l64 = otrain.bce_with_logits(o64, mask.double())
        l64.backward()
    finally:
        _, _, used = nets.Kinks.stop()
    assert used == (len(relu), len(pool)) and len(pool) == 1, used
    assert _rel(out.detach().cpu().numpy(), o64.detach().numpy()) < 5e-4          # ~110 fp32 layers (measured 1.4e-4)
    assert abs(float(loss.detach()) - float(l64.detach())) < 2e-5
    params = dict(m.named_parameters())
    gmax = max(float(s64[k].grad.abs().max()) for k in pk)
    errs = {}
    for k in pk:
        ref = s64[k].grad
        sc = float(ref.abs().max())
        if sc >= 1e-6 * gmax:
            errs[k] = float((params[k].grad.cpu().double() - ref).abs().max()) / sc
    e = np.array(list(errs.values()))
    # (the decoder sees the frozen ResNet50 encoder's fp32 activations, 1.4e-4 off fp64 at the logits: measured median 6e-5, max 5e-4)
    assert np.median(e) <= 2e-4 and e.max() <= 2e-3, (np.median(e), max(errs, key=errs.get), e.max())
    msd = m.state_dict()
    for k, v in s64.items():
        if k.endswith(("running_mean", "running_var")):
            assert _rel(msd[k].cpu().numpy(), v.detach().numpy()) < 1e-4, k

    # ... and the REFERENCE's own ResNetUnet (ResnetUnet.py:29-83 run over a plain-torch container of torchvision's layout,
    # tests/golden/model_ResNetUnet.npz): logits, loss, which parameters are trainable / received a gradient, every BatchNorm
    # buffer after the train-mode forward (the frozen encoder's too), the decoder's gradient norms (two fp32 evaluations with
    # their own mask flips: bounds of 2-3x what is measured, as for the other model fixtures)
    z = np.load(os.path.join(G, "model_ResNetUnet.npz"), allow_pickle=False)
    names = [str(s_) for s_ in z["frozen/param_names"]]
    assert names == [k for k, _ in m.named_parameters()] and [str(s_) for s_ in z["state_keys"]] == list(msd.keys())
    assert np.array_equal(z["frozen/requires_grad"], np.array([params[k].requires_grad for k in names]))
    assert np.array_equal(z["frozen/has_grad"], np.array([params[k].grad is not None for k in names]))
    assert _rel(ev.numpy(), z["frozen/logits_eval"]) < RTOL
    assert _rel(out.detach().cpu().numpy(), z["frozen/logits_train"]) < 5e-4
    assert abs(float(loss.detach()) - float(z["frozen/loss"])) < 2e-5
    bufs = [str(b) for b in z["frozen/buffer_names"]]
    l2 = np.array([float(msd[k].double().norm()) for k in bufs])
    assert np.allclose(l2, z["frozen/buffer_l2"], rtol=1e-4)
    want = z["frozen/grad_norm"]
    gn = np.array([float(params[k].grad.double().norm()) if params[k].grad is not None else -1.0 for k in names])
    big = want > 1e-6 * want.max()
    dev = np.abs(gn[big] / want[big] - 1)
    assert np.median(dev) <= 2e-3 and dev.max() <= 5e-2, (np.median(dev), dev.max())


def test_tester_functions_match_oracle_metrics(capsys):
    """utils.tester (tester.py:92-312 counterpart): metric helpers and the segmentation eval loop."""
    from torch.utils.data import DataLoader, TensorDataset
    from utils import tester
    g = torch.Generator().manual_seed(4)
    pred = torch.rand(1, 40, 40, generator=g); tgt = (torch.rand(1, 40, 40, generator=g) > 0.6).float()
    ref = otrain.seg_metrics(pred, tgt)
    got_cpu = tester.calculate_segmentation_metrics(pred, tgt)
    got_gpu = tester.calculate_segmentation_metrics(pred.to(DEV), tgt.to(DEV))
    for k in ref:
        assert abs(ref[k] - got_cpu[k]) < 1e-9 and abs(ref[k] - got_gpu[k]) < 1e-4, k
    assert abs(tester.calculate_dice(pred.to(DEV), tgt.to(DEV)) * 100 - ref["dice"]) < 1e-4
    m, sd, _ = _build("AttentionUNet", torch.float32)
    xs, ms = zip(*[otrain.synthetic_batch(3, 32, seed=s) for s in (5, 6)])
    dl = DataLoader(TensorDataset(torch.cat(xs), torch.cat(ms)), batch_size=3)
    avg = tester.test_segmentation_model(m, dl, torch.device(DEV), "AttentionUNet")
    capsys.readouterr()
    tot = {k: 0.0 for k in avg}
    with torch.no_grad():
        for x_, m_ in zip(xs, ms):
            o = torch.sigmoid(nets.attention_unet({k: v.clone() for k, v in sd.items()}, x_, False))
            for i in range(o.shape[0]):
                mm = otrain.seg_metrics(o[i], m_[i])
                for k in tot:
                    tot[k] += mm[k] / 6
    for k in avg:
        assert abs(avg[k] - tot[k]) < 0.2, (k, avg[k], tot[k])        # percent; a handful of pixels sit at p = 0.5 +- 1e-4
    pr = np.array([0, 1, 2, 2, 1, 0, 0]); lb = np.array([0, 1, 1, 2, 1, 2, 0])
    cm = tester.calculate_classification_metrics(pr, lb)
    assert abs(cm["accuracy"] - 100 * 5 / 7) < 1e-9 and cm["confusion_matrix"].sum() == 7


def test_tester_eval_loops_match_the_reference_tester(capsys):
    """utils.tester.test_segmentation_model / test_classification_model on the GPU against the dictionaries AND the printed text
    the REFERENCE's loops (tester.py:197-312, run by oracle/make_golden.py on its own AttentionUNet / ResNet18 at the same
    closed-form weights and loaders) returned; the metric helpers on device tensors against the reference's values."""
    from torch.utils.data import DataLoader, TensorDataset
    from utils import tester
    z = np.load(os.path.join(G, "tester.npz"), allow_pickle=False)
    keys = [str(k) for k in z["seg_keys"]]
    for tag in [str(s_) for s_ in z["seg/names"]]:
        p, t = torch.from_numpy(z[f"seg/{tag}/pred"]).to(DEV), torch.from_numpy(z[f"seg/{tag}/target"]).to(DEV)
        got = tester.calculate_segmentation_metrics(p, t)
        assert np.allclose([got[k] for k in keys], z[f"seg/{tag}/metrics"], rtol=2e-5, atol=2e-5), tag
        assert abs(tester.calculate_dice(p, t) - float(z[f"seg/{tag}/dice"])) < 1e-6
    m, _, _ = _build("AttentionUNet", torch.float32)
    xs, ms = zip(*[otrain.synthetic_batch(3, 32, seed=s_) for s_ in (5, 6)])
    dl = DataLoader(TensorDataset(torch.cat(xs), torch.cat(ms)), batch_size=3)
    avg = tester.test_segmentation_model(m, dl, torch.device(DEV), "AttentionUNet")
    out = capsys.readouterr().out
    assert list(avg) == keys
    # percent; one pixel of the 6 x 1024 is 0.016 and a handful sit at p = 0.5 +- 1e-4
    assert np.abs(np.array([avg[k] for k in keys]) - z["segloop/metrics"]).max() < 0.2, (avg, z["segloop/metrics"])
    ref_text = str(z["segloop/stdout"])
    strip = lambda txt: [ln.split(":")[0] if "%" in ln else ln for ln in txt.splitlines()]      # same lines, numbers aside
    assert strip(out) == strip(ref_text)

    cm, _, _ = _build("ResNet18", torch.float32)
    with torch.no_grad():
        cm.fc[1].bias -= torch.from_numpy(z["clsloop/bias_shift"]).to(DEV)
    x, y = otrain.synthetic_batch(12, 64, seed=20, classes=3)
    assert z["clsloop/logit_margin"].min() > 0.05                      # the fixture's decisions are clear of fp32 noise
    res = tester.test_classification_model(cm, DataLoader(TensorDataset(x, y), batch_size=5), torch.device(DEV), "ResNet18")
    out = capsys.readouterr().out
    for k in ("accuracy", "precision", "recall", "f1"):
        assert abs(res[k] - float(z[f"clsloop/{k}"])) < 1e-9, k
    for k in ("precision_per_class", "recall_per_class", "f1_per_class", "confusion_matrix"):
        assert np.allclose(res[k], z[f"clsloop/{k}"], atol=1e-9), k
    assert out == str(z["clsloop/stdout"])                              # identical decisions => identical text


def test_test_all_models_walks_the_weight_directories(tmp_path, capsys):
    """utils.tester.test_all_models (tester.py:513-735): evaluates the checkpoints that exist under the reference's file
    names, skips the missing ones with the reference's warning, takes the dataset-not-found branch without a loader, and
    returns the same dictionary the per-model loops return; print_summary / save_results_to_csv accept it."""
    from torch.utils.data import DataLoader, TensorDataset
    from utils import tester
    from utils.helpers import get_class_model, get_seg_model
    seg_dir, cls_dir = tmp_path / "seg", tmp_path / "cls"
    seg_dir.mkdir(); cls_dir.mkdir()
    torch.manual_seed(11)
    seg = get_seg_model("AttentionUNet")
    torch.save(seg.state_dict(), seg_dir / "AttentionUNet_best_loss.pt")
    cls, _ = get_class_model("ResNet18")
    torch.save(cls.state_dict(), cls_dir / "ResNet18_best_acc.pt")
    (cls_dir / "CLIP_best_acc.pt").write_bytes(b"")
    xs, ms = zip(*[otrain.synthetic_batch(2, 32, seed=s) for s in (1, 2)])
    seg_dl = DataLoader(TensorDataset(torch.cat(xs), torch.cat(ms)), batch_size=2)
    cls_dl = DataLoader(TensorDataset(torch.cat(xs), torch.tensor([0, 1, 2, 1])), batch_size=4)
    res = tester.test_all_models("cuda", 4, cls_loader=cls_dl, seg_loader=seg_dl, cls_weights_dir=str(cls_dir),
                                 seg_weights_dir=str(seg_dir))
    out = capsys.readouterr().out
    assert set(res) == {"ResNet18", "AttentionUNet"}
    assert "Weights not found for VGG16" in out and "Weights not found for R2Unet" in out and "CLIP is a hub model" in out
    direct = tester.test_segmentation_model(seg.to(DEV), seg_dl, torch.device(DEV), "AttentionUNet")
    assert all(abs(direct[k] - res["AttentionUNet"][k]) < 1e-9 for k in direct)
    assert res["ResNet18"]["confusion_matrix"].sum() == 4
    none = tester.test_all_models("cuda", 4, cls_weights_dir=str(cls_dir), seg_weights_dir=str(seg_dir))
    out = capsys.readouterr().out
    assert none == {} and "Classification test dataset not found" in out and "Segmentation test dataset not found" in out
    # with a dataset directory in the reference's layout the loaders are built from DATA_ROOT/splits/test.csv (tester.py:567-580,
    # 651-666): classification at batch_size, segmentation at batch_size // 2, PNGs decoded natively, transforms on the GPU
    pytest.importorskip("PIL.Image")
    from test_dataset_cpu import make_tree
    make_tree(str(tmp_path / "dataset"), n=7, splits=("train", "test"))
    old_root = tester.DATA_ROOT
    tester.DATA_ROOT = str(tmp_path / "dataset")
    try:
        own = tester.test_all_models("cuda", 4, cls_weights_dir=str(cls_dir), seg_weights_dir=str(seg_dir))
    finally:
        tester.DATA_ROOT = old_root
    out = capsys.readouterr().out
    assert set(own) == {"ResNet18", "AttentionUNet"} and "Classification Test Dataset: 7 samples" in out
    assert "Segmentation Test Dataset: 6 samples" in out and own["ResNet18"]["confusion_matrix"].sum() == 7
    tester.print_summary(res)
    tester.save_results_to_csv(res, str(tmp_path / "c.csv"), str(tmp_path / "s.csv"))
    assert (tmp_path / "c.csv").read_text().startswith("Model,accuracy,precision,recall,f1\nResNet18,")
    assert (tmp_path / "s.csv").read_text().startswith("Model,iou,dice,pixel_accuracy,precision,recall,f1\nAttentionUNet,")


def test_amp_overflow_in_a_later_param_group_skips_every_group():
    """Two parameter groups on one model, fp16 + GradScaler, NO clip_grad_norm_ call: an inf gradient in the SECOND group must be
    found (the inf / nan check covers the union of all groups, like torch's GradScaler.step over the whole optimizer), every
    group skips the step, the scale backs off, and a clean step afterwards moves both groups."""
    from mi355 import amp as mamp, nn as mnn, optim as moptim
    from models.segmentation_models.AttentionUNet import AttentionUNet
    torch.manual_seed(5)
    m = AttentionUNet()
    m.compute_dtype = torch.float16
    m = m.to(DEV).train()
    x, y = otrain.synthetic_batch(2, 32, seed=1)
    crit = mnn.BCEWithLogitsLoss()
    crit(m(x.to(DEV)), y.to(DEV)).backward()                       # (flattens the parameters)
    ps = list(m.parameters())
    cut = len(ps) // 2
    opt = moptim.AdamW([{"params": ps[:cut]}, {"params": ps[cut:], "lr": 2e-3}], lr=1e-3, weight_decay=5e-4)
    sc = mamp.GradScaler(init_scale=1024.0, enabled=True)

    def one_step(poison):
        opt.zero_grad(set_to_none=True)
        sc.scale(crit(m(x.to(DEV)), y.to(DEV))).backward()
        if poison:
            ps[-3].grad.view(-1)[7] = float("inf")                  # a parameter of the second group
        sc.unscale_(opt)
        sc.step(opt)
        sc.update()
        torch.cuda.synchronize()

    before = [p.detach().clone() for p in ps]
    one_step(poison=True)
    assert all(torch.equal(a, b.detach()) for a, b in zip(before, ps)), "a group stepped on an overflowed gradient"
    assert sc.get_scale() == 512.0 and [int(st["step"]) for st in opt._st] == [0, 0]
    one_step(poison=False)
    assert [int(st["step"]) for st in opt._st] == [1, 1] and sc.get_scale() == 512.0
    moved = [not torch.equal(a, b.detach()) for a, b in zip(before, ps)]
    assert any(moved[:cut]) and any(moved[cut:]) and all(torch.isfinite(p).all() for p in ps)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_resume_from_checkpoint_is_bit_exact(tmp_path, dtype):
    """N3: a run restarted from {model, optimiser, loss scaler} state dicts written with torch.save
    continues exactly where the uninterrupted run goes — weights, BatchNorm buffers and AdamW moments bit for bit (the
    kernels reduce in fixed orders).  The reference's own checkpoint is the model state_dict alone (helpers.py:394-400)."""
    from mi355 import amp as mamp, nn as mnn, optim as moptim
    from models.segmentation_models.AttentionUNet import AttentionUNet
    batches = [otrain.synthetic_batch(2, 32, seed=s) for s in range(4)]

    def make(state=None):
        torch.manual_seed(3)
        m = AttentionUNet()
        m.compute_dtype = dtype
        m = m.to(DEV).train()
        opt = moptim.AdamW(m.parameters(), lr=1e-3, weight_decay=5e-4)
        sc = mamp.GradScaler(enabled=dtype == torch.float16)
        if state is not None:
            m.load_state_dict(state["model"])
            opt.load_state_dict(state["opt"])
            sc.load_state_dict(state["scaler"])
        return m, opt, sc

    def run(m, opt, sc, bs):
        crit = mnn.BCEWithLogitsLoss()
        for x, y in bs:
            opt.zero_grad(set_to_none=True)
            loss = crit(m(x.to(DEV)), y.to(DEV))
            sc.scale(loss).backward()
            sc.unscale_(opt)
            moptim.clip_grad_norm_(m.parameters(), max_norm=1.0)
            sc.step(opt)
            sc.update()
        torch.cuda.synchronize()

    m, opt, sc = make()
    run(m, opt, sc, batches)
    want = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    want_opt = opt.state_dict()

    m, opt, sc = make()
    run(m, opt, sc, batches[:2])
    torch.save({"model": m.state_dict(), "opt": opt.state_dict(), "scaler": sc.state_dict()}, tmp_path / "ckpt.pt")
    del m, opt, sc
    m, opt, sc = make(torch.load(tmp_path / "ckpt.pt", map_location=DEV))
    run(m, opt, sc, batches[2:])
    got = m.state_dict()
    for k, v in want.items():
        assert torch.equal(got[k].cpu(), v), k
    go = opt.state_dict()
    for i, st in want_opt["state"].items():
        for key in ("exp_avg", "exp_avg_sq"):
            assert torch.equal(go["state"][i][key].cpu(), st[key].cpu()), (i, key)
        assert float(go["state"][i]["step"]) == float(st["step"]) == 4.0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_recurrent_weight_gradients_one_launch_matches_per_application(monkeypatch, dtype):
    """The shared convolution of a recurrent block gets its weight gradient from ONE mi355_conv2d_wgrad_multi launch over the
    six (x, dy) pairs (levels the nine-tap kernel serves: 32x32 and, two images at a time, 16x16) — same fp32 sums as six
    mi355_conv2d_wgrad + reduce rounds (MI355_WGRAD_MULTI=0), up to the order of the additions."""
    from mi355 import nn as mnn
    from models.segmentation_models.R2AttU_Net import R2AttU_Net
    x, y = otrain.synthetic_batch(2, 32, seed=21)
    grads, counts = [], []
    for multi in ("1", "0"):
        monkeypatch.setenv("MI355_WGRAD_MULTI", multi)
        torch.manual_seed(5)
        m = R2AttU_Net()
        m.compute_dtype = dtype
        m = m.to(DEV).train()
        loss = mnn.BCEWithLogitsLoss()(m(x.to(DEV)), y.to(DEV))
        loss.backward()
        torch.cuda.synchronize()
        grads.append({k: p.grad.detach().float().cpu().clone() for k, p in m.named_parameters()})
        plan = [p for p in m.engine.plans.values() if p.training][0]
        counts.append(sum(l.name == "mi355_conv2d_wgrad_multi" for l in plan.bwd))
    assert counts[0] > 0 and counts[1] == 0
    for k, g in grads[0].items():
        ref = grads[1][k]
        assert float((g - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 1e-12, k


@pytest.mark.parametrize("name", ["VGG16_BN", "VGG19_BN"])
def test_vgg_bn_matches_oracle_fp32(name):
    """torchvision-layout vgg16_bn / vgg19_bn (the models helpers.py:158-166 requests from the hub; vgg16_bn is config 5's classifier).
    torchvision is absent, so the check is against the oracle only (parity unpinned at this boundary): eval and
    train logits, loss and every gradient tensor against the fp64 oracle on the GPU's own kink decisions.  64x64 input -> 2x2 feature
    map, so AdaptiveAvgPool2d((7,7)) runs its up-sampling branch (overlapping / repeated windows)."""
    from mi355 import nn as mnn
    from models.classification_models import VGG
    from utils.helpers import add_dropout_to_fc
    sd = nets.closed_form_state(name, num_classes=3, head_dropout=True)
    m = getattr(VGG, name)(num_classes=1000)
    assert add_dropout_to_fc(m, p=0.0) == "classifier"
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    m.load_state_dict(sd)
    m.compute_dtype = torch.float32
    m = m.to(DEV)
    x, _ = otrain.closed_form_input(2, 64)
    y = torch.tensor([1, 2])
    m.eval()
    with torch.no_grad():
        ev = m(x.to(DEV)).cpu()
        ev_ref = nets.NETS[name]({k: v.clone() for k, v in sd.items()}, x, False)
    assert _rel(ev.numpy(), ev_ref.numpy()) < RTOL
    m.train()
    out = m(x.to(DEV))
    loss = mnn.CrossEntropyLoss(label_smoothing=0.1)(out, y.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    # 260 k max-pool windows, the tightest ~1e-7 (relative) wide: any fp32 evaluation may route such a window's gradient to the
    # runner-up.  The fp64 oracle therefore replays the GPU's ReLU masks and pool arg-maxes (tests/test_gpu_kinks.py).
    from gpu_util import gpu_kinks, replayed_oracle
    relu, pool = gpu_kinks(out._mi355_plan)
    l64, o64, g64 = replayed_oracle(name, sd, x, y, relu, pool, seg=False)
    assert len(pool) == 5
    assert _rel(out.detach().cpu().numpy(), o64.numpy()) < 1e-4
    assert abs(float(loss.detach()) - l64) < 1e-5 * max(1.0, abs(l64))
    params = dict(m.named_parameters())
    gmax = max(float(v.abs().max()) for v in g64.values())
    errs = {}
    for k, ref in g64.items():
        sc = float(ref.abs().max())
        if sc >= 1e-6 * gmax:
            errs[k] = float((params[k].grad.cpu().double() - ref).abs().max()) / sc
    e = np.array(list(errs.values()))
    assert np.median(e) <= 5e-5 and e.max() <= 1e-3, (np.median(e), max(errs, key=errs.get), e.max())


@pytest.mark.parametrize("name,ctor_kw,spec_kw,cin", [
    ("AttentionUNet", dict(in_channel=3, out_channel=3), dict(out_channels=3), 3),
    ("R2U_Net", dict(in_channels=1, out_channels=2, t=2), dict(in_channels=1, out_channels=2), 1),
])
def test_constructor_channel_arguments(name, ctor_kw, spec_kw, cin):
    """Constructor surface beyond the defaults (AttentionUNet.py:56-84 ``out_channel``; R2U_Net.py:50-83 ``in_channels`` /
    ``out_channels`` / ``t``): K-channel logit heads and non-RGB inputs run on the HIP path and match the oracle."""
    from mi355 import nn as mnn
    if name == "AttentionUNet":
        from models.segmentation_models.AttentionUNet import AttentionUNet as C
    else:
        from models.segmentation_models.R2U_Net import R2U_Net as C
    t = ctor_kw.get("t", 5)
    sd = nets.closed_form_state(name, **spec_kw)
    m = C(**ctor_kw)
    m.load_state_dict(sd)
    m.compute_dtype = torch.float32
    m = m.to(DEV)
    K = spec_kw["out_channels"]
    x, mask = otrain.closed_form_input(2, 32, c=cin)
    y = torch.cat([mask.roll(3 * k, dims=3) for k in range(K)], 1)
    okw = {} if name == "AttentionUNet" else {"t": t}
    m.eval()
    with torch.no_grad():
        ev = m(x.to(DEV)).cpu()
        ref = nets.NETS[name]({k: v.clone() for k, v in sd.items()}, x, False, **okw)
    assert ev.shape == ref.shape == (2, K, 32, 32)
    assert _rel(ev.numpy(), ref.numpy()) < 1e-4
    m.train()
    out = m(x.to(DEV))
    loss = mnn.BCEWithLogitsLoss()(out, y.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    # gradients against the fp64 oracle evaluated on the GPU's own ReLU / max-pool decisions (tests/test_gpu_kinks.py): sharp
    from gpu_util import gpu_kinks, replayed_oracle
    relu, pool = gpu_kinks(out._mi355_plan)
    l64, o64, g64 = replayed_oracle(name, sd, x, y, relu, pool, **okw)
    assert _rel(out.detach().cpu().numpy(), o64.numpy()) < 1e-4
    assert abs(float(loss.detach()) - l64) < 1e-5
    params = dict(m.named_parameters())
    gmax = max(float(v.abs().max()) for v in g64.values())
    errs = []
    for k, refg in g64.items():
        sc = float(refg.abs().max())
        if sc >= 1e-6 * gmax:
            errs.append(float((params[k].grad.cpu().double() - refg).abs().max()) / sc)
    assert np.median(errs) <= 5e-5 and np.max(errs) <= 1e-3, (np.median(errs), np.max(errs))
    head = "out" if name == "AttentionUNet" else "conv_1x1"
    for k in (head + ".weight", head + ".bias"):
        assert _rel(params[k].grad.cpu().numpy(), g64[k].numpy()) < 1e-4, k

@pytest.mark.parametrize("name", ["resnet18_tv", "resnet50_tv"])
def test_torchvision_layout_resnets_match_oracle_fp32(name):
    """The hub layouts of helpers.py:158-161 (torchvision resnet18 / resnet50; parity unpinned — torchvision is absent, the
    oracle restates the public architecture): logits, loss and every gradient tensor against the fp64 oracle."""
    from mi355 import nn as mnn
    from utils.helpers import get_class_model
    sd = nets.closed_form_state(name, num_classes=3, head_dropout=True)
    m, head = get_class_model(name)
    assert head == "fc"
    m.fc[0].p = 0.0
    m.load_state_dict(sd)
    m.compute_dtype = torch.float32
    m = m.to(DEV)
    x, y = otrain.synthetic_batch(8, 64, seed=3, classes=3)
    m.eval()
    with torch.no_grad():
        ev = m(x.to(DEV)).cpu()
        ev_ref = nets.NETS[name]({k: v.clone() for k, v in sd.items()}, x, False)
    assert _rel(ev.numpy(), ev_ref.numpy()) < 1e-4
    m.train()
    out = m(x.to(DEV))
    loss = mnn.CrossEntropyLoss(label_smoothing=0.1)(out, y.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    from gpu_util import gpu_kinks, replayed_oracle
    relu, pool = gpu_kinks(out._mi355_plan)
    l64, o64, g64 = replayed_oracle(name, sd, x, y, relu, pool, seg=False)
    assert _rel(out.detach().cpu().numpy(), o64.numpy()) < 1e-4
    assert abs(float(loss.detach()) - l64) < 1e-4 * max(1.0, abs(l64))
    params = dict(m.named_parameters())
    gmax = max(float(v.abs().max()) for v in g64.values())
    errs = {}
    for k, ref in g64.items():
        sc = float(ref.abs().max())
        if sc >= 1e-6 * gmax:
            errs[k] = float((params[k].grad.cpu().double() - ref).abs().max()) / sc
    e = np.array(list(errs.values()))
    assert np.median(e) <= 2e-4 and e.max() <= 2e-3, (np.median(e), max(errs, key=errs.get), e.max())
    msd = m.state_dict()
    _, _, _ = otrain.forward_backward(name, sd, x, y, False)          # (updates the oracle's BN buffers in place)
    for k, v in sd.items():
        if k.endswith(("running_mean", "running_var")):
            assert _rel(msd[k].cpu().numpy(), v.numpy()) < RTOL, k
