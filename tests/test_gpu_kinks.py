"""-m gpu: SHARP whole-model gradient parity of the fp32 HIP path.

The gradients of these ReLU / max-pool networks are discontinuous; any fp32 evaluation — the reference's own CPU path included —
differs from fp64 by ~1e-3 on every parameter gradient because a handful of the ~10^7 pre-activations land on the other side
of zero (tests/test_oracle_kinks.py measures it on the CPU: median 5.5e-3 on this fixture).  The other model tests therefore
anchor gradient errors on "as close to fp64 as the reference's fp32 path".  Here the discontinuity is taken out instead: the
ReLU masks and max-pool arg-max decisions the GPU ACTUALLY took (read back from the launch plan's activation buffers,
Plan.acts) are replayed inside the fp64 oracle, so both evaluate the same smooth function — and then every parameter gradient
of the whole network must agree to rounding accuracy: 3e-4 of the tensor's maximum (measured: median ~1e-5)."""
import numpy as np
import pytest
import torch
from gpu_util import gpu_kinks, replayed_oracle
from oracle import nets
from oracle import train as otrain

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("name,bs,hw", [("AttentionUNet", 2, 64), ("R2AttU_Net", 2, 32), ("R2U_Net", 2, 32), ("AttentionUNet", 4, 128)])
def test_fp32_gradients_match_fp64_oracle_on_the_same_masks(name, bs, hw):
    from mi355 import nn as mnn
    from utils.helpers import get_seg_model
    sd = nets.closed_form_state(name)
    m = get_seg_model({"AttentionUNet": "attentionunet", "R2AttU_Net": "r2attunet", "R2U_Net": "r2unet"}[name])
    m.load_state_dict(sd)
    m.compute_dtype = torch.float32
    m = m.to(DEV).train()
    x, y = otrain.closed_form_input(bs, hw)
    out = m(x.to(DEV))
    loss = mnn.BCEWithLogitsLoss()(out, y.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    relu, pool = gpu_kinks(out._mi355_plan)
    l64, o64, g64 = replayed_oracle(name, sd, x, y, relu, pool)
    assert len(pool) == 4
    assert float((out.detach().cpu().double() - o64).abs().max() / o64.abs().max()) < 1e-4
    assert abs(float(loss.detach()) - l64) < 1e-5
    gmax = max(float(v.abs().max()) for v in g64.values())
    errs = {}
    for k, p in m.named_parameters():
        ref = g64[k]
        sc = float(ref.abs().max())
        if sc < 1e-6 * gmax:
            assert float(p.grad.abs().max()) <= 1e-5 * gmax, k          # conv bias in front of a train-mode BN: exactly zero
            continue
        errs[k] = float((p.grad.cpu().double() - ref).abs().max()) / sc
    e = np.array(list(errs.values()))
    worst = max(errs, key=errs.get)
    assert np.median(e) <= 5e-5, np.median(e)
    assert e.max() <= 1e-3, (worst, errs[worst])
    assert np.mean(e <= 3e-4) >= 0.97, sorted(errs.items(), key=lambda kv: -kv[1])[:5]


@pytest.mark.parametrize("name,kw,bs,hw", [
    ("ResNet18", {"head_dropout": True}, 4, 64), ("ResNet50", {"head_dropout": True}, 8, 64),
    ("VGG16", {"head_dropout": True}, 2, 32), ("VGG16_BN", {"num_classes": 3, "head_dropout": True}, 2, 64),
])
def test_classifier_gradients_match_fp64_oracle_on_the_same_masks(name, kw, bs, hw):
    """The same statement for the classifier families (ResNet.py / VGG.py / the hub's vgg16_bn): 3x3-stride-2 and global max
    pools (the latter an arg-max over the whole feature map, ResNet.py:112) and the ReLUs behind the head's Linears are kinks too."""
    from mi355 import nn as mnn
    from utils.helpers import add_dropout_to_fc, get_class_model
    sd = nets.closed_form_state(name, **kw)
    m, head = get_class_model({"ResNet18": "resnet18", "ResNet50": "resnet50", "VGG16": "vgg16", "VGG16_BN": "vgg16_bn"}[name])
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    m.load_state_dict(sd)
    m.compute_dtype = torch.float32
    m = m.to(DEV).train()
    x, y = otrain.synthetic_batch(bs, hw, seed=3, classes=3)
    out = m(x.to(DEV))
    loss = mnn.CrossEntropyLoss(label_smoothing=0.1)(out, y.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    relu, pool = gpu_kinks(out._mi355_plan)
    l64, o64, g64 = replayed_oracle(name, sd, x, y, relu, pool, seg=False)
    assert float((out.detach().cpu().double() - o64).abs().max() / o64.abs().max()) < 1e-4
    assert abs(float(loss.detach()) - l64) < 1e-4 * max(1.0, abs(l64))
    gmax = max(float(v.abs().max()) for v in g64.values())
    errs = {}
    for k, p in m.named_parameters():
        ref = g64[k]
        sc = float(ref.abs().max())
        if sc >= 1e-6 * gmax:
            errs[k] = float((p.grad.cpu().double() - ref).abs().max()) / sc
    e = np.array(list(errs.values()))
    worst = max(errs, key=errs.get)
    # (ResNet50, 2048 channels over 8 x 2 x 2 samples at the last stage: the reference's CPU fp32 path itself is 1.4e-2 from fp64
    # on this fixture and 4.9e-5 with its own masks replayed)
    assert np.median(e) <= (2e-4 if name == "ResNet50" else 5e-5) and e.max() <= 1e-3, (np.median(e), worst, errs[worst])
