"""-m gpu: shapes off the benchmark path, fp32 HIP path vs the oracle on the same weights:
non-square images, sizes that no tile shape divides (the generic kernels and zero-page padding paths), batch 1,
odd extents through strided convolutions and padded max-pooling, and the drop-in train() on a loader whose last
batch is ragged (a second launch plan for the smaller batch, sample-weighted epoch means as helpers.py:337-343)."""
import re

import numpy as np
import pytest
import torch

from oracle import nets
from oracle import train as otrain

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


def _he(sd):
    for v in sd.values():
        if v.dim() == 4:
            v.mul_(6 ** 0.5)          # He gain: activations stay O(1) through eval-mode (identity) BatchNorm
    return sd


@pytest.mark.parametrize("shape", [(1, 3, 48, 80), (3, 3, 16, 16), (2, 3, 112, 32)])
def test_attention_unet_odd_shapes(shape):
    from mi355 import nn as mnn
    from models.segmentation_models.AttentionUNet import AttentionUNet
    sd = _he(nets.default_init_state("AttentionUNet", seed=11))
    m = AttentionUNet()
    m.load_state_dict(sd)
    m.compute_dtype = torch.float32
    m = m.to(DEV)
    g = torch.Generator().manual_seed(shape[2])
    x = torch.randn(*shape, generator=g)
    y = (torch.rand(shape[0], 1, shape[2], shape[3], generator=g) > 0.6).float()
    m.eval()
    with torch.no_grad():
        ev = m(x.to(DEV)).cpu()
    assert _rel(ev, nets.attention_unet({k: v.clone() for k, v in sd.items()}, x, False)) < 1e-3
    if shape[0] * shape[2] * shape[3] // 256 > 1:      # train-mode BN needs more than one value per channel at the 1/16 level
        m.train()
        out = m(x.to(DEV))
        loss = mnn.BCEWithLogitsLoss()(out, y.to(DEV))
        loss.backward()
        l_ref, o_ref, _ = otrain.forward_backward("AttentionUNet", {k: v.clone() for k, v in sd.items()}, x, y, True)
        assert _rel(out.detach().cpu(), o_ref) < 2e-3 and abs(float(loss.detach()) - l_ref) < 1e-3


@pytest.mark.parametrize("name,shape", [("ResNet18", (3, 3, 70, 90)), ("ResNet18", (1, 3, 33, 47)), ("VGG16", (2, 3, 40, 72)),
                                        ("ResNet50", (2, 3, 65, 65))])
def test_classifiers_odd_shapes(name, shape):
    from models.classification_models import ResNet, VGG
    sd = _he(nets.default_init_state(name, seed=5, num_classes=3))
    m = getattr(ResNet if name.startswith("ResNet") else VGG, name)(num_classes=3)
    m.load_state_dict(sd)
    m.compute_dtype = torch.float32
    m = m.to(DEV).eval()
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(shape[3]))
    with torch.no_grad():
        ev = m(x.to(DEV)).cpu()
    assert _rel(ev, nets.NETS[name]({k: v.clone() for k, v in sd.items()}, x, False)) < 1e-3


def test_train_loop_with_ragged_last_batch(tmp_path, capsys):
    from torch.utils.data import DataLoader, TensorDataset
    from models.segmentation_models.AttentionUNet import AttentionUNet
    from utils import helpers
    hw, epochs, lr = 64, 2, 1e-4
    xs, ys = otrain.synthetic_batch(7, hw, seed=21)          # 7 samples, batch 3 -> batches of 3, 3, 1
    xv, yv = otrain.synthetic_batch(5, hw, seed=22)          # validation: 3 + 2
    sd0 = nets.default_init_state("AttentionUNet", seed=2)
    m = AttentionUNet()
    m.load_state_dict(sd0)
    m.compute_dtype = torch.float32
    tr = DataLoader(TensorDataset(xs, ys), batch_size=3, shuffle=False)
    va = DataLoader(TensorDataset(xv, yv), batch_size=3, shuffle=False)
    best = helpers.train(m, tr, va, torch.device(DEV), epochs, lr, "AttentionUNet", str(tmp_path), seg=True)
    rows = re.findall(r"Ep(\d+): TrainLoss ([\d.]+) \| ValLoss ([\d.]+) \| IoU ([\d.]+)", capsys.readouterr().out)
    sd = {k: v.clone() for k, v in sd0.items()}
    ref_best, hist = otrain.train_seg("AttentionUNet", sd, [(xs[i:i + 3], ys[i:i + 3]) for i in (0, 3, 6)],
                                      [(xv[:3], yv[:3]), (xv[3:], yv[3:])], epochs, lr)
    assert len(rows) == epochs == len(hist)
    for row, (tl, vl, iou) in zip(rows, hist):
        # (printed with three decimals; the batch of ONE sample leaves 16 values per channel to the deepest train-mode BNs)
        assert abs(float(row[1]) - tl) <= 3e-3 and abs(float(row[2]) - vl) <= 3e-3 and abs(float(row[3]) - iou) <= 5e-3
    assert abs(best - ref_best) <= 5e-3 * ref_best
    assert len(m.engine.plans) >= 4            # train / eval plans for both batch sizes of each loader
