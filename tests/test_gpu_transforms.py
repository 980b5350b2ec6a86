"""-m gpu: the batched GPU input transforms (utils/gpu_transforms.py, csrc/input_pipeline.hip) against the per-sample
numpy restatement of the reference's Albumentations pipelines (oracle/transforms.py).  Bound: identical uint8 images
except for <= 0.5 % of the pixels by one grey level (float rounding of a value that sits on .5), i.e. 1/255/std after
normalisation; masks identical."""
import numpy as np
import pytest
import torch

from oracle import transforms as ot

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _batch(n, hs, ws, seed):
    g = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:hs, 0:ws]
    imgs, masks = [], []
    for i in range(n):      # smooth structure + noise, so that interpolation matters; one ellipse mask per image
        base = 127 + 90 * np.sin(xx / (7.0 + i)) * np.cos(yy / (5.0 + 2 * i))
        img = np.clip(base[..., None] + g.normal(0, 12, (hs, ws, 3)), 0, 255).astype(np.uint8)
        mask = ((((xx - ws * 0.5) / (ws * 0.3)) ** 2 + ((yy - hs * 0.45) / (hs * 0.25)) ** 2) < 1).astype(np.uint8) * 255
        imgs.append(img); masks.append(mask)
    return np.stack(imgs), np.stack(masks)


def _close(x, ref, tol_levels=1):
    d = np.abs(x - ref)
    lim = tol_levels / 255 / ot.IMAGENET_STD.min() * 1.001
    assert d.max() <= lim, d.max()
    assert (d > 1e-6).mean() <= 5e-3, (d > 1e-6).mean()


@pytest.mark.parametrize("src", [(299, 299), (256, 256), (180, 333)])
def test_val_transform_matches_oracle(src):
    from utils.gpu_transforms import SegBatchTransform
    imgs, masks = _batch(3, src[0], src[1], 1)
    x, y = SegBatchTransform(256, train=False, device=DEV)(torch.from_numpy(imgs), torch.from_numpy(masks))
    for i in range(3):
        xr, yr = ot.val_seg_sample(imgs[i], masks[i][..., None], 256)
        _close(x[i].cpu().numpy(), xr)
        assert np.array_equal(y[i].cpu().numpy(), yr)
    assert set(np.unique(y.cpu().numpy())) <= {0.0, 1.0}


def test_train_transform_matches_oracle():
    from utils.gpu_transforms import SegBatchTransform, shift_scale_rotate_matrix
    imgs, masks = _batch(4, 299, 299, 2)
    t = SegBatchTransform(256, train=True, seed=5, device=DEV)
    draws = [(12.5, 1.04, 0.03, -0.05, True, 1.08, -0.06), (-15.0, 0.95, -0.05, 0.05, False, 0.9, 0.1),
             (0.0, 1.0, 0.0, 0.0, True, 1.0, 0.0), (7.0, 1.0, 0.02, 0.0, False, 1.1, 0.1)]
    mats = [shift_scale_rotate_matrix(256, 256, a, s, dx, dy, f) for a, s, dx, dy, f, _, _ in draws]
    bcs = [[al, be] for *_, al, be in draws]
    x, y = t(torch.from_numpy(imgs), torch.from_numpy(masks), params=(mats, bcs))
    for i, (a, s, dx, dy, f, al, be) in enumerate(draws):
        xr, yr = ot.train_seg_sample(imgs[i], masks[i][..., None], a, s, dx, dy, f, al, be, 256)
        _close(x[i].cpu().numpy(), xr, tol_levels=2)       # two uint8 roundings in sequence
        assert (y[i].cpu().numpy() != yr).mean() <= 1e-3   # nearest sampling exactly on a pixel boundary
    # the random path: reference ranges, reproducible from the seed, masks stay binary
    x1, y1 = SegBatchTransform(256, train=True, seed=9, device=DEV)(torch.from_numpy(imgs), torch.from_numpy(masks))
    x2, y2 = SegBatchTransform(256, train=True, seed=9, device=DEV)(torch.from_numpy(imgs), torch.from_numpy(masks))
    assert torch.equal(x1, x2) and torch.equal(y1, y2) and set(np.unique(y1.cpu().numpy())) <= {0.0, 1.0}
    assert tuple(x1.shape) == (4, 3, 256, 256) and tuple(y1.shape) == (4, 1, 256, 256)


@pytest.mark.parametrize("src", [(299, 299), (180, 333), (400, 250), (256, 256), (101, 77)])
def test_cls_transforms_match_oracle(src):
    """ClsBatchTransform: LongestMaxSize + zero PadIfNeeded (trainer.py:52-82) in front of the shared augmentation tail."""
    from utils.gpu_transforms import ClsBatchTransform, shift_scale_rotate_matrix
    imgs, _ = _batch(3, src[0], src[1], 7)
    x = ClsBatchTransform(256, train=False, device=DEV)(torch.from_numpy(imgs))
    assert tuple(x.shape) == (3, 3, 256, 256)
    for i in range(3):
        _close(x[i].cpu().numpy(), ot.val_cls_sample(imgs[i], 256))
    draws = [(10.0, 1.03, 0.02, -0.04, True, 1.05, -0.03), (-14.0, 0.96, -0.05, 0.05, False, 0.92, 0.08), (0.0, 1.0, 0.0, 0.0, False, 1.0, 0.0)]
    mats = [shift_scale_rotate_matrix(256, 256, a, s, dx, dy, f) for a, s, dx, dy, f, _, _ in draws]
    x = ClsBatchTransform(256, train=True, seed=1, device=DEV)(torch.from_numpy(imgs), params=(mats, [[al, be] for *_, al, be in draws]))
    for i, d in enumerate(draws):
        _close(x[i].cpu().numpy(), ot.train_cls_sample(imgs[i], *d, 256), tol_levels=2)
    if src[0] != src[1]:        # the padded band is exactly normalised black in the un-augmented output
        v = ClsBatchTransform(256, train=False, device=DEV)(torch.from_numpy(imgs))[0].cpu().numpy()
        black = (0 - ot.IMAGENET_MEAN) / ot.IMAGENET_STD
        edge = v[:, 0, :] if src[0] < src[1] else v[:, :, 0]
        assert np.allclose(edge, black[:, None], atol=1e-6)


def test_gpu_batch_loader_feeds_train_ready_batches(tmp_path):
    """utils/dataset.py::GpuBatchLoader: PNG files -> native threaded decode -> pinned batch -> GPU transforms; the batches equal
    the oracle's per-sample pipeline applied to PIL-decoded arrays (dataset.py:100-126 + trainer.py:98-112), images 299x299 with
    256x256 masks as in the COVID-19 Radiography files."""
    Image = pytest.importorskip("PIL.Image")
    from test_dataset_cpu import make_tree
    from utils.dataset import GpuBatchLoader, SegmentationDataset, ClassificationDataset
    from utils.gpu_transforms import SegBatchTransform, ClsBatchTransform
    root = str(tmp_path / "dataset")
    make_tree(root, n=7)
    ds = SegmentationDataset(root, SegBatchTransform(256, train=False, device=DEV), "train")
    dl = GpuBatchLoader(ds, batch_size=4, shuffle=False, device=DEV)
    assert len(dl) == 2 and len(dl.dataset) == 6
    seen = 0
    for x, y in dl:
        assert x.is_cuda and x.shape[1:] == (3, 256, 256) and y.shape[1:] == (1, 256, 256)
        for j in range(x.shape[0]):
            ip, mp = ds.pairs[seen + j]
            img = np.array(Image.open(ip).convert("RGB")); mask = np.array(Image.open(mp).convert("L"))
            xr, yr = ot.val_seg_sample(img, mask[..., None], 256)
            _close(x[j].cpu().numpy(), xr)
            assert np.array_equal(y[j].cpu().numpy(), yr)
        seen += x.shape[0]
    assert seen == 6
    cds = ClassificationDataset(root, ClsBatchTransform(256, train=False, device=DEV), "train")
    got = [(x.shape, lab.tolist()) for x, lab in GpuBatchLoader(cds, batch_size=3, shuffle=True, seed=1, device=DEV)]
    assert sum(len(l) for _, l in got) == 7 and all(s[1:] == (3, 256, 256) for s, _ in got)


def test_train_runs_from_png_files_end_to_end(tmp_path, capsys):
    """The whole input side in the reference's shape (trainer.py:119-160 -> helpers.train): PNG files on disk, two dataset objects
    with the train / val transforms over one index split, GpuBatchLoader in DataLoader's place, the drop-in train() on top."""
    pytest.importorskip("PIL.Image")
    import os
    from test_dataset_cpu import make_tree
    from utils.dataset import GpuBatchLoader, SegmentationDataset
    from utils.gpu_transforms import SegBatchTransform
    from utils.helpers import get_seg_model, train
    root = str(tmp_path / "dataset")
    make_tree(root, n=13)
    ds_tr = SegmentationDataset(root, SegBatchTransform(64, train=True, device=DEV), "train")
    ds_va = SegmentationDataset(root, SegBatchTransform(64, train=False, device=DEV), "train")
    assert len(ds_tr) == 12
    perm = torch.randperm(12, generator=torch.Generator().manual_seed(0)).tolist()
    tr = GpuBatchLoader(ds_tr, 4, shuffle=True, device=DEV, indices=perm[:9])
    va = GpuBatchLoader(ds_va, 4, shuffle=False, device=DEV, indices=perm[9:])
    assert len(tr.dataset) == 9 and len(tr) == 3 and len(va.dataset) == 3 and len(va) == 1
    best = train(get_seg_model("attentionunet"), tr, va, torch.device(DEV), 2, 1e-3, "AttentionUNet", str(tmp_path / "w"), seg=True)
    out = capsys.readouterr().out
    assert np.isfinite(best) and "Ep2" in out and os.path.exists(tmp_path / "w" / "AttentionUNet_best_loss.pt")
