"""CPU: utils/dataset.py (drop-in for the reference's ClassificationDataset / SegmentationDataset, dataset.py:24-137): sample
lists from the split CSVs with the reference's skipping rule, and native batch decoding equal to PIL byte for byte."""
import os

import numpy as np
import pytest
import torch

from mi355.lib import available

pytestmark = pytest.mark.skipif(not available(), reason="libmi355conv.so not built")
Image = pytest.importorskip("PIL.Image")


def make_tree(root, n=7, seed=0, splits=("train",)):
    g = np.random.RandomState(seed)
    rows = []
    for i in range(n):
        cls = ["COVID", "Healthy", "Non-COVID"][i % 3]
        for sub in ("images", "masks"):
            os.makedirs(os.path.join(root, cls, sub), exist_ok=True)
        img = g.randint(0, 256, (299, 299)).astype(np.uint8)
        Image.fromarray(img, "L").save(os.path.join(root, cls, "images", f"{cls}-{i}.png"))
        if i != 4:                                          # sample 4 has no mask: the segmentation dataset skips it
            yy, xx = np.mgrid[0:256, 0:256]
            m = (((yy - 100 - 3 * i) ** 2 + (xx - 128) ** 2) < 60 ** 2).astype(np.uint8) * 255
            Image.fromarray(m, "L").save(os.path.join(root, cls, "masks", f"{cls}-{i}.png"))
        rows.append((f"{cls}-{i}", cls))
    rows.append(("COVID-missing", "COVID"))                 # listed in the CSV, absent on disk: skipped by both
    os.makedirs(os.path.join(root, "splits"), exist_ok=True)
    for sp in splits:
        with open(os.path.join(root, "splits", f"{sp}.csv"), "w") as f:
            f.write("id,class\n" + "".join(f"{a},{b}\n" for a, b in rows))
    return rows


def test_sample_lists_and_native_decode_match_pil(tmp_path):
    from utils.dataset import CLASSES, ClassificationDataset, SegmentationDataset
    root = str(tmp_path / "dataset")
    make_tree(root)
    cls = ClassificationDataset(root, None, "train")
    seg = SegmentationDataset(root, None, "train")
    assert len(cls) == 7 and len(seg) == 6
    assert [lab for _, lab in cls.samples] == [CLASSES.index(["COVID", "Healthy", "Non-COVID"][i % 3]) for i in range(7)]
    with pytest.raises(FileNotFoundError, match="Split file not found"):
        ClassificationDataset(root, None, "val")
    imgs, labels = cls.load_batch([0, 3, 5], threads=2)
    assert imgs.shape == (3, 299, 299, 3) and imgs.dtype == torch.uint8 and labels.tolist() == [0, 0, 2]
    for j, i in enumerate([0, 3, 5]):
        assert np.array_equal(imgs[j].numpy(), np.array(Image.open(cls.samples[i][0]).convert("RGB")))
    im, mk = seg.load_batch([1, 4], threads=2)
    assert im.shape == (2, 299, 299, 3) and mk.shape == (2, 256, 256)
    for j, i in enumerate([1, 4]):
        assert np.array_equal(mk[j].numpy(), np.array(Image.open(seg.pairs[i][1]).convert("L")))
    x, y = seg[2]                                           # transform=None: the reference's ToTensorV2 fallback (dataset.py:129-133)
    assert x.shape == (3, 299, 299) and y.shape == (1, 256, 256) and float(y.max()) == 1.0 and y.dtype == torch.float32


def test_decode_batch_names_the_file_of_another_size(tmp_path):
    """decode_batch decodes files of ONE size into one buffer; a stray file of another size is refused BEFORE decoding, by path
    (it used to surface as 'image i failed' with no reason)."""
    from utils.dataset import decode_batch, read_files
    g = np.random.RandomState(3)
    paths = []
    for i, hw in enumerate([(40, 50), (40, 50), (41, 50)]):
        p = str(tmp_path / f"im{i}.png")
        Image.fromarray(g.randint(0, 256, hw).astype(np.uint8), "L").save(p)
        paths.append(p)
    assert decode_batch(read_files(paths[:2]), 3, 2, names=paths[:2]).shape == (2, 40, 50, 3)
    with pytest.raises(ValueError, match=r"im2\.png is 50 x 41, the batch started with 50 x 40"):
        decode_batch(read_files(paths), 3, 2, names=paths)
    with pytest.raises(ValueError, match="buffer 2 is 50 x 41"):
        decode_batch(read_files(paths), 1, 2)
