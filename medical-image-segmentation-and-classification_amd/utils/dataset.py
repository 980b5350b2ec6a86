"""Datasets of the MI355X path — same class names, constructor arguments and sample lists as the reference's
utils/dataset.py (``ClassificationDataset`` :24-67, ``SegmentationDataset`` :70-137: ``dataset/splits/{split}.csv`` rows
``id, class`` -> ``{root}/{class}/images/{id}.png`` [+ ``masks/{id}.png``], files that do not exist are skipped), feeding the GPU
input pipeline instead of per-sample Albumentations in DataLoader workers:

    ds = SegmentationDataset("dataset", SegBatchTransform(256, train=True), split="train")
    for x, y in GpuBatchLoader(ds, batch_size=8, shuffle=True):      # x [B,3,256,256] float32, y [B,1,256,256] on the GPU
        ...                                                             # exactly what helpers.train() consumes

PNG files are decoded by native threads (csrc/png_decode.cpp: PIL's ``Image.open(p).convert("RGB" / "L")`` semantics, pinned
bit-exactly by tests/test_png_decode.py) into ONE pinned uint8 batch, copied to HBM, and resized / augmented / normalised
there (utils/gpu_transforms.py).  ``dataset[i]`` still works sample by sample (a batch of one) for code that indexes."""
from __future__ import annotations

import csv
import ctypes
import os

import numpy as np
import torch

from mi355.lib import lib

CLASSES = ["COVID", "Healthy", "Non-COVID"]          # utils/pipeline.py:22, imported by the reference's dataset.py:16


def _rows(root, split):
    path = os.path.join(root, "splits", f"{split}.csv")
    if not os.path.exists(path):
        raise FileNotFoundError(f"Split file not found: {path}")
    with open(path, newline="") as f:
        return [(r["id"], r["class"]) for r in csv.DictReader(f)]


def read_files(paths):
    out = []
    for p in paths:
        with open(p, "rb") as f:
            out.append(np.frombuffer(f.read(), dtype=np.uint8))
    return out


def png_size(buf):
    w, h = ctypes.c_int(), ctypes.c_int()
    rc = lib.raw("mi355_png_info")(buf.ctypes.data, buf.size, ctypes.byref(w), ctypes.byref(h), None, None)
    if rc:
        raise RuntimeError(f"PNG header: {lib.raw('mi355_last_error')().decode()}")
    return h.value, w.value


def decode_batch(bufs, channels, threads=8, pin=True, names=None):
    """PNG byte buffers of ONE size -> uint8 tensor [N, H, W, channels] in (pinned) host memory.  A buffer of another size is
    reported by name (``names[i]``, else its index) before anything is decoded."""
    n = len(bufs)
    h, w = png_size(bufs[0])
    for i in range(1, n):
        hi, wi = png_size(bufs[i])
        if (hi, wi) != (h, w):
            who = names[i] if names is not None else f"buffer {i}"
            raise ValueError(f"decode_batch: {who} is {wi} x {hi}, the batch started with {w} x {h} ({names[0] if names is not None else 'buffer 0'}); "
                             "decode files of one size per call")
    out = torch.empty((n, h, w, channels), dtype=torch.uint8, pin_memory=pin and torch.cuda.is_available())
    ptrs = (ctypes.c_void_p * n)(*[b.ctypes.data for b in bufs])
    sizes = (ctypes.c_longlong * n)(*[b.size for b in bufs])
    rc = lib.raw("mi355_png_decode_batch")(ptrs, sizes, n, channels, out.data_ptr(), h * w * channels, w, h, threads)
    if rc:
        raise RuntimeError(f"PNG decode: {lib.raw('mi355_last_error')().decode()}")
    return out


class ClassificationDataset:
    is_seg = False

    def __init__(self, root, transform, split="train"):
        self.root, self.transform = root, transform
        self.samples = []
        for img_id, cls in _rows(root, split):
            p = os.path.join(root, cls, "images", f"{img_id}.png")
            if os.path.exists(p):
                self.samples.append((p, CLASSES.index(cls)))

    def __len__(self):
        return len(self.samples)

    def load_batch(self, idxs, threads=8):
        paths = [self.samples[i][0] for i in idxs]
        labels = torch.tensor([self.samples[i][1] for i in idxs], dtype=torch.int64)
        return decode_batch(read_files(paths), 3, threads, names=paths), labels

    def __getitem__(self, idx):
        img, label = self.load_batch([idx], 1)
        if self.transform is None:
            return img[0].permute(2, 0, 1), int(label[0])                 # ToTensorV2 of the raw image (dataset.py:62)
        return self.transform(img)[0], int(label[0])


class SegmentationDataset:
    is_seg = True

    def __init__(self, root, transform, split="train"):
        self.root, self.transform = root, transform
        self.pairs = []
        for img_id, cls in _rows(root, split):
            ip = os.path.join(root, cls, "images", f"{img_id}.png")
            mp = os.path.join(root, cls, "masks", f"{img_id}.png")
            if os.path.exists(ip) and os.path.exists(mp):
                self.pairs.append((ip, mp))

    def __len__(self):
        return len(self.pairs)

    def load_batch(self, idxs, threads=8):
        ip, mp = [self.pairs[i][0] for i in idxs], [self.pairs[i][1] for i in idxs]
        imgs = decode_batch(read_files(ip), 3, threads, names=ip)
        masks = decode_batch(read_files(mp), 1, threads, names=mp)
        return imgs, masks[..., 0]

    def __getitem__(self, idx):
        img, mask = self.load_batch([idx], 1)
        if self.transform is None:
            return img[0].permute(2, 0, 1), mask[0][None].float() / 255.0   # dataset.py:129-133
        x, y = self.transform(img, mask)
        return x[0], y[0]


class Subset:
    """Index subset of one of the datasets above (the reference splits with torch.utils.data.Subset, trainer.py:131-151)."""

    def __init__(self, dataset, indices):
        self.base, self.indices = dataset, [int(i) for i in indices]
        self.transform, self.is_seg = dataset.transform, dataset.is_seg

    def __len__(self):
        return len(self.indices)

    def load_batch(self, idxs, threads=8):
        return self.base.load_batch([self.indices[i] for i in idxs], threads)

    def __getitem__(self, idx):
        return self.base[self.indices[idx]]


class GpuBatchLoader:
    """DataLoader stand-in for the two datasets above: yields device batches; ``len(loader)`` / ``loader.dataset`` as
    helpers.train() uses them (helpers.py:317-342, 365)."""

    def __init__(self, dataset, batch_size, shuffle=False, drop_last=False, seed=0, threads=8, device="cuda", indices=None):
        """``indices``: iterate this subset only (torch.utils.data.Subset semantics, trainer.py:131-151); ``loader.dataset`` then
        has the subset's length, which is what helpers.train() divides the epoch sums by."""
        self.batch_size, self.shuffle, self.drop_last = int(batch_size), shuffle, drop_last
        self.gen = torch.Generator().manual_seed(seed)
        self.threads, self.device = threads, torch.device(device)
        self.dataset = dataset if indices is None else Subset(dataset, indices)

    def __len__(self):
        n = len(self.dataset)
        return n // self.batch_size if self.drop_last else -(-n // self.batch_size)

    def epoch_batches(self):
        """The index batches of one epoch, in order (draws this epoch's permutation when shuffling)."""
        n = len(self.dataset)
        order = torch.randperm(n, generator=self.gen).tolist() if self.shuffle else list(range(n))
        return [order[b * self.batch_size:(b + 1) * self.batch_size] for b in range(len(self))]

    def load(self, idxs):
        """One device batch (x, y) of the samples ``idxs``: native PNG decode into pinned memory, transforms on the GPU."""
        tf = self.dataset.transform
        if self.dataset.is_seg:
            imgs, masks = self.dataset.load_batch(idxs, self.threads)
            return tf(imgs.to(self.device, non_blocking=True), masks.to(self.device, non_blocking=True))
        imgs, labels = self.dataset.load_batch(idxs, self.threads)
        return tf(imgs.to(self.device, non_blocking=True)), labels.to(self.device, non_blocking=True)

    def __iter__(self):
        for idxs in self.epoch_batches():
            yield self.load(idxs)
