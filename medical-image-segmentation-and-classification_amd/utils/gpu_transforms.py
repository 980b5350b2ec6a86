"""The reference's per-sample Albumentations pipelines (utils/trainer.py:52-115) as batched GPU transforms over uint8
images that already sit in HBM: ``A.Resize`` -> ``A.ShiftScaleRotate`` -> ``A.HorizontalFlip`` ->
``A.RandomBrightnessContrast`` -> ``A.Normalize`` -> ``ToTensorV2`` (masks: nearest sampling, ``/ 255``,
utils/dataset.py:120-126).  Four CPU DataLoader workers running those per image cannot feed a GPU that trains at
1500 images/s; here a 32-image batch costs a few launches over 6 MB.

The random draws (angle, scale, shift, flip, brightness / contrast — same ranges and probabilities as the reference)
come from a ``torch.Generator`` on the host and travel as a [N, 6] matrix and a [N, 2] vector."""
from __future__ import annotations

import math

import torch

from mi355.lib import lib

IMAGENET_MEAN = (0.485, 0.456, 0.406)          # trainer.py:48-49
IMAGENET_STD = (0.229, 0.224, 0.225)


def resize_matrix(hs, ws, h, w):
    fx, fy = ws / w, hs / h
    return [fx, 0.0, 0.5 * fx - 0.5, 0.0, fy, 0.5 * fy - 0.5]


def shift_scale_rotate_matrix(h, w, angle_deg, scale, dx, dy, hflip=False):
    """dst -> src map of A.ShiftScaleRotate followed by A.HorizontalFlip (see oracle/transforms.py for the convention)."""
    cx, cy = w / 2 - 0.5, h / 2 - 0.5
    a = math.radians(angle_deg)
    al, be = scale * math.cos(a), scale * math.sin(a)
    m = torch.tensor([[al, be, (1 - al) * cx - be * cy + dx * w], [-be, al, be * cx + (1 - al) * cy + dy * h], [0, 0, 1]],
                     dtype=torch.float64)
    if hflip:
        m = torch.tensor([[-1, 0, w - 1], [0, 1, 0], [0, 0, 1]], dtype=torch.float64) @ m
    return torch.linalg.inv(m)[:2].reshape(-1).tolist()


class SegBatchTransform:
    """train_seg_transform / val_seg_transform (trainer.py:83-112) for a batch: ``(images uint8 [N,Hs,Ws,3], masks uint8
    [N,Hs,Ws]) -> (x float32 [N,3,S,S] normalised, y float32 [N,1,S,S] in {0,1})`` on the GPU."""

    def __init__(self, size=256, train=False, seed=0, device="cuda"):
        self.size, self.train, self.device = size, train, torch.device(device)
        self.gen = torch.Generator().manual_seed(seed)
        self.mean = torch.tensor(IMAGENET_MEAN, device=self.device)
        self.std = torch.tensor(IMAGENET_STD, device=self.device)

    def _u(self, lo, hi):
        return float(torch.rand((), generator=self.gen)) * (hi - lo) + lo

    def draw(self, n):
        """per-sample parameters with the reference's ranges: ShiftScaleRotate(0.05, 0.05, 15, p=0.7), HorizontalFlip(0.5),
        RandomBrightnessContrast(0.1, 0.1, p=0.5)"""
        mats, bcs = [], []
        for _ in range(n):
            ssr = self._u(0, 1) < 0.7
            angle, scale = (self._u(-15, 15), 1 + self._u(-0.05, 0.05)) if ssr else (0.0, 1.0)
            dx, dy = (self._u(-0.05, 0.05), self._u(-0.05, 0.05)) if ssr else (0.0, 0.0)
            flip = self._u(0, 1) < 0.5
            mats.append(shift_scale_rotate_matrix(self.size, self.size, angle, scale, dx, dy, flip))
            bcs.append([1 + self._u(-0.1, 0.1), self._u(-0.1, 0.1)] if self._u(0, 1) < 0.5 else [1.0, 0.0])
        return mats, bcs

    def _to_square(self, images, masks, n, hs, ws):
        """A.Resize(size, size) (trainer.py:85-87): aspect ratio not kept."""
        s = self.size
        m0 = torch.tensor([resize_matrix(hs, ws, s, s)] * n, dtype=torch.float32, device=self.device)
        img = torch.empty(n, s, s, 3, dtype=torch.uint8, device=self.device)
        lib.mi355_warp_u8(images, n, hs, ws, 3, m0, img, s, s, 0, 0)
        msk = None
        if masks is not None:
            hm, wm = masks.shape[1], masks.shape[2]         # (the dataset's masks are 256x256 next to 299x299 images: A.Resize maps each)
            masks = masks.to(self.device).contiguous().view(n, hm, wm, 1)
            mm = m0 if (hm, wm) == (hs, ws) else torch.tensor([resize_matrix(hm, wm, s, s)] * n, dtype=torch.float32, device=self.device)
            msk = torch.empty(n, s, s, 1, dtype=torch.uint8, device=self.device)
            lib.mi355_warp_u8(masks, n, hm, wm, 1, mm, msk, s, s, 1, 0)
        return img, msk

    @torch.no_grad()
    def __call__(self, images, masks=None, params=None):
        images = images.to(self.device).contiguous()
        n, hs, ws, c = images.shape
        assert images.dtype == torch.uint8 and c == 3
        s = self.size
        img, msk = self._to_square(images, masks, n, hs, ws)
        bc = None
        if self.train:
            mats, bcs = params if params is not None else self.draw(n)
            m1 = torch.tensor(mats, dtype=torch.float32, device=self.device)
            bc = torch.tensor(bcs, dtype=torch.float32, device=self.device)
            img2 = torch.empty_like(img)
            lib.mi355_warp_u8(img, n, s, s, 3, m1, img2, s, s, 0, 1)
            img = img2
            if msk is not None:
                msk2 = torch.empty_like(msk)
                lib.mi355_warp_u8(msk, n, s, s, 1, m1, msk2, s, s, 1, 1)
                msk = msk2
        x = torch.empty(n, 3, s, s, dtype=torch.float32, device=self.device)
        lib.mi355_normalize_u8(img, n, s, s, 3, bc, self.mean, self.std, x)
        if msk is None:
            return x
        y = torch.empty(n, 1, s, s, dtype=torch.float32, device=self.device)
        lib.mi355_normalize_u8(msk, n, s, s, 1, None, None, None, y)
        return x, y


def _py3round(v):
    """Albumentations' rounding of the resized extent (round half to even)."""
    return int(round(v))


def longest_max_size_shape(hs, ws, size):
    sc = size / max(hs, ws)
    return _py3round(hs * sc), _py3round(ws * sc)


class ClsBatchTransform(SegBatchTransform):
    """train_cls_transform / val_cls_transform (trainer.py:52-82) for a batch of equally sized images:
    ``A.LongestMaxSize(size)`` (aspect ratio kept, bilinear) -> ``A.PadIfNeeded(size, size, BORDER_CONSTANT, 0)`` (centred:
    the odd pixel goes to the bottom / right) -> the same augmentation / normalisation tail as the segmentation pipeline.
    ``images uint8 [N,Hs,Ws,3] -> x float32 [N,3,S,S]``."""

    def _to_square(self, images, masks, n, hs, ws):
        assert masks is None, "classification samples carry no mask"
        s = self.size
        h1, w1 = longest_max_size_shape(hs, ws, s)
        m0 = torch.tensor([resize_matrix(hs, ws, h1, w1)] * n, dtype=torch.float32, device=self.device)
        if (h1, w1) == (s, s):
            img = torch.empty(n, s, s, 3, dtype=torch.uint8, device=self.device)
            lib.mi355_warp_u8(images, n, hs, ws, 3, m0, img, s, s, 0, 0)
            return img, None
        small = torch.empty(n, h1, w1, 3, dtype=torch.uint8, device=self.device)
        lib.mi355_warp_u8(images, n, hs, ws, 3, m0, small, h1, w1, 0, 0)
        img = torch.zeros(n, s, s, 3, dtype=torch.uint8, device=self.device)
        top, left = (s - h1) // 2, (s - w1) // 2
        img[:, top:top + h1, left:left + w1] = small
        return img, None
