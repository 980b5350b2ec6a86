"""TEST INFRASTRUCTURE — numpy restatement of the reference's per-sample input transforms
(utils/trainer.py:52-115: A.LongestMaxSize / A.PadIfNeeded / A.Resize / A.ShiftScaleRotate / A.HorizontalFlip / A.RandomBrightnessContrast / A.Normalize /
ToTensorV2; utils/dataset.py:100-134: masks as float / 255).

Albumentations and OpenCV are absent from the build container, so the arithmetic below follows their published
semantics and PARITY IS UNPINNED at this boundary: cv2.resize(INTER_LINEAR) samples at (dst + 0.5) * scale - 0.5 with a
replicated border, cv2.warpAffine(INTER_LINEAR, BORDER_REFLECT_101) samples at M^-1 [x, y, 1]; both round to nearest
(OpenCV's 11-bit fixed-point weights may differ from this float evaluation by one grey level).  Masks use
INTER_NEAREST.  Only tests/ may import this module."""
import numpy as np

IMAGENET_MEAN = np.array([0.485, 0.456, 0.406], np.float32)      # trainer.py:48-49
IMAGENET_STD = np.array([0.229, 0.224, 0.225], np.float32)


def resize_matrix(hs, ws, h, w):
    """dst pixel -> src coordinates of cv2.resize."""
    fx, fy = ws / w, hs / h
    return np.array([fx, 0, 0.5 * fx - 0.5, 0, fy, 0.5 * fy - 0.5], np.float32)


def shift_scale_rotate_matrix(h, w, angle_deg, scale, dx, dy, hflip=False):
    """Inverse (dst -> src) map of A.ShiftScaleRotate(angle, scale, dx, dy) followed by A.HorizontalFlip:
    forward M = getRotationMatrix2D(centre=(w/2 - 0.5, h/2 - 0.5), angle, scale) with M[:, 2] += (dx * w, dy * h)."""
    cx, cy = w / 2 - 0.5, h / 2 - 0.5
    a = np.deg2rad(angle_deg)
    al, be = scale * np.cos(a), scale * np.sin(a)
    M = np.array([[al, be, (1 - al) * cx - be * cy + dx * w], [-be, al, be * cx + (1 - al) * cy + dy * h], [0, 0, 1]], np.float64)
    if hflip:                                  # x' = w - 1 - x applied after the warp
        M = np.array([[-1, 0, w - 1], [0, 1, 0], [0, 0, 1]], np.float64) @ M
    return np.linalg.inv(M)[:2].reshape(-1).astype(np.float32)


def _reflect101(i, n):
    if n == 1:
        return np.zeros_like(i)
    i = np.abs(i)
    period = 2 * n - 2
    i = i % period
    return np.where(i >= n, period - i, i)


def warp_u8(img, m, h, w, nearest=False, reflect=False):
    """img [Hs, Ws, C] uint8 -> [h, w, C] uint8."""
    hs, ws, _ = img.shape
    m = m.astype(np.float32)
    xx, yy = np.meshgrid(np.arange(w, dtype=np.float32), np.arange(h, dtype=np.float32))
    sx = m[0] * xx + m[1] * yy + m[2]
    sy = m[3] * xx + m[4] * yy + m[5]

    def at(yi, xi):
        if reflect:
            yi, xi = _reflect101(yi, hs), _reflect101(xi, ws)
        else:
            yi, xi = np.clip(yi, 0, hs - 1), np.clip(xi, 0, ws - 1)
        return img[yi, xi].astype(np.float32)

    if nearest:
        return at(np.floor(sy + np.float32(0.5)).astype(np.int64), np.floor(sx + np.float32(0.5)).astype(np.int64)).astype(np.uint8)
    fx, fy = np.floor(sx), np.floor(sy)
    x0, y0 = fx.astype(np.int64), fy.astype(np.int64)
    ax, ay = (sx - fx)[..., None], (sy - fy)[..., None]
    top = at(y0, x0) * (1 - ax) + at(y0, x0 + 1) * ax
    bot = at(y0 + 1, x0) * (1 - ax) + at(y0 + 1, x0 + 1) * ax
    return np.clip(np.rint(top * (1 - ay) + bot * ay), 0, 255).astype(np.uint8)


def normalize_u8(img, alpha=None, beta=None, mean=IMAGENET_MEAN, std=IMAGENET_STD):
    """[H, W, C] uint8 -> [C, H, W] float32 (A.RandomBrightnessContrast in the uint8 domain, A.Normalize, ToTensorV2)."""
    v = img.astype(np.float32)
    if alpha is not None:
        v = np.clip(np.rint(np.float32(alpha) * v + np.float32(beta) * np.float32(255)), 0, 255)
    v = v * np.float32(1 / 255)
    if mean is not None:
        v = (v - mean) / std
    return np.ascontiguousarray(v.transpose(2, 0, 1))


def val_seg_sample(img, mask, size=256):
    """val_seg_transform (trainer.py:100-112) + dataset.py:120-126."""
    hs, ws, _ = img.shape
    m = resize_matrix(hs, ws, size, size)
    mm = resize_matrix(mask.shape[0], mask.shape[1], size, size)      # (A.Resize maps image and mask from their own extents)
    return normalize_u8(warp_u8(img, m, size, size)), normalize_u8(warp_u8(mask, mm, size, size, nearest=True), mean=None)


def train_seg_sample(img, mask, angle, scale, dx, dy, hflip, alpha, beta, size=256):
    """train_seg_transform (trainer.py:83-98) with its random draws given explicitly."""
    hs, ws, _ = img.shape
    m0 = resize_matrix(hs, ws, size, size)
    mm = resize_matrix(mask.shape[0], mask.shape[1], size, size)
    i1, k1 = warp_u8(img, m0, size, size), warp_u8(mask, mm, size, size, nearest=True)
    m1 = shift_scale_rotate_matrix(size, size, angle, scale, dx, dy, hflip)
    i2, k2 = warp_u8(i1, m1, size, size, reflect=True), warp_u8(k1, m1, size, size, nearest=True, reflect=True)
    return normalize_u8(i2, alpha, beta), normalize_u8(k2, mean=None)


def longest_max_size_pad(img, size=256):
    """A.LongestMaxSize(size) then A.PadIfNeeded(size, size, border_mode=BORDER_CONSTANT, value=0) (trainer.py:54-60):
    the longer side becomes `size` (bilinear, aspect ratio kept, extents rounded half-to-even), the shorter one is padded
    with zeros, centred, the odd pixel on the bottom / right."""
    hs, ws, c = img.shape
    sc = size / max(hs, ws)
    h1, w1 = int(round(hs * sc)), int(round(ws * sc))
    small = warp_u8(img, resize_matrix(hs, ws, h1, w1), h1, w1)
    out = np.zeros((size, size, c), np.uint8)
    top, left = (size - h1) // 2, (size - w1) // 2
    out[top:top + h1, left:left + w1] = small
    return out


def val_cls_sample(img, size=256):
    """val_cls_transform (trainer.py:70-82)."""
    return normalize_u8(longest_max_size_pad(img, size))


def train_cls_sample(img, angle, scale, dx, dy, hflip, alpha, beta, size=256):
    """train_cls_transform (trainer.py:52-68) with its random draws given explicitly."""
    i1 = longest_max_size_pad(img, size)
    m1 = shift_scale_rotate_matrix(size, size, angle, scale, dx, dy, hflip)
    return normalize_u8(warp_u8(i1, m1, size, size, reflect=True), alpha, beta)
