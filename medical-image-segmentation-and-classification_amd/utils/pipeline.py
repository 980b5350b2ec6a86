"""Joint inference path of the reference (utils/pipeline.py:324-357 ``_predict_classification`` /
``_predict_segmentation``, decision logic of ``process_image`` :359-418) as ONE batched pass on the GPU:

    classify the batch -> softmax / argmax / confidence on the device -> compact the samples predicted
    "COVID" -> segment only those -> sigmoid > 0.5 -> uint8 masks scattered back to their batch slots.

The reference handles one PIL image per call and synchronises after every model; here the only host
round-trip is the number of kept samples (it sizes the segmentation launch).  The compacted batch is
padded to a multiple of ``bucket`` so that at most B / bucket launch plans ever exist per image size.
Image decoding / resizing / normalisation (pipeline.py:196-215, 381-390) and the red overlay (:399-413)
are host-side presentation code outside the hot path: ``predict`` takes the already normalised tensor
(``val_transform`` output) and returns the masks."""
from __future__ import annotations

import torch

from mi355.lib import lib

CLASSES = ["COVID", "Healthy", "Non-COVID"]          # pipeline.py:22


class JointPipeline:
    def __init__(self, classification_model, segmentation_model, device="cuda", classes=CLASSES, positive="COVID", bucket=4):
        self.device = torch.device(device)
        self.classes = list(classes)
        self.keep = self.classes.index(positive)
        self.bucket = int(bucket)
        self.classification_model = classification_model.to(self.device).eval()
        self.segmentation_model = None if segmentation_model is None else segmentation_model.to(self.device).eval()

    @torch.no_grad()
    def predict(self, x):
        """x: [B,3,H,W] normalised float (B <= 1024).  Returns a dict of device tensors:
        ``pred`` int32 [B] class index, ``confidence`` float [B] in percent, ``masks`` uint8 [B,H,W] (0 / 255; all zero
        where no segmentation ran), ``segmented`` bool [B]."""
        x = x.to(self.device, dtype=torch.float32).contiguous()
        B, _, H, W = x.shape
        logits = self.classification_model(x).float().contiguous()
        pred = torch.empty(B, dtype=torch.int32, device=self.device)
        conf = torch.empty(B, dtype=torch.float32, device=self.device)
        kept = torch.empty(B + self.bucket, dtype=torch.int32, device=self.device)
        n_kept = torch.empty(1, dtype=torch.int32, device=self.device)
        lib.mi355_cls_decide(logits, B, logits.shape[1], self.keep, pred, conf, kept, n_kept)
        masks = torch.zeros(B, H, W, dtype=torch.uint8, device=self.device)
        segmented = pred == self.keep
        n = int(n_kept)                               # the one host sync: sizes the segmentation launch
        if n and self.segmentation_model is not None:
            npad = -(-n // self.bucket) * self.bucket
            if npad > n:
                kept[n:npad] = kept[0]                 # padding rows repeat a kept sample; their output is dropped
            xs = torch.empty(npad, 3, H, W, dtype=torch.float32, device=self.device)
            lib.mi355_gather_rows(x, kept, npad, 3 * H * W, xs)
            z = self.segmentation_model(xs)
            if z.dim() == 3:
                z = z.unsqueeze(1)
            lib.mi355_mask_scatter(z.float().contiguous(), kept, n, H * W, 0.5, masks)
        elif self.segmentation_model is None:
            segmented = torch.zeros_like(segmented)
        return {"pred": pred, "confidence": conf, "masks": masks, "segmented": segmented}

    def process_files(self, paths, size=256, threads=8):
        """PNG files -> the reference's per-image results.  The reference opens each file with PIL, applies ``A.Resize(256, 256)`` +
        ``A.Normalize`` + ``ToTensorV2`` (pipeline.py:186-193, 381-390) and runs the two models; here the files of one size are
        decoded by native threads (utils/dataset.py) and resized / normalised on the GPU (utils/gpu_transforms.py) as ONE batch."""
        from utils.dataset import decode_batch, read_files
        from utils.gpu_transforms import SegBatchTransform
        imgs = decode_batch(read_files(paths), 3, threads, names=list(paths))
        x = SegBatchTransform(size, train=False, device=self.device)(imgs.to(self.device, non_blocking=True))
        return self.process_batch(x)

    def process_batch(self, x):
        """The reference's per-image result shape: list of (prediction, confidence_percent, mask uint8 [H,W] or None)."""
        r = self.predict(x)
        pred, conf, seg = r["pred"].cpu(), r["confidence"].cpu(), r["segmented"].cpu()
        masks = r["masks"].cpu()
        return [(self.classes[int(pred[i])], float(conf[i]), masks[i] if bool(seg[i]) else None) for i in range(len(pred))]
