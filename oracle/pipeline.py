"""TEST INFRASTRUCTURE — CPU restatement of the reference's joint inference path (utils/pipeline.py:324-357
``_predict_classification`` / ``_predict_segmentation`` and the decision in ``process_image`` :359-418), over the
functional models of oracle/nets.py.  One image at a time, exactly as the reference runs it:

    logits = cls(img); probs = softmax(logits)[0]; confidence, idx = max(probs); prediction = CLASSES[idx]
    if prediction == "COVID":  mask = (sigmoid(seg(img)) > 0.5).uint8 * 255   else  no mask

Parity unpinned at this boundary: utils/pipeline.py cannot be imported here (torchvision / cv2 /
albumentations are absent), so there are no reference-generated vectors for the glue; the two models it composes
are pinned by tests/golden/model_*.npz.  Only tests/ may import this module."""
import torch

from . import nets

CLASSES = ["COVID", "Healthy", "Non-COVID"]          # pipeline.py:22

_FWD = nets.NETS


@torch.no_grad()
def process_batch(cls_name, cls_sd, seg_name, seg_sd, x):
    """x: [B,3,H,W] normalised float.  -> list of (prediction, confidence_percent, mask uint8 [H,W] or None)."""
    out = []
    for i in range(x.shape[0]):
        img = x[i:i + 1]
        logits = _FWD[cls_name]({k: v.clone() for k, v in cls_sd.items()}, img, False)
        probs = torch.softmax(logits, dim=1)[0]
        conf, idx = torch.max(probs, 0)
        pred = CLASSES[int(idx)]
        mask = None
        if pred == "COVID":
            z = _FWD[seg_name]({k: v.clone() for k, v in seg_sd.items()}, img, False)
            mask = ((torch.sigmoid(z)[0, 0] > 0.5).to(torch.uint8) * 255)
        out.append((pred, float(conf) * 100, mask))
    return out
