"""Evaluation metrics and loops of the MI355X path — same names and return values as the
reference's utils/tester.py (segmentation metrics :92-193, classification metrics :49-88, eval
loops :197-312).  Per-sample counters come from one HIP reduction per batch instead of ~9 host
syncs per sample."""
import numpy as np
import torch

from mi355.lib import lib

CLASSES = ["COVID", "Healthy", "Non-COVID"]
_E = 1e-7


def _counts(pred, target, threshold, is_logit=False):
    """[B,4] float64 numpy: tp, predicted-positive, target-positive, equal — per sample."""
    B = pred.shape[0]
    per = pred[0].numel()
    if pred.is_cuda:
        cnt = torch.empty(B, 4, dtype=torch.float32, device=pred.device)
        lib.mi355_seg_counts(pred.float().contiguous(), target.float().contiguous(), cnt, B, per, 1 if is_logit else 0,
                             float(threshold))
        return cnt.double().cpu().numpy(), per
    p = (torch.sigmoid(pred) if is_logit else pred) > threshold
    t = target > threshold
    f = lambda z: z.reshape(B, -1).sum(1).double()
    return torch.stack([f(p & t), f(p), f(t), f(p == t)], 1).numpy(), per


def _metrics_from_counts(c, per):
    tp, pp, tt, eq = c
    fp, fn, union = pp - tp, tt - tp, pp + tt - tp
    prec = (tp + _E) / (tp + fp + _E)
    rec = (tp + _E) / (tp + fn + _E)
    return {"iou": (tp + _E) / (union + _E) * 100, "dice": (2.0 * tp + _E) / (pp + tt + _E) * 100,
            "pixel_accuracy": eq / per * 100, "precision": prec * 100, "recall": rec * 100,
            "f1": 2 * (prec * rec) / (prec + rec + _E) * 100}


def _single(pred, target, threshold):
    c, per = _counts(pred.reshape(1, -1), target.reshape(1, -1), threshold)
    return _metrics_from_counts(c[0], per)


def calculate_iou(pred, target, threshold=0.5):
    return _single(pred, target, threshold)["iou"] / 100


def calculate_dice(pred, target, threshold=0.5):
    return _single(pred, target, threshold)["dice"] / 100


def calculate_pixel_accuracy(pred, target, threshold=0.5):
    return _single(pred, target, threshold)["pixel_accuracy"] / 100


def calculate_segmentation_metrics(pred, target, threshold=0.5):
    return _single(pred, target, threshold)


def calculate_classification_metrics(all_preds, all_labels):
    """Accuracy and support-weighted precision/recall/F1 (+ per class, confusion matrix), the
    quantities sklearn's *_score(average="weighted", zero_division=0) return (tester.py:49-88)."""
    p = np.asarray(all_preds).astype(np.int64)
    y = np.asarray(all_labels).astype(np.int64)
    labels = np.unique(np.concatenate([p, y]))
    k = len(labels)
    idx = {int(l): i for i, l in enumerate(labels)}
    cm = np.zeros((k, k), dtype=np.int64)
    for a, b in zip(y, p):
        cm[idx[int(a)], idx[int(b)]] += 1
    tp = np.diag(cm).astype(np.float64)
    pred_pos, support = cm.sum(0).astype(np.float64), cm.sum(1).astype(np.float64)
    prec = np.divide(tp, pred_pos, out=np.zeros(k), where=pred_pos > 0)
    rec = np.divide(tp, support, out=np.zeros(k), where=support > 0)
    f1 = np.divide(2 * prec * rec, prec + rec, out=np.zeros(k), where=(prec + rec) > 0)
    w = support / support.sum()
    return {"accuracy": float((p == y).mean()) * 100, "precision": float((prec * w).sum()) * 100,
            "recall": float((rec * w).sum()) * 100, "f1": float((f1 * w).sum()) * 100,
            "precision_per_class": prec * 100, "recall_per_class": rec * 100, "f1_per_class": f1 * 100,
            "confusion_matrix": cm}


def test_classification_model(model, test_loader, device, model_name):
    model.eval()
    preds, labels = [], []
    print(f"\n{'=' * 60}\nTesting Classification Model: {model_name}\n{'=' * 60}")
    with torch.no_grad():
        for images, y in test_loader:
            out = model(images.to(device))
            preds.append(torch.max(out, 1)[1])
            labels.append(y.to(device))
    m = calculate_classification_metrics(torch.cat(preds).cpu().numpy(), torch.cat(labels).cpu().numpy())
    print(f"\n{model_name} Test Results:\n{'-' * 60}")
    print(f"Accuracy:  {m['accuracy']:.2f}%\nPrecision: {m['precision']:.2f}%\nRecall:    {m['recall']:.2f}%\nF1 Score:  {m['f1']:.2f}%")
    print(f"{'=' * 60}\n")
    return m


def test_segmentation_model(model, test_loader, device, model_name):
    model.eval()
    tot = {k: 0.0 for k in ("iou", "dice", "pixel_accuracy", "precision", "recall", "f1")}
    n = 0
    print(f"\n{'=' * 60}\nTesting Segmentation Model: {model_name}\n{'=' * 60}")
    pending = []
    with torch.no_grad():
        for images, masks in test_loader:
            out = model(images.to(device))
            if out.dim() == 3:
                out = out.unsqueeze(1)
            masks = masks.to(device)
            B, per = out.shape[0], out[0].numel()
            cnt = torch.empty(B, 4, dtype=torch.float32, device=out.device)
            lib.mi355_seg_counts(out.float().contiguous(), masks.float().contiguous(), cnt, B, per, 1, 0.5)
            pending.append((cnt, per))
    for cnt, per in pending:                      # single read-back after the loop
        for c in cnt.double().cpu().numpy():
            m = _metrics_from_counts(c, per)
            for k in tot:
                tot[k] += m[k]
            n += 1
    avg = {k: v / n for k, v in tot.items()}
    print(f"\n{model_name} Test Results:\n{'-' * 60}")
    print(f"IoU (Jaccard):     {avg['iou']:.2f}%\nDice Coefficient:  {avg['dice']:.2f}%\nPixel Accuracy:    {avg['pixel_accuracy']:.2f}%")
    print(f"Precision:         {avg['precision']:.2f}%\nRecall:            {avg['recall']:.2f}%\nF1 Score:          {avg['f1']:.2f}%\n{'=' * 60}\n")
    return avg
