"""Evaluation metrics and loops of the MI355X path — same names and return values as the
reference's utils/tester.py (segmentation metrics :92-193, classification metrics :49-88, eval
loops :197-312).  Per-sample counters come from one HIP reduction per batch instead of ~9 host
syncs per sample."""
import os

import sys

import numpy as np
import torch

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))       # `python utils/tester.py` (tester.py:22-24 does the same)
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

from mi355.lib import lib  # noqa: E402

CLASSES = ["COVID", "Healthy", "Non-COVID"]
DATA_ROOT = "dataset"
WEIGHTS_ROOT = "weights"
CLS_WEIGHTS_DIR = os.path.join(WEIGHTS_ROOT, "classification_models")
SEG_WEIGHTS_DIR = os.path.join(WEIGHTS_ROOT, "segmentation_models")
IMG_SIZE = 256
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
    print("\nPer-Class Metrics:")
    for i, c in enumerate(CLASSES[:len(m["f1_per_class"])]):   # (the reference indexes all of CLASSES, tester.py:282-296, and dies when a class is absent)
        print(f"\n{c}:\n  Precision: {m['precision_per_class'][i]:.2f}%\n  Recall:    {m['recall_per_class'][i]:.2f}%\n"
              f"  F1 Score:  {m['f1_per_class'][i]:.2f}%")
    print("\nConfusion Matrix:")
    print((" " * 25).join(f"{c:>12}" for c in CLASSES))      # the reference's header: names joined by 12 + 1 + 12 blanks (:299)
    for i, row in enumerate(m["confusion_matrix"][:len(CLASSES)]):
        print(f"{CLASSES[i]:<12}" + "".join(f"{v:>12}" for v in row))
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


# ---- the whole-zoo driver and its reports (tester.py:513-876) -------------------------------------------------------
_CLS_FILES = {"ResNet18": "ResNet18_best_acc.pt", "ResNet50": "ResNet50_best_acc.pt", "VGG16": "VGG16_best_acc.pt",
              "VGG19": "VGG19_best_acc.pt", "CLIP": "CLIP_best_acc.pt"}
_SEG_FILES = {"ResNetUnet": "ResNetUnet_best_loss.pt", "AttentionUNet": "AttentionUNet_best_loss.pt",
              "R2Unet": "R2Unet_best_loss.pt", "R2AttUnet": "R2AttUnet_best_loss.pt", "CLIPSeg": "CLIPSeg_best_loss.pt"}


def test_all_models(device="cuda", batch_size=16, cls_loader=None, seg_loader=None, cls_weights_dir=None,
                    seg_weights_dir=None):
    """Evaluate every checkpoint found under the weights directories (tester.py:513-735): same model names, file
    names, skip rules and result dictionary.  Like the reference (:531-555, :569-580, :651-666) the test loaders are built
    from ``DATA_ROOT/splits/test.csv`` with the validation transforms — here `utils.dataset` + `GpuBatchLoader` (native PNG
    decode, transforms on the GPU), batch_size for classification and batch_size // 2 for segmentation (:663); a caller may pass
    its own loaders instead, and when neither exists the reference's "dataset not found" branch is taken (:637-639, :729-731).
    The CLIP / CLIPSeg entries (hub models, out of scope: SURVEY.md section 8) are reported and skipped.  Checkpoints are the
    reference's own format: a plain `state_dict` saved by `train` (helpers.py:394-400)."""
    from utils.helpers import get_class_model, get_seg_model
    if not torch.cuda.is_available():
        raise RuntimeError("test_all_models: the MI355X path needs a GPU (the reference falls back to the CPU, tester.py:524)")
    device = torch.device(device)
    print(f"[INFO] Using device: {device}")
    cls_weights_dir = CLS_WEIGHTS_DIR if cls_weights_dir is None else cls_weights_dir
    seg_weights_dir = SEG_WEIGHTS_DIR if seg_weights_dir is None else seg_weights_dir
    results = {}

    def run(files, wdir, loader, build, test, n_samples_label):
        print(f"\n[INFO] {n_samples_label} Test Dataset: {len(loader.dataset)} samples")
        for model_name, weight_file in files.items():
            weight_path = os.path.join(wdir, weight_file)
            if not os.path.exists(weight_path):
                print(f"\n[WARNING] Weights not found for {model_name}: {weight_path}")
                print(f"Skipping {model_name}...")
                continue
            if model_name in ("CLIP", "CLIPSeg"):
                print(f"\n[WARNING] {model_name} is a hub model outside the MI355X conv path; skipping {weight_path}")
                continue
            try:
                model = build(model_name)
                model.load_state_dict(torch.load(weight_path, map_location=device))
                model = model.to(device)
                results[model_name] = test(model, loader, device, model_name)
                del model
                torch.cuda.empty_cache()
            except Exception as e:      # (the reference reports and moves on to the next model, :630-635)
                print(f"\n[ERROR] Failed to test {model_name}: {e}")
                import traceback
                traceback.print_exc()
                continue

    def default_loader(seg):
        """test split of DATA_ROOT through the GPU input pipeline, or None (split file / directory absent)"""
        try:
            from utils.dataset import ClassificationDataset, GpuBatchLoader, SegmentationDataset
            from utils.gpu_transforms import ClsBatchTransform, SegBatchTransform
            if seg:
                ds = SegmentationDataset(DATA_ROOT, SegBatchTransform(IMG_SIZE, train=False, device=device), split="test")
                return GpuBatchLoader(ds, max(1, batch_size // 2), shuffle=False, device=device)
            ds = ClassificationDataset(DATA_ROOT, ClsBatchTransform(IMG_SIZE, train=False, device=device), split="test")
            return GpuBatchLoader(ds, batch_size, shuffle=False, device=device)
        except FileNotFoundError:
            return None

    if cls_loader is None:
        cls_loader = default_loader(False)
    if seg_loader is None:
        seg_loader = default_loader(True)
    if cls_loader is None:
        print(f"\n[WARNING] Classification test dataset not found: no loader given for {DATA_ROOT!r}")
        print("Skipping classification model testing...")
    else:
        run(_CLS_FILES, cls_weights_dir, cls_loader, lambda n: get_class_model(n)[0], test_classification_model, "Classification")
    if seg_loader is None:
        print(f"\n[WARNING] Segmentation test dataset not found: no loader given for {DATA_ROOT!r}")
        print("Skipping segmentation model testing...")
    elif len(seg_loader.dataset) == 0:
        print("\n[WARNING] Segmentation test dataset is empty. Skipping segmentation testing.")
    else:
        run(_SEG_FILES, seg_weights_dir, seg_loader, get_seg_model, test_segmentation_model, "Segmentation")
    return results


def print_summary(results):
    """The two result tables and the best model of each family (tester.py:738-805), same text."""
    if not results:
        print("\n[INFO] No test results to display.")
        return
    print("\n" + "=" * 80)
    print(" " * 25 + "TEST RESULTS SUMMARY")
    print("=" * 80)
    cls_models = [m for m in ["ResNet18", "ResNet50", "VGG16", "VGG19", "CLIP"] if m in results]
    if cls_models:
        print("\nCLASSIFICATION MODELS:")
        print("-" * 80)
        print(f"{'Model':<20} {'Accuracy':<12} {'Precision':<12} {'Recall':<12} {'F1 Score':<12}")
        print("-" * 80)
        for model in cls_models:
            m = results[model]
            print(f"{model:<20} {m['accuracy']:>10.2f}% {m['precision']:>10.2f}% {m['recall']:>10.2f}% {m['f1']:>10.2f}%")
        best = max(cls_models, key=lambda x: results[x]["accuracy"])
        print(f"\n\U0001F3C6 Best Classification Model: {best} (Accuracy: {results[best]['accuracy']:.2f}%)")
    seg_models = [m for m in ["ResNetUnet", "AttentionUNet", "R2Unet", "R2AttUnet", "CLIPSeg"] if m in results]
    if seg_models:
        print("\n\nSEGMENTATION MODELS:")
        print("-" * 80)
        print(f"{'Model':<20} {'IoU':<10} {'Dice':<10} {'Precision':<12} {'Recall':<12} {'F1 Score':<12}")
        print("-" * 80)
        for model in seg_models:
            m = results[model]
            print(f"{model:<20} {m['iou']:>8.2f}% {m['dice']:>8.2f}% {m['precision']:>10.2f}% {m['recall']:>10.2f}% {m['f1']:>10.2f}%")
        best = max(seg_models, key=lambda x: results[x]["dice"])
        print(f"\n\U0001F3C6 Best Segmentation Model: {best} (Dice: {results[best]['dice']:.2f}%)")
    print("=" * 80 + "\n")


def save_results_to_csv(results, cls_output_path="results/classification_test_results.csv",
                        seg_output_path="results/segmentation_test_results.csv"):
    """One CSV per family, a `Model` column followed by the scalar metrics in dictionary order (tester.py:808-876;
    written through pandas like the reference, so the float formatting is identical)."""
    if not results:
        print("\n[INFO] No results to save.")
        return
    import pandas as pd
    cls_models = [k for k in results.keys() if any(x in k for x in ["ResNet18", "ResNet50", "VGG", "CLIP"]) and "Seg" not in k]
    seg_models = [k for k in results.keys() if "Unet" in k or "UNet" in k or "CLIPSeg" in k]
    if cls_models:
        rows = []
        for name in cls_models:
            row = {"Model": name}
            row.update(results[name])
            for k in ("confusion_matrix", "precision_per_class", "recall_per_class", "f1_per_class"):
                row.pop(k, None)
            rows.append(row)
        pd.DataFrame(rows).to_csv(cls_output_path, index=False)
        print(f"\n[INFO] Classification results saved to: {cls_output_path}")
    else:
        print("\n[INFO] No classification results to save.")
    if seg_models:
        rows = []
        for name in seg_models:
            row = {"Model": name}
            row.update(results[name])
            rows.append(row)
        pd.DataFrame(rows).to_csv(seg_output_path, index=False)
        print(f"[INFO] Segmentation results saved to: {seg_output_path}")
    else:
        print("\n[INFO] No segmentation results to save.")


if __name__ == "__main__":          # python utils/tester.py (tester.py:879-898): same banner, same three calls
    print("\n" + "=" * 80)
    print(" " * 20 + "MODEL TESTING UTILITY")
    print("=" * 80)
    results = test_all_models(device="cuda", batch_size=16)
    print_summary(results)
    save_results_to_csv(results, cls_output_path="classification_test_results.csv", seg_output_path="segmentation_test_results.csv")
    print("\n[INFO] Testing complete!")
