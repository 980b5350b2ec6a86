"""Training driver and factories of the MI355X path — same public names, signatures and
printed lines as the reference's utils/helpers.py (acc/iou :219-227, factories :124-213,
train :231-412), executing on the HIP launch-plan engine.

Differences that do not change results: the compute dtype is a property of the model (bf16 default, fp32,
or fp16 with the device-side ``mi355.amp.GradScaler`` in the place of ``torch.amp.GradScaler``,
helpers.py:285,323-336) instead of autocast, the per-step ``loss.item()`` host syncs (helpers.py:337,341) are replaced
by device-side accumulation read back once per epoch, and model factories construct the local
classes directly (the reference first tries a torch.hub download, helpers.py:158-166, which has
no network here).

Data parallel: when ``torch.distributed`` is initialised with more than one rank (``torchrun --nproc-per-node N
utils/trainer.py ...``), ``train()`` shards the loaders' batches by rank (utils/distributed.py), wraps the model in
``mi355.dp.DataParallel`` (bucketed RCCL all-reduce of the flat gradient buffer overlapped with backward, 1/world folded
into clip / AdamW), sums the epoch's loss / metric accumulators over ranks with one collective per pass, and lets rank 0
alone print and write checkpoints; every rank returns the same ``best_score`` and stops early at the same epoch."""
import math
import os
import time

import torch
import torch.nn as nn

from mi355 import amp as mamp
from mi355 import nn as mnn
from mi355 import optim as moptim
from mi355.lib import lib
from utils.distributed import RankShard, all_reduce_sums, dist_info

CLASSES = ["COVID", "Healthy", "Non-COVID"]


# ---- model factories ---------------------------------------------------------------------------
def add_dropout_to_fc(model, p=0.5, classes=CLASSES):
    """Swap the classification layer for Dropout(p) + Linear(in, len(classes)); returns the name of
    the head attribute ("fc" / "classifier") that stage 1 trains (helpers.py:124-144)."""
    n_out = len(classes)
    if hasattr(model, "fc"):
        model.fc = nn.Sequential(nn.Dropout(p=p), nn.Linear(model.fc.in_features, n_out))
        return "fc"
    head = getattr(model, "classifier", None)
    if isinstance(head, nn.Sequential):
        kept = list(head.children())
        last = kept.pop()
        model.classifier = nn.Sequential(*kept, nn.Dropout(p=p), nn.Linear(last.in_features, n_out))
        return "classifier"
    return None


def get_class_model(name, hub=None):
    """name -> (model, head attribute name); random init.  The reference first asks torch.hub for the torchvision
    model (`resnet18` / `resnet50`; `vgg16` -> `vgg16_bn`, helpers.py:158-166) and falls back to its local classes when
    that fails (:170-192).  There is no network here, so by default "vgg16" / "vgg19" / "resnet*" are the local classes
    (the reference's offline behaviour); ``hub=True`` (or MI355_HUB_LAYOUT=1) selects the torchvision layouts the online
    reference trains — `TorchvisionResNet.resnet18/50`, `VGG.VGG16_BN/VGG19_BN` — into which hub checkpoints load
    unchanged.  The hub layouts are also reachable by their own names "vgg16_bn" / "vgg19_bn" / "resnet18_tv" / "resnet50_tv"."""
    from models.classification_models import ResNet, TorchvisionResNet, VGG
    local = {"resnet18": ResNet.ResNet18, "resnet50": ResNet.ResNet50, "vgg16": VGG.VGG16, "vgg19": VGG.VGG19}
    tv = {"resnet18": TorchvisionResNet.resnet18, "resnet50": TorchvisionResNet.resnet50, "vgg16": VGG.VGG16_BN, "vgg19": VGG.VGG19_BN}
    named = {"vgg16_bn": VGG.VGG16_BN, "vgg19_bn": VGG.VGG19_BN, "resnet18_tv": TorchvisionResNet.resnet18,
             "resnet50_tv": TorchvisionResNet.resnet50}
    if hub is None:
        hub = os.environ.get("MI355_HUB_LAYOUT", "0") == "1"
    key = name.lower()
    table = {**(tv if hub else local), **named}
    if key not in table:
        raise ValueError(f"Unknown classification model: {name}")
    model = table[key](num_classes=1000)
    return model, add_dropout_to_fc(model, p=0.5)


def get_seg_model(name):
    """name -> segmentation model instance (helpers.py:195-213)."""
    key = name.lower()
    if key == "attentionunet":
        from models.segmentation_models.AttentionUNet import AttentionUNet
        return AttentionUNet()
    if key == "r2unet":
        from models.segmentation_models.R2U_Net import R2U_Net
        return R2U_Net()
    if key == "r2attunet":
        from models.segmentation_models.R2AttU_Net import R2AttU_Net
        return R2AttU_Net()
    if key == "resnetunet":
        from models.segmentation_models.ResnetUnet import ResNetUnet
        return ResNetUnet()
    raise ValueError(f"Unknown segmentation model: {name}")


# ---- metrics --------------------------------------------------------------------------------------
def acc(logits, y):
    return (torch.argmax(logits, 1) == y).sum().item(), y.size(0)


def _iou_device(pred, mask, t=0.5, is_logit=False):
    """Whole-batch IoU of helpers.py:223-227 as a 0-dim device tensor (no sync)."""
    cnt = torch.empty(4, dtype=torch.float32, device=pred.device)
    lib.mi355_seg_counts(pred.contiguous(), mask.contiguous(), cnt, 1, pred.numel(), 1 if is_logit else 0, float(t))
    inter, union = cnt[0], cnt[1] + cnt[2] - cnt[0]
    return inter / (union + 1e-7)


def iou(pred, mask, t=0.5):
    if pred.is_cuda:
        return _iou_device(pred.float(), mask.float(), t).item()
    p = (pred > t).float()
    inter = (p * mask).sum()
    union = ((p + mask) > 0).float().sum()
    return (inter / (union + 1e-7)).item()


# ---- training ----------------------------------------------------------------------------------------
def _make_optimizer(params, lr):
    return moptim.AdamW(params, lr=lr, weight_decay=5e-4)


def train(model, train_dl, val_dl, device, epochs, lr, name, save_dir, seg=False, cls_head_name=None):
    device = torch.device(device)
    model = model.to(device)
    if hasattr(model, "engine"):
        model.engine._check_storage()                 # flat fp32 parameter / gradient buffers
    criterion = mnn.BCEWithLogitsLoss() if seg else mnn.CrossEntropyLoss(label_smoothing=0.1)
    STAGE1 = 5

    # data parallel (module docstring): the reference's loop on this rank's batches, gradients averaged over ranks
    rank, world = dist_info()
    dp, inv_scale = None, 1.0
    say = print if rank == 0 else (lambda *a, **k: None)
    n_train, n_val, n_val_batches = len(train_dl.dataset), len(val_dl.dataset), len(val_dl)
    if world > 1:
        from mi355.dp import DataParallel
        dp = DataParallel(model)                      # replicas start from rank 0's parameters and buffers
        inv_scale = dp.inv_scale
        train_dl, val_dl = RankShard(train_dl, rank, world, pad=True), RankShard(val_dl, rank, world, pad=False)

    def make_optimizer(params, lr_):
        opt = _make_optimizer(params, lr_)
        opt.inv_scale = inv_scale                     # gradient averaging over ranks is folded into clip + AdamW
        return opt

    if seg:
        optimizer = make_optimizer(model.parameters(), lr)
        say(f"Training Segmentation model (all layers unfrozen) with LR: {lr}")
        scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max=epochs)
    else:
        say(f"--- STAGE 1: Feature Extraction (Epochs 1-{STAGE1}) ---")
        for p in model.parameters():
            p.requires_grad = False
        head_params = []
        if cls_head_name:
            for p in getattr(model, cls_head_name).parameters():
                p.requires_grad = True
                head_params.append(p)
        optimizer = make_optimizer(head_params, 1e-4)
        scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max=STAGE1)

    # helpers.py:285 — scaling is enabled exactly where the reference's autocast would compute in fp16
    from mi355 import engine as _engine
    scaler = mamp.GradScaler(enabled=(getattr(model, "compute_dtype", None) or _engine._DEFAULT_DTYPE) == torch.float16)
    best_score = float("inf") if seg else 0.0
    patience, stale = 10, 0
    t0 = time.time()

    for epoch in range(1, epochs + 1):
        if not seg and epoch == STAGE1 + 1:
            say(f"\n--- STAGE 2: Full Fine-Tuning (Epochs {epoch}-{epochs}) ---")
            for p in model.parameters():
                p.requires_grad = True
            optimizer = make_optimizer(model.parameters(), lr)
            scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, mode="max", factor=0.1, patience=3)
            say(f"Full fine-tuning (all layers unfrozen) with very low LR: {lr}. Using ReduceLROnPlateau scheduler.")

        model.train()
        loss_sum = torch.zeros((), device=device)
        hit_sum = torch.zeros((), device=device)
        seen = 0
        seen_all = 0                                  # samples this rank trained on (padding batches of the last step included)
        for x, y in train_dl:
            x, y = x.to(device, non_blocking=True), y.to(device, non_blocking=True)
            optimizer.zero_grad(set_to_none=True)
            out = model(x)
            if seg and out.dim() == 3:
                out = out.unsqueeze(1)
            loss = criterion(out, y)
            scaler.scale(loss).backward()
            scaler.unscale_(optimizer)
            moptim.clip_grad_norm_(model.parameters(), max_norm=1.0, inv_scale=inv_scale)
            scaler.step(optimizer)
            scaler.update()
            loss_sum += loss.detach() * x.size(0)
            seen_all += x.size(0)
            if not seg:
                hit_sum += (torch.argmax(out.detach(), 1) == y).sum()
                seen += y.size(0)

        if dp is not None:
            dp.sync_buffers()                         # rank 0's BatchNorm statistics everywhere: one model is validated and saved
        model.eval()
        vloss = torch.zeros((), device=device)
        vmetric = torch.zeros((), device=device)
        with torch.no_grad():
            for x, y in val_dl:
                x, y = x.to(device), y.to(device)
                out = model(x)
                if seg and out.dim() == 3:
                    out = out.unsqueeze(1)
                vloss += criterion(out, y) * x.size(0)
                if seg:
                    vmetric += _iou_device(out.float(), y.float(), 0.5, is_logit=True)
                else:
                    vmetric += (torch.argmax(out, 1) == y).sum()

        # one host read-back per epoch (data parallel: one collective over the five accumulators first; the padded last
        # optimiser step makes the number of samples trained on larger than the dataset, so the mean divides by what was seen)
        if dp is not None:
            cnt = torch.tensor([float(seen_all), float(seen)], device=device, dtype=torch.float64)
            loss_sum, hit_sum, vloss, vmetric, seen_all_t, seen_t = all_reduce_sums(loss_sum, hit_sum, vloss, vmetric, cnt[0], cnt[1])
            n_train_seen, seen = int(seen_all_t.item()), int(seen_t.item())
        else:
            n_train_seen = n_train
        train_loss = loss_sum.item() / n_train_seen
        val_loss = vloss.item() / n_val
        if seg:
            val_iou = vmetric.item() / n_val_batches
            score = val_loss
            say(f"[{name}] Ep{epoch}: TrainLoss {train_loss:.3f} | ValLoss {val_loss:.3f} | IoU {val_iou:.3f}")
            improved = val_loss < best_score
        else:
            train_acc = 100 * hit_sum.item() / max(seen, 1)
            val_acc = 100 * vmetric.item() / n_val
            score = val_acc
            say(f"[{name}] Ep{epoch}: TrainLoss {train_loss:.3f} (Acc {train_acc:.2f}%) | ValLoss {val_loss:.3f} | ValAcc {val_acc:.2f}%")
            improved = val_acc > best_score

        if seg or epoch <= STAGE1:
            scheduler.step()
        else:
            scheduler.step(score)

        if improved:
            best_score, stale = score, 0
            if rank == 0:
                os.makedirs(save_dir, exist_ok=True)
                fname = f"{name}_best_loss.pt" if seg else f"{name}_best_acc.pt"
                torch.save(model.state_dict(), os.path.join(save_dir, fname))
        else:
            stale += 1
        if stale >= patience:
            say(f"Early stopping at epoch {epoch}. Best score: {best_score:.2f}")
            break

    say(f"Training for {name} finished in {(time.time() - t0) / 60:.2f} minutes.")
    return best_score
