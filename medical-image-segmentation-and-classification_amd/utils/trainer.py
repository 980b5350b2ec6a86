"""Training entry point of the MI355X path — the counterpart of the reference's utils/trainer.py
(a ``__main__`` script: datasets -> loaders -> ``train()`` for each model, trainer.py:163-232).

The reference reads the COVID-19 Radiography dataset through Albumentations/cv2 (absent offline and
out of scope); this driver keeps its protocol — classification bs=16, segmentation bs=8,
80/20 split, 20 epochs, lr 1e-6 (trainer.py:28-37,159-160,199-210) — on any ``(x, y)`` datasets and
defaults to synthetic 256x256 batches so that it runs anywhere:

    python utils/trainer.py --task seg --model attentionunet --epochs 2 --samples 64

With ``--data-root dataset`` (the reference's DATA_ROOT layout: ``splits/train.csv``, ``<class>/images|masks/<id>.png``) it reads the
real files instead: two dataset objects per task with the train / val transforms, one 80/20 index split shared by both
(trainer.py:119-151), PNGs decoded by native threads and transformed on the GPU (utils/dataset.py, utils/gpu_transforms.py).

Data parallel (BASELINE.json north star; the reference is single-device): one process per GPU,

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 utils/trainer.py --task seg ...

The script joins the RCCL process group (backend "nccl"), draws the 80/20 split from a seed rank 0 broadcasts, and ``train()``
does the rest: batches sharded by rank, gradients all-reduced over xGMI during backward, rank 0 prints and saves.
"""
import argparse
import os
import sys

import torch
from torch.utils.data import DataLoader, TensorDataset, random_split

PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if PKG not in sys.path:
    sys.path.insert(0, PKG)

from utils.helpers import get_class_model, get_seg_model, train  # noqa: E402

IMG_SIZE = 256
CLS_MODELS = ["resnet18", "resnet50", "vgg16", "vgg19"]
SEG_MODELS = ["resnetunet", "attentionunet", "r2unet", "r2attunet"]


def synthetic_dataset(task, n, size, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, 3, size, size, generator=g)
    if task == "cls":
        return TensorDataset(x, torch.randint(0, 3, (n,), generator=g))
    yy, xx = torch.meshgrid(torch.arange(size, dtype=torch.float32), torch.arange(size, dtype=torch.float32), indexing="ij")
    c = torch.rand(n, 4, generator=g)
    cy, cx = (0.4 + 0.2 * c[:, 0]) * size, (0.4 + 0.2 * c[:, 1]) * size
    ry, rx = (0.25 + 0.1 * c[:, 2]) * size, (0.25 + 0.1 * c[:, 3]) * size
    m = ((((yy[None] - cy[:, None, None]) / ry[:, None, None]) ** 2 + ((xx[None] - cx[:, None, None]) / rx[:, None, None]) ** 2) <= 1).float()
    return TensorDataset(x + m[:, None] * torch.tensor([1.0, -0.7, 0.4]).view(1, 3, 1, 1), m[:, None])


def make_loader(ds, bs, shuffle):
    return DataLoader(ds, batch_size=bs, shuffle=shuffle, num_workers=0, pin_memory=True, drop_last=False)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--task", choices=["cls", "seg", "both"], default="seg")
    ap.add_argument("--model", default=None, help="one model name; default: every model of the task (trainer.py:163-168)")
    ap.add_argument("--epochs", type=int, default=20)
    ap.add_argument("--lr", type=float, default=1e-6)
    ap.add_argument("--samples", type=int, default=64)
    ap.add_argument("--size", type=int, default=IMG_SIZE)
    ap.add_argument("--save-dir", default="weights")
    ap.add_argument("--data-root", default=None, help="dataset directory in the reference's layout; default: synthetic batches")
    args = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("the MI355X path needs a GPU (no CPU fallback)")
    # one process per GPU under torch.distributed.run: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the launcher
    world, rank, local_rank = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    # (MI355_DP_BACKEND=gloo: ranks may share a GPU — a rehearsal of the launcher path on a one-GPU box, tests/test_gpu_dp_train.py;
    #  RCCL wants a device per rank)
    backend = os.environ.get("MI355_DP_BACKEND", "nccl")
    local_dev = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_dev)
    device = torch.device("cuda", local_dev)
    split_gen = None                                      # (single process: unseeded, like the reference's random_split)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
        seed = [int(torch.randint(0, 2 ** 31 - 1, (1,)))]
        dist.broadcast_object_list(seed, src=0)          # every rank must cut the SAME 80/20 split
        split_gen = torch.Generator().manual_seed(seed[0])
    say = print if rank == 0 else (lambda *a, **k: None)
    results = {}
    for task in (["cls", "seg"] if args.task == "both" else [args.task]):
        names = [args.model] if args.model else (CLS_MODELS if task == "cls" else SEG_MODELS)
        bs = 16 if task == "cls" else 8
        if args.data_root:
            from utils.dataset import ClassificationDataset, GpuBatchLoader, SegmentationDataset
            from utils.gpu_transforms import ClsBatchTransform, SegBatchTransform
            DS, TF = (ClassificationDataset, ClsBatchTransform) if task == "cls" else (SegmentationDataset, SegBatchTransform)
            ds_tr = DS(args.data_root, TF(args.size, train=True, device=device), "train")
            ds_va = DS(args.data_root, TF(args.size, train=False, device=device), "train")
            if len(ds_tr) == 0:
                say(f"{task} dataset is empty under {args.data_root}. Skipping.")
                continue
            n_train = int(0.8 * len(ds_tr))
            perm = torch.randperm(len(ds_tr), generator=split_gen).tolist()      # (unseeded, like the reference's random_split)
            train_dl = GpuBatchLoader(ds_tr, bs, shuffle=True, device=device, indices=perm[:n_train])
            val_dl = GpuBatchLoader(ds_va, bs, shuffle=True, device=device, indices=perm[n_train:])
        else:
            full = synthetic_dataset(task, args.samples, args.size)
            n_train = int(0.8 * len(full))
            tr, va = random_split(full, [n_train, len(full) - n_train], generator=split_gen) if split_gen is not None else \
                random_split(full, [n_train, len(full) - n_train])
            train_dl, val_dl = make_loader(tr, bs, True), make_loader(va, bs, False)
        for name in names:
            say(f"\n{'=' * 20} {task.upper()} :: {name} {'=' * 20}")
            if task == "cls":
                model, head = get_class_model(name)
                best = train(model, train_dl, val_dl, device, args.epochs, args.lr, name, os.path.join(args.save_dir, "classification_models"),
                             seg=False, cls_head_name=head)
            else:
                model = get_seg_model(name)
                best = train(model, train_dl, val_dl, device, args.epochs, args.lr, name, os.path.join(args.save_dir, "segmentation_models"), seg=True)
            results[(task, name)] = best
    say("\n===== SUMMARY =====")
    for (task, name), best in results.items():
        say(f"{task:>4} {name:<14} best {'val loss' if task == 'seg' else 'val acc'}: {best:.4f}")
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
