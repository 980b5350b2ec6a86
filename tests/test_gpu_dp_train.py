"""-m gpu: the drop-in ``train()`` (utils/helpers.py) under data parallelism, world_size 2, real kernels.

Both ranks share the test box's one GPU and exchange over gloo (RCCL wants a device per rank; everything else — RankShard,
DataParallel's bucket schedule and streams, inv_scale in clip / AdamW, the per-epoch buffer sync and accumulator all-reduce,
rank-0-only printing and checkpointing — is the production path of ``torchrun utils/trainer.py``).  Checked against a
single-process EMULATION of the same protocol (two replicas stepped in lock-step, gradients summed by hand):
  * both ranks return the same best score and hold bit-identical state_dicts (parameters AND BatchNorm buffers);
  * rank 0's epoch log is the emulation's, line for line; rank 1 prints nothing; only rank 0 writes the checkpoint;
  * parameters equal the emulation's to 1e-6.
Reference loop: /root/reference/utils/helpers.py:317-342 (train), 345-360 (validation), 394-406 (checkpoint / early stop)."""
import io
import os
import socket
import sys
from contextlib import redirect_stdout

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
EPOCHS, BS, HW, LR, N_TRAIN, N_VAL = 2, 4, 32, 1e-3, 20, 12          # 5 train batches (padded to 6: 3 steps / epoch), 3 val batches


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _paths():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "medical-image-segmentation-and-classification_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _model():
    from oracle import nets
    from models.segmentation_models.AttentionUNet import AttentionUNet
    m = AttentionUNet()
    m.load_state_dict(nets.default_init_state("AttentionUNet", seed=0))
    m.compute_dtype = torch.bfloat16
    return m


def _loaders():
    from torch.utils.data import DataLoader, TensorDataset
    from oracle import train as otrain
    xt, yt = otrain.synthetic_batch(N_TRAIN, HW, seed=1)
    xv, yv = otrain.synthetic_batch(N_VAL, HW, seed=2)
    return (DataLoader(TensorDataset(xt, yt), batch_size=BS, shuffle=False), DataLoader(TensorDataset(xv, yv), batch_size=BS, shuffle=False))


def _worker(rank, world, port, save_dir, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    _paths()
    m = _model()                                           # imports the package before the first CUDA call
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from utils.helpers import train
        if rank == 1:                                      # a rank-dependent start: train() must begin from rank 0's state
            with torch.no_grad():
                for p in m.parameters():
                    p.mul_(1.5)
        tr, va = _loaders()
        buf = io.StringIO()
        with redirect_stdout(buf):
            best = train(m, tr, va, DEV, EPOCHS, LR, "AttentionUNet", save_dir, seg=True)
        torch.cuda.synchronize()
        sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
        q.put((rank, "ok", best, buf.getvalue(), sd))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, "fail", traceback.format_exc(), "", None))
    finally:
        dist.destroy_process_group()


def _emulate():
    """Two replicas in one process: the protocol of utils/distributed.py + helpers.train() written out by hand."""
    _paths()
    from mi355 import nn as mnn, optim as moptim
    from utils.distributed import shard_batches
    from utils.helpers import _iou_device
    ms = [_model().to(DEV) for _ in range(2)]
    opts = []
    for m in ms:
        m.engine._check_storage()
        o = moptim.AdamW(m.parameters(), lr=LR, weight_decay=5e-4)
        o.inv_scale = 0.5
        opts.append(o)
    scheds = [torch.optim.lr_scheduler.CosineAnnealingLR(o, T_max=EPOCHS) for o in opts]
    crit = mnn.BCEWithLogitsLoss()
    tr, va = _loaders()
    log, best = [], float("inf")
    for epoch in range(1, EPOCHS + 1):
        tb = [shard_batches(list(tr), r, 2, pad=True) for r in range(2)]
        loss_sum = [torch.zeros((), device=DEV) for _ in range(2)]
        seen = [0, 0]
        for m in ms:
            m.train()
        for k in range(len(tb[0])):
            grads = []
            for r, m in enumerate(ms):
                x, y = (t.to(DEV) for t in tb[r][k])
                opts[r].zero_grad(set_to_none=True)
                loss = crit(m(x), y)
                loss.backward()
                loss_sum[r] += loss.detach() * x.size(0)
                seen[r] += x.size(0)
                grads.append(m.engine.flat_g.clone())
            total = grads[0] + grads[1]
            for r, m in enumerate(ms):
                m.engine.flat_g.copy_(total)
                moptim.clip_grad_norm_(m.parameters(), max_norm=1.0, inv_scale=0.5)
                opts[r].step()
        with torch.no_grad():
            for b0, b1 in zip(ms[0].buffers(), ms[1].buffers()):
                b1.copy_(b0)
        vb = [shard_batches(list(va), r, 2, pad=False) for r in range(2)]
        vloss = [torch.zeros((), device=DEV) for _ in range(2)]
        viou = [torch.zeros((), device=DEV) for _ in range(2)]
        for r, m in enumerate(ms):
            m.eval()
            with torch.no_grad():
                for x, y in vb[r]:
                    x, y = x.to(DEV), y.to(DEV)
                    out = m(x)
                    vloss[r] += crit(out, y) * x.size(0)
                    viou[r] += _iou_device(out.float(), y.float(), 0.5, is_logit=True)
        s = lambda pair: float(pair[0].double() + pair[1].double())
        train_loss, val_loss, val_iou = s(loss_sum) / (seen[0] + seen[1]), s(vloss) / N_VAL, s(viou) / len(va)
        log.append(f"[AttentionUNet] Ep{epoch}: TrainLoss {train_loss:.3f} | ValLoss {val_loss:.3f} | IoU {val_iou:.3f}")
        for sc in scheds:
            sc.step()
        best = min(best, val_loss)
    torch.cuda.synchronize()
    return log, best, ms[0]


def test_train_world2_matches_single_process_emulation(tmp_path):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    save_dir = str(tmp_path / "w")
    procs = [ctx.Process(target=_worker, args=(r, 2, port, save_dir, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
    for rank, status, info, _, _ in res:
        assert status == "ok", f"rank {rank}: {info}"
    (_, _, best0, log0, sd0), (_, _, best1, log1, sd1) = res
    import numpy as np
    assert best0 == best1
    assert set(sd0) == set(sd1) and all(np.array_equal(sd0[k], sd1[k]) for k in sd0)       # parameters AND BatchNorm buffers
    assert log1 == ""                                                                      # rank 1 is silent
    assert os.listdir(save_dir) == ["AttentionUNet_best_loss.pt"]                          # written once, by rank 0
    log, best, m = _emulate()
    lines0 = [ln for ln in log0.splitlines() if ln.startswith("[AttentionUNet] Ep")]
    assert lines0 == log, (lines0, log)
    assert abs(best0 - best) <= 1e-6 * abs(best)
    ref = m.state_dict()
    for k, v in sd0.items():
        r = ref[k].detach().cpu().numpy()
        assert np.abs(v.astype(np.float64) - r).max() <= 1e-6 * max(np.abs(r).max(), 1e-30) + 1e-12, k
    saved = torch.load(os.path.join(save_dir, "AttentionUNet_best_loss.pt"), map_location="cpu")
    assert set(saved) == set(sd0)


def test_trainer_script_under_the_launcher_two_ranks(tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 2 utils/trainer.py ...` end to end (the command INTEGRATION.md gives, with
    MI355_DP_BACKEND=gloo because both ranks share this box's one GPU): the script joins the group, both ranks cut the same 80 / 20
    split, train() shards the batches, rank 0 alone prints the epoch lines and the summary and writes the checkpoint."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = os.path.join(root, "medical-image-segmentation-and-classification_amd", "utils", "trainer.py")
    env = dict(os.environ, MI355_DP_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), script, "--task", "seg", "--model", "attentionunet", "--epochs", "2", "--samples", "20",
           "--size", "32", "--save-dir", str(tmp_path / "w")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("[attentionunet] Ep")]
    assert len(lines) == 2, r.stdout                       # one line per epoch: rank 1 printed nothing
    assert r.stdout.count("===== SUMMARY =====") == 1 and "best val loss" in r.stdout
    assert os.listdir(str(tmp_path / "w" / "segmentation_models")) == ["attentionunet_best_loss.pt"]


def _cls_worker(rank, world, port, save_dir, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    _paths()
    from utils.helpers import get_class_model, train
    torch.manual_seed(rank)                                # rank-dependent initialisation: train() starts from rank 0's
    m, head = get_class_model("resnet18")
    m.compute_dtype = torch.bfloat16
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from torch.utils.data import DataLoader, TensorDataset
        g = torch.Generator().manual_seed(3)
        xt, yt = torch.randn(24, 3, 64, 64, generator=g), torch.randint(0, 3, (24,), generator=g)
        xv, yv = torch.randn(12, 3, 64, 64, generator=g), torch.randint(0, 3, (12,), generator=g)
        tr = DataLoader(TensorDataset(xt, yt), batch_size=4, shuffle=False)
        va = DataLoader(TensorDataset(xv, yv), batch_size=4, shuffle=False)
        buf = io.StringIO()
        with redirect_stdout(buf):
            best = train(m, tr, va, DEV, 7, 1e-4, "resnet18", save_dir, seg=False, cls_head_name=head)
        torch.cuda.synchronize()
        sd = {k: v.detach().float().cpu().numpy() for k, v in m.state_dict().items()}
        frozen = [k for k, p in m.named_parameters() if not p.requires_grad]
        q.put((rank, "ok", best, buf.getvalue(), sd, frozen))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, "fail", traceback.format_exc(), "", None, None))
    finally:
        dist.destroy_process_group()


def test_train_world2_classification_crosses_the_stage_switch(tmp_path):
    """The two-stage classifier protocol (helpers.py:267-283 head only for five epochs, :296-309 everything from epoch 6) under data
    parallelism: stage 1's plan has gradients for the head alone (two small buckets), stage 2 re-plans with every parameter — the
    bucket schedule, inv_scale and the buffer sync must follow.  Dropout masks differ per rank, so there is no single-process
    emulation to compare with; the invariants are: both ranks end with bit-identical parameters AND BatchNorm buffers, the same
    best accuracy, nothing left frozen, rank 1 silent, one checkpoint written by rank 0, both stage banners printed once."""
    import numpy as np
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    save_dir = str(tmp_path / "c")
    port = _free_port()
    procs = [ctx.Process(target=_cls_worker, args=(r, 2, port, save_dir, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
    for rank, status, info, *_ in res:
        assert status == "ok", f"rank {rank}: {info}"
    (_, _, best0, log0, sd0, fr0), (_, _, best1, log1, sd1, fr1) = res
    assert best0 == best1 and 0.0 <= best0 <= 100.0          # (accuracy in percent, helpers.py:363)
    assert set(sd0) == set(sd1)
    for k in sd0:
        assert np.array_equal(sd0[k], sd1[k]), k
    assert fr0 == [] and fr1 == []
    assert log1 == ""
    assert log0.count("--- STAGE 1") == 1 and log0.count("--- STAGE 2") == 1
    assert len([ln for ln in log0.splitlines() if ln.startswith("[resnet18] Ep")]) == 7
    assert os.listdir(save_dir) == ["resnet18_best_acc.pt"]
