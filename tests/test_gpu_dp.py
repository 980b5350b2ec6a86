"""-m gpu: the data-parallel step end to end with REAL kernels and streams, world_size 2.
Both ranks share the one GPU of the test box (RCCL needs one device per rank, so the exchange runs over gloo on
CUDA tensors — the bucket schedule, the comm / side-stream event edges, the inv_scale folding and the kernels are
the production ones).  Checks after two optimiser steps of Attention U-Net (bf16, several gradient buckets):
  * both ranks hold bit-identical parameters;
  * they equal a single-process emulation: gradients of shard 0 and shard 1 computed one after the other from the
    same weights, summed, then clip(1.0) + AdamW with inv_scale = 1/2 (train-mode BN statistics stay per shard, as
    in DDP without SyncBN — SURVEY.md §8e)."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
STEPS, B, HW, LR = 2, 4, 64, 1e-3


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _setup():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "medical-image-segmentation-and-classification_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from oracle import nets, train as otrain
    from models.segmentation_models.AttentionUNet import AttentionUNet
    m = AttentionUNet()
    m.load_state_dict(nets.default_init_state("AttentionUNet", seed=0))
    m.compute_dtype = torch.bfloat16
    m = m.to(DEV).train()
    shards = [[otrain.synthetic_batch(B, HW, seed=10 * s + r) for s in range(STEPS)] for r in range(2)]
    return m, shards


def _worker(rank, world, port, q, wire=None):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m, shards = _setup()
        from mi355 import nn as mnn, optim as moptim
        from mi355.dp import DataParallel
        dp = DataParallel(m, bucket_mb=8.0, overlap=True, bucket_dtype=wire)
        assert dp.world == 2 and dp.wire_dtype == wire
        opt = moptim.AdamW(m.parameters(), lr=LR, weight_decay=5e-4)
        opt.inv_scale = dp.inv_scale
        crit = mnn.BCEWithLogitsLoss()
        for s in range(STEPS):
            x, y = shards[rank][s]
            opt.zero_grad(set_to_none=True)
            loss = crit(dp(x.to(DEV)), y.to(DEV))
            loss.backward()
            moptim.clip_grad_norm_(m.parameters(), 1.0, inv_scale=dp.inv_scale)
            opt.step()
        torch.cuda.synchronize()
        plan = [p for p in m.engine.plans.values() if p.dout is not None][0]
        # a numpy array travels through the queue by value; a torch tensor travels as a file descriptor that the parent must fetch
        # from this process while it is still alive (seen once: ConnectionResetError when the worker had already exited)
        q.put((rank, "ok", m.engine.flat_p.detach().cpu().numpy(), len(dp.schedule(plan))))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, "fail", traceback.format_exc(), 0))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("wire", [None, torch.bfloat16])
def test_dp_world2_matches_single_process_emulation(wire):
    """wire=bfloat16: SURVEY.md 8e's perf mode — every bucket rounded to bf16 (mi355_grads_to_wire), summed in bf16, widened back
    (mi355_grads_from_wire).  The emulation rounds each shard's gradients, adds the two in fp32 and rounds the sum: what a two-rank
    sum of bf16 values is."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, wire)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=400) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
    for rank, status, info, _ in res:
        assert status == "ok", f"rank {rank}: {info}"
    p0, p1 = torch.from_numpy(res[0][2]), torch.from_numpy(res[1][2])
    assert res[0][3] > 1                              # 140 MB of gradients in 8 MB buckets: the overlap path ran
    assert torch.equal(p0, p1)                        # same reduced gradients + same update => bit-identical replicas

    m, shards = _setup()
    from mi355 import nn as mnn, optim as moptim
    opt = moptim.AdamW(m.parameters(), lr=LR, weight_decay=5e-4)
    opt.inv_scale = 0.5
    crit = mnn.BCEWithLogitsLoss()
    eng = m.engine
    for s in range(STEPS):
        total = None
        for r in range(2):
            x, y = shards[r][s]
            opt.zero_grad(set_to_none=True)
            crit(m(x.to(DEV)), y.to(DEV)).backward()
            g = eng.flat_g.clone() if wire is None else eng.flat_g.to(wire).float()
            total = g if total is None else total + g
        eng.flat_g.copy_(total if wire is None else total.to(wire).float())
        moptim.clip_grad_norm_(m.parameters(), 1.0, inv_scale=0.5)
        opt.step()
    torch.cuda.synchronize()
    ref = eng.flat_p.detach().cpu()
    err = float((p0 - ref).abs().max() / ref.abs().max())
    assert err <= 1e-6, err


def _rccl_worker(port, native, q, wire=None):
    """A fresh process: RCCL (torch.distributed backend "nccl") with ONE rank — the production code path of bench.py --gpus N
    (process group, comm stream, bucket schedule, event edges, all-reduce launches on the device) minus the other ranks."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if native:
        os.environ["MI355_DP_NATIVE"] = "1"
    try:
        m, shards = _setup()                                   # imports the package BEFORE the first CUDA call
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
        from mi355 import nn as mnn, optim as moptim
        from mi355.dp import DataParallel
        from mi355.lib import lib
        out = {}
        for tag in ("dp", "plain"):
            m, _ = _setup()
            if tag == "dp":
                dp = DataParallel(m, bucket_mb=8.0, overlap=True, force=True, bucket_dtype=wire)
                assert dp.native == bool(native) and dp.inv_scale == 1.0
            opt = moptim.AdamW(m.parameters(), lr=LR, weight_decay=5e-4)
            crit = mnn.BCEWithLogitsLoss()
            for s_ in range(STEPS):
                x, y = shards[0][s_]
                opt.zero_grad(set_to_none=True)
                crit(m(x.to(DEV)), y.to(DEV)).backward()
                if tag == "plain" and wire is not None:          # what one rank's bucket is after the trip over the wire
                    m.engine.flat_g.copy_(m.engine.flat_g.to(wire).float())
                moptim.clip_grad_norm_(m.parameters(), 1.0)
                opt.step()
            torch.cuda.synchronize()
            out[tag] = m.engine.flat_p.detach().cpu().numpy()
            if tag == "dp":
                plan = [p for p in m.engine.plans.values() if p.dout is not None][0]
                out["buckets"] = len(dp.schedule(plan))
                out["events"] = len(dp._events[id(plan)])
                out["comm_world"] = lib.mi355_comm_world()
        q.put(("ok", out))
    except Exception:  # noqa: BLE001
        import traceback
        q.put(("fail", traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("native,wire", [(0, None), (1, None), (0, torch.bfloat16), (1, torch.float16)])
def test_rccl_code_path_with_one_rank(native, wire):
    """backend="nccl" IS RCCL on ROCm: a spawned child runs init_process_group("nccl", world_size=1), DataParallel(force=True)
    and two optimiser steps.  An all-reduce over one rank is the identity, so the parameters must be BIT-identical to the
    same steps without the data-parallel runner, while everything around it is the multi-GPU path: several buckets, their
    pre-allocated event pairs, the comm stream.  native=1: the same through the library's own RCCL entry points
    (mi355_comm_init / mi355_allreduce_bucket, MI355_DP_NATIVE=1).  wire: two-byte buckets (RCCL sums bf16 / fp16 staging buffers);
    the plain run rounds its gradients the same way before the clip, so the parameters stay bit-identical."""
    import numpy as np
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), native, q, wire))
    p.start()
    status, out = q.get(timeout=400)
    p.join(timeout=60)
    assert status == "ok", out
    assert out["buckets"] > 1 and out["events"] == out["buckets"]
    assert out["comm_world"] == (1 if native else 0)
    assert np.array_equal(out["dp"], out["plain"])


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_gradient_wire_kernels_round_like_torch(dtype):
    """mi355_grads_to_wire / mi355_grads_from_wire on bucket-shaped ranges: a start that is a multiple of four gradients but not of
    eight (8-byte aligned wire), a length that is not a multiple of four; bit-identical to torch's rounding, neighbours untouched."""
    _setup()
    from mi355.lib import lib
    code = {torch.bfloat16: 1, torch.float16: 2}[dtype]
    gen = torch.Generator().manual_seed(5)
    g = (torch.randn(5000, generator=gen) * torch.logspace(-6, 3, 5000)).to(DEV)
    w = torch.full((5000,), 7.0, dtype=dtype, device=DEV)
    s = torch.cuda.current_stream().cuda_stream
    for lo, hi in ((4, 1007), (1012, 1012), (2000, 4999)):
        lib.mi355_grads_to_wire(g[lo:hi], w[lo:hi], hi - lo, code, s)
        assert torch.equal(w[lo:hi], g[lo:hi].to(dtype))
        assert float(w[lo - 1]) == 7.0 and float(w[hi]) == 7.0
        back = torch.full_like(g, -3.0)
        lib.mi355_grads_from_wire(w[lo:hi], back[lo:hi], hi - lo, code, s)
        assert torch.equal(back[lo:hi], g[lo:hi].to(dtype).float())
        assert float(back[lo - 1]) == -3.0 and float(back[hi]) == -3.0
    with pytest.raises(RuntimeError):           # a wire pointer that is only 2-byte aligned
        lib.mi355_grads_to_wire(g[4:100], w[5:101], 96, code, s)
