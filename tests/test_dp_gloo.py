"""CPU, world_size 2 over gloo: the data-parallel wrapper's bucket schedule and gradient exchange.
No kernels run here (no GPU): the backward plan's launches are replaced by a fake launcher that
writes a rank-dependent value into every parameter-gradient range a launch would write, so the test
checks exactly the host logic that matters for N>1: every bucket is reduced only after the last
launch writing into it, buckets tile the gradient buffer, and after backward every rank holds the
SUM over ranks; replicas built from different seeds start from rank 0's parameters and buffers (the 1/world factor is folded into clip/AdamW via inv_scale)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, bucket_mb, q, wire=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path[:0] = [root, os.path.join(root, "medical-image-segmentation-and-classification_amd")]
        from mi355 import graph
        from mi355.dp import DataParallel
        from models.segmentation_models.AttentionUNet import AttentionUNet
        torch.manual_seed(rank)                         # rank-dependent initialisation AND BatchNorm buffers ...
        net = AttentionUNet().train()
        eng = net.engine
        eng.flatten()                                   # CPU flat buffers (no kernels are launched)
        with torch.no_grad():
            for b in net.buffers():
                b.add_(rank + 1)
        plan = eng.plan_for((2, 3, 32, 32), True, True, torch.float32)
        dp = DataParallel(net, bucket_mb=bucket_mb, overlap=True, bucket_dtype=wire)
        assert dp.wire_dtype == wire
        assert dp.world == world and abs(dp.inv_scale - 1.0 / world) < 1e-12
        # ... are replaced by rank 0's at construction (DDP semantics): replicas apply summed gradients to the SAME weights
        torch.manual_seed(0)
        ref = AttentionUNet()
        for (k, v), (_, w) in zip(net.state_dict().items(), ref.state_dict().items()):
            want = w if k in dict(ref.named_parameters()) else w + 1
            assert torch.equal(v, want), k
        assert all(p.data_ptr() == eng.flat_p.data_ptr() + eng.offsets[id(p)][0] * 4 for p in net.parameters())
        buckets = dp.schedule(plan)
        # 1) buckets tile the gradient ranges of all parameters, each exactly once
        spans = sorted((eng.offsets[id(p)][0], eng.offsets[id(p)][0] + eng.offsets[id(p)][1]) for p in plan.grad_params)
        assert len(plan.grad_params) == len(list(net.parameters()))
        covered = sorted((lo, hi) for _, lo, hi in buckets)
        assert covered[0][0] == spans[0][0] and covered[-1][1] == spans[-1][1]
        for (a0, a1), (b0, b1) in zip(covered, covered[1:]):
            assert a1 <= b0
        # 2) a bucket is ready only after the last launch that writes any parameter inside it
        for ready, lo, hi in buckets:
            for p in plan.grad_params:
                o, n = eng.offsets[id(p)]
                if lo <= o < hi:
                    assert plan.last_write[id(p)] <= ready
        assert [b[0] for b in buckets] == sorted(b[0] for b in buckets)
        # 3) fake backward: every launch "writes" rank+1 into the gradient ranges it references
        log = []
        real_allreduce = dp._allreduce

        def fake_run(first, last):
            for fn, args, name, l in plan.bind(0)[1][first:last]:
                for a in l.args:
                    if isinstance(a, graph.GRef):
                        o, n = eng.offsets[id(a.param)]
                        eng.flat_g[o:o + n] = float(rank + 1)
                log.append(("launch", name))

        def spy_allreduce(lo, hi):
            log.append(("allreduce", lo, hi))
            real_allreduce(lo, hi)

        dp.run_calls, dp._allreduce = fake_run, spy_allreduce
        eng.flat_g.zero_()
        dp._run_backward(plan, 0)
        expect = float(sum(r + 1 for r in range(world)))
        zero = {id(p) for p in plan.zero_grad_params}    # biases in front of BN: no launch, the sum of zeros stays zero
        for p in plan.grad_params:
            o, n = eng.offsets[id(p)]
            assert torch.all(eng.flat_g[o:o + n] == (0.0 if id(p) in zero else expect)), p.shape
        n_ar = sum(1 for e in log if e[0] == "allreduce")
        assert n_ar == len(buckets)
        # all-reduces are interleaved with launches (overlap schedule), not all at the end
        first_ar = next(i for i, e in enumerate(log) if e[0] == "allreduce")
        last_launch = max(i for i, e in enumerate(log) if e[0] == "launch")
        if len(buckets) > 1:
            assert first_ar < last_launch
        q.put((rank, "ok", len(buckets)))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "fail", traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("bucket_mb,wire", [(8.0, None), (1000.0, None), (8.0, torch.bfloat16)])
def test_bucketed_allreduce_world2(bucket_mb, wire):
    """wire=bfloat16: the buckets travel as bf16 staging buffers (SURVEY.md 8e's perf mode); the fake gradients 1.0 / 2.0 and their
    sum are exact in bf16, so the same assertions hold."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, bucket_mb, q, wire)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, status, info in res:
        assert status == "ok", f"rank {rank}: {info}"
    if bucket_mb < 100:
        assert res[0][2] > 1            # 140 MB of gradients in 8 MB buckets
    else:
        assert res[0][2] == 1
