#!/usr/bin/env python3
"""Headline benchmark: images/sec of one Attention U-Net TRAIN STEP (256x256, bs=32 per GPU, bf16)
on the MI355X launch-plan path — zero_grad -> forward -> BCEWithLogits -> backward -> [RCCL gradient
all-reduce] -> clip_grad_norm(1.0) -> AdamW, exactly the per-batch body of the reference's train()
(utils/helpers.py:320-336) with inputs already resident in HBM.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (contract in the task description) carrying two extra objects:
  roofline      dominant kernel: algorithmic FLOPs per launch / average launch duration measured
                with HIP events on the launch stream in an instrumented replay of the same plan
  cpu_baseline  the CPU oracle (plain torch fp32 restatement of the reference) timed on this
                box's host cores on a bounded sample of the same workload
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "medical-image-segmentation-and-classification_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

# The step runs on up to four HIP streams (main, weight-gradient side stream, all-reduce stream, RCCL's own); the runtime maps
# streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues, and two of OUR streams sharing a queue serialises them
# (measured, one rank with the RCCL path on: 20.11 ms/step with 4 queues, 19.41 with 8; without RCCL 19.24 either way).
# Must be in the environment before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch
import torch.distributed as dist

PEAK_TFLOPS = {"bf16": 2500.0, "fp16": 2500.0, "fp32": 157.3}     # dense MFMA peaks, MI355X_MICROARCH.md
FWD_GFLOP_PER_IMG = 132.850                       # AttentionUNet 256x256 forward
TRAIN_GFLOP_PER_IMG = 398.32                      # 3x forward minus the first layer's dgrad


def make_batch(b, hw, seed, device):
    """images ~ N(0,1); target = one filled ellipse per image (25-40 % foreground) — SURVEY.md 8d."""
    import math
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(b, 3, hw, hw, generator=g)
    yy, xx = torch.meshgrid(torch.arange(hw, dtype=torch.float32), torch.arange(hw, dtype=torch.float32), indexing="ij")
    m = torch.zeros(b, 1, hw, hw)
    for i in range(b):
        r = torch.rand(4, generator=g)
        cy, cx = (0.4 + 0.2 * r[0]) * hw, (0.4 + 0.2 * r[1]) * hw
        area = (0.25 + 0.15 * r[2]) * hw * hw
        ratio = 0.7 + 0.6 * r[3]
        a, bb = math.sqrt(area / math.pi * ratio), math.sqrt(area / math.pi / ratio)
        m[i, 0] = ((((yy - cy) / a) ** 2 + ((xx - cx) / bb) ** 2) <= 1.0).float()
    return x.to(device), m.to(device)


SPLIT_BY_SHAPE = False


def dp_child_command(argv, gpus, port):
    """The command `bench.py --gpus N` runs when it was started WITHOUT a launcher (no WORLD_SIZE in the environment): one rank
    per GPU of this node under torch.distributed.run, the same arguments passed through."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(argv, gpus):
    """`python bench.py --gpus N` with N > 1 and no launcher: start the N ranks as a CHILD process (never exec: this process has
    imported torch; and never fall through to a one-GPU run labelled as N), relay rank 0's JSON line, exit with the child's code."""
    import socket
    import subprocess
    have = torch.cuda.device_count()                 # (counting devices does not initialise the runtime)
    if have < gpus:
        raise SystemExit(f"bench.py --gpus {gpus}: this node shows {have} GPU(s); refusing to report a smaller run under that label")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run(dp_child_command(argv, gpus, port), stdout=subprocess.PIPE, env=env, text=True)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        sys.stderr.write(r.stdout)
        raise SystemExit(r.returncode or f"bench.py --gpus {gpus}: the {gpus}-rank child printed no result line")
    res = json.loads(lines[-1])
    if res.get("n_gpus") != gpus:
        raise SystemExit(f"bench.py --gpus {gpus}: the child reported n_gpus={res.get('n_gpus')}")
    print(lines[-1], flush=True)
    raise SystemExit(0)


def profile_plan(plan, x, stream, reps=2):
    """Instrumented replay: HIP events around every launch of the forward and backward plans."""
    # single-stream resolution on purpose: HIP events on this stream must bracket every kernel (the production
    # plan forks the weight-gradient launches onto a side stream)
    fwd, bwd = plan._resolve(plan.pre + plan.fwd, stream), plan._resolve(plan.bwd, stream)
    agg = {}
    # one untimed replay first: the side-stream launches run on THIS stream here for the first time (vgg16_bn 512 x 512: the
    # stem's weight gradient took 60 ms in its first single-stream execution and 0.19 ms ever after)
    for i, (fn, args, name, l) in enumerate(list(fwd) + list(bwd)):
        rc = fn(x.data_ptr(), *args[1:]) if i == 0 else fn(*args)
        if rc:
            raise RuntimeError(f"{name} failed rc={rc}")
    torch.cuda.synchronize()
    for _ in range(reps):
        recs = []
        for i, (fn, args, name, l) in enumerate(list(fwd) + list(bwd)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(x.data_ptr(), *args[1:]) if i == 0 else fn(*args)
            e1.record()
            if rc:
                raise RuntimeError(f"{name} failed rc={rc}")
            recs.append((l, e0, e1))
        torch.cuda.synchronize()
        for l, e0, e1 in recs:
            key = l.tag or l.name
            if SPLIT_BY_SHAPE and l.name in ("mi355_bn_act", "mi355_bn_bwd_reduce", "mi355_bn_bwd_apply"):
                key = f"{l.name} M={l.args[-4]} C={l.args[-3]}"
            a = agg.setdefault(key, [0.0, 0, 0.0, 0.0, 0.0])
            dt = e0.elapsed_time(e1)
            a[0] += dt
            a[1] += 1
            a[2] += l.flops
            a[3] += l.bytes
            a[4] += dt * l.cus          # chip-time: a launch sized for half the CUs holds half the chip for its duration
    return agg


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 PMC pass (profiles/*pmc_traffic.json:
    separate --pmc FETCH_SIZE / WRITE_SIZE runs of this same command, FETCH_SIZE doubled as the gfx950 guide
    prescribes).  PMC counters cannot be read from inside the process, hence the indirection; None if absent."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")), reverse=True):
        try:
            d = json.load(open(f))
            e = d["kernels"].get(kernel)
            if e:
                return int(e["hbm_MB_per_launch"] * 1e6), os.path.relpath(f, ROOT)
        except Exception:  # noqa: BLE001
            continue
    return None, None


def host_cores():
    """Cores this process may actually use: the affinity mask, capped by the cgroup CPU quota when the box hands out a share of
    a larger host (a 1-GPU box owns 16 cores of the machine; running the oracle on every core the mask shows would only
    oversubscribe that share)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1"):
                n = max(1, min(n, int(float(quota) / period + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


# model of the GPU line -> (oracle net, state keyword arguments, segmentation?, CPU batch): the CPU leg runs the SAME model as the
# GPU line on a batch small enough for ~10-30 s of host work (BASELINE.md section 3 states the reduced batches)
CPU_CONFIGS = {
    "AttentionUNet": ("AttentionUNet", {}, True, 8),          # the reference's own segmentation batch size (trainer.py:160)
    "R2AttU_Net": ("R2AttU_Net", {}, True, 4),
    "R2U_Net": ("R2U_Net", {}, True, 4),
    "ResNetUnet": ("ResNetUnet", {}, True, 8),                # frozen ResNet-50 encoder: only the decoder trains (ResnetUnet.py:60-66)
    "resnet18": ("ResNet18", {"head_dropout": True}, False, 8),      # BASELINE.json configs[0]: bs 8
    "resnet50": ("ResNet50", {"head_dropout": True}, False, 8),
    "vgg16": ("VGG16", {"head_dropout": True}, False, 4),
    "vgg16_bn": ("VGG16_BN", {"num_classes": 3, "head_dropout": True}, False, 4),
}


def cpu_baseline(model, hw, steps=3):
    """SURVEY.md 8(d) protocol: the reference-equivalent CPU path (oracle/: plain torch fp32 restatement of the reference's
    train step, helpers.py:320-336) for the model of the GPU line, on ALL of this box's host cores (the process's affinity
    mask), a reduced batch (CPU_CONFIGS), one warm-up + 3 timed steps, images/s = B / median step time; CPU model and core
    count reported."""
    from oracle import nets, train as otrain
    net, kw, seg, bs = CPU_CONFIGS[model]
    # every core this process is entitled to: affinity mask / cgroup quota, and — the GPU pool's rule for a box that shows the
    # whole host — the 16-core share that comes with each visible GPU (BENCH_CPU_THREADS overrides)
    cores = min(host_cores(), 16 * max(1, torch.cuda.device_count()))
    if os.environ.get("BENCH_CPU_THREADS"):
        cores = max(1, int(os.environ["BENCH_CPU_THREADS"]))
    torch.set_num_threads(cores)
    sd = nets.default_init_state(net, seed=0, **kw)
    x, y = otrain.synthetic_batch(bs, hw, seed=0, classes=None if seg else 3)
    keys = nets.param_keys(sd)
    trainable = [k for k in keys if not k.startswith("encoder")] if net == "ResNetUnet" else None
    opt = otrain.AdamW(trainable or keys, 1e-6)
    otrain.train_step(net, sd, x, y, opt, seg, trainable)
    times = []
    for _ in range(steps):
        t0 = time.time()
        otrain.train_step(net, sd, x, y, opt, seg, trainable)
        times.append(time.time() - t0)
    dt = sorted(times)[len(times) // 2]
    return {"value": round(bs / dt, 3), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "cpu_model": cpu_model_name(),
            "sample": f"{net} {hw}x{hw} fp32 train step (fwd+{'BCE' if seg else 'CE'}+bwd+clip+AdamW"
                      f"{', frozen encoder' if trainable else ''}), B={bs} (the GPU line runs its own batch size), "
                      f"1 warm-up + {steps} timed steps, B / median step time",
            "step_seconds": [round(t, 3) for t in times]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--model", default="AttentionUNet")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--serial-streams", action="store_true",
                    help="keep the weight-gradient launches on the main stream (MI355_SIDE_STREAM=0): per-kernel durations "
                         "in a rocprofv3 trace are then not inflated by the wgrad / dgrad overlap of the production schedule")
    ap.add_argument("--table-rows", type=int, default=25)
    ap.add_argument("--split-by-shape", action="store_true", help="kernel table: one row per (launcher, M x C) of the BN kernels")
    ap.add_argument("--graph", type=int, default=0, help="capture the step into a hipGraph (1) or run eagerly (0)")
    ap.add_argument("--kernel-table", action="store_true", help="print the per-kernel time table to stderr")
    ap.add_argument("--force-dp", action="store_true", help="exercise the RCCL data-parallel path even with one rank")
    args = ap.parse_args()
    if args.serial_streams:
        os.environ["MI355_SIDE_STREAM"] = "0"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(sys.argv[1:], args.gpus)          # (does not return)

    t_start = time.perf_counter()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the result line would be mislabelled")
    # stdout carries exactly ONE line (the JSON, rank 0): libraries that print banners to fd 1 (RCCL prints "Hostname : ..." /
    # "Librccl path : ..." when the communicator is created) are pointed at stderr for the rest of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    use_dp = world > 1 or args.force_dp
    if use_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=device)

    from mi355 import amp as mamp, nn as mnn, optim as moptim
    from mi355.dp import DataParallel
    from utils.helpers import get_class_model, get_seg_model

    torch.manual_seed(0)
    seg_names = {"AttentionUNet": "attentionunet", "R2AttU_Net": "r2attunet", "R2U_Net": "r2unet", "ResNetUnet": "resnetunet"}
    seg = args.model in seg_names
    model = get_seg_model(seg_names[args.model]) if seg else get_class_model(args.model)[0]      # (classifier: stage-2, all layers train)
    model.compute_dtype = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[args.dtype]
    model = model.to(device).train()
    model.engine._check_storage()
    dp = DataParallel(model, force=args.force_dp) if use_dp else None
    crit = mnn.BCEWithLogitsLoss() if seg else mnn.CrossEntropyLoss(label_smoothing=0.1)
    opt = moptim.AdamW(model.parameters(), lr=1e-6, weight_decay=5e-4)
    inv_scale = dp.inv_scale if dp is not None else 1.0
    opt.inv_scale = inv_scale          # gradient averaging over ranks is folded into clip + AdamW
    scaler = mamp.GradScaler(enabled=args.dtype == "fp16")      # helpers.py:285,323-336: part of the fp16 step
    x, y = make_batch(args.batch, args.size, seed=rank, device=device)
    if not seg:
        y = torch.randint(0, 3, (args.batch,), generator=torch.Generator().manual_seed(rank)).to(device)

    def step():
        opt.zero_grad(set_to_none=True)
        out = model(x)
        loss = crit(out, y)
        scaler.scale(loss).backward()
        scaler.unscale_(opt)
        moptim.clip_grad_norm_(model.parameters(), max_norm=1.0, inv_scale=inv_scale)
        scaler.step(opt)
        scaler.update()
        return loss

    def log(msg):
        if rank == 0:
            print(f"[bench +{time.perf_counter() - t_start:7.1f}s] {msg}", file=sys.stderr, flush=True)

    log("model / optimizer / batch ready; first step (builds the launch plan)")
    step()
    torch.cuda.synchronize()
    log(f"first step done; plan launches fwd/bwd = {[p.n_launches for p in model.engine.plans.values()]}")
    runner = step
    if args.graph:
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            static_loss = step()
        runner = lambda: (g.replay(), static_loss)[1]

    for _ in range(args.warmup):
        loss = runner()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = runner()
    issue_ms = (time.perf_counter() - t0) / args.steps * 1e3      # host time to ISSUE a step (== ms_per_step when launch-bound)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    final_loss = float(loss.detach())
    # host cost of ISSUING one step, measured on an empty queue (over many back-to-back steps the runtime's bounded launch queue
    # makes the host wait for the GPU, so `issue_ms` above converges to the GPU time and says nothing about the host)
    one = []
    for _ in range(3):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        runner()
        one.append((time.perf_counter() - t1) * 1e3)
    torch.cuda.synchronize()
    issue_one_ms = min(one)
    log(f"timed region done: {elapsed / args.steps * 1e3:.2f} ms/step (host time to issue ONE step on an idle queue "
        f"{issue_one_ms:.2f} ms; {issue_ms:.2f} ms/step over the back-to-back region, bounded by the launch queue)")

    result = None
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = args.batch * world * args.steps / elapsed
        result = {
            "metric": f"images/sec (train step) {'Attention U-Net' if args.model == 'AttentionUNet' else args.model} "
                      f"{args.size}x{args.size} bs={args.batch}/GPU",
            "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{args.model} {args.size}x{args.size} train step (fwd+{'BCE' if seg else 'CE'}+bwd+"
                                   f"{'unscale+' if args.dtype == 'fp16' else ''}clip+AdamW), "
                                   f"bs={args.batch}/GPU, NHWC {args.dtype} activations, fp32 master weights",
                       "global_batch": args.batch * world, "parallelism": f"dp{world}", "hip_graph": bool(args.graph),
                       "wgrad_side_stream": os.environ.get("MI355_SIDE_STREAM", "1") != "0",
                       "final_loss": round(final_loss, 5), "host_issue_ms_per_step": round(issue_one_ms, 3),
                       "plan_replay": "C (mi355_plan_run)" if os.environ.get("MI355_PLAN_C", "1") != "0" else "python loop"},
        }
        if dp is not None:                        # fp32 = parity mode (default); MI355_DP_BUCKET_DTYPE=bf16: two-byte buckets on the wire
            result["config"]["grad_bucket_dtype"] = {None: "fp32", torch.bfloat16: "bf16", torch.float16: "fp16"}[dp.wire_dtype]
        if args.model == "AttentionUNet":         # (SURVEY.md 8d gives the per-image FLOPs of this model only)
            step_tflops = TRAIN_GFLOP_PER_IMG * (args.size / 256) ** 2 * value / world / 1e3
            result["config"]["step_mfma_frac"] = round(step_tflops / PEAK_TFLOPS[args.dtype], 4)

    # ---- per-kernel roofline (instrumented replay of the same plan, rank 0) -----------------------------
    if rank == 0 and not args.no_profile:
        plan = [p for p in model.engine.plans.values() if p.training and p.dout is not None][0]
        global SPLIT_BY_SHAPE
        SPLIT_BY_SHAPE = args.split_by_shape
        agg = profile_plan(plan, x, torch.cuda.current_stream().cuda_stream)
        # Kernels are ranked by CHIP-TIME (duration x share of the CUs the launch is sized for, Launch.cus): every launch but the
        # eight-wave weight gradient spans the chip; that one is launched on 128 CU-owning workgroups so that the main stream keeps
        # the other 128 CUs (csrc/conv_wgrad.hip), and its fraction of peak is quoted for the whole chip AND for the CUs it holds.
        total = sum(v[4] for v in agg.values())
        ranked = sorted(agg.items(), key=lambda kv: -kv[1][4])
        if args.kernel_table:
            print(f"{'kernel / launcher':58s} {'ms/step':>9s} {'launches':>8s} {'TFLOP/s':>9s} {'GB/s':>8s} {'CUs':>5s} {'share':>6s}", file=sys.stderr)
            reps = 2
            for k, (ms_, n, fl, nb_, chip) in ranked[:args.table_rows]:
                tf = fl / (ms_ * 1e-3) / 1e12 if fl else 0.0
                gbs = nb_ / (ms_ * 1e-3) / 1e9 if nb_ else 0.0      # algorithmic bytes / time
                print(f"{k:58s} {ms_ / reps:9.3f} {n // reps:8d} {tf:9.1f} {gbs:8.0f} {chip / ms_:5.2f} {chip / total:6.1%}", file=sys.stderr)
            print(f"{'sum of plan launches (ms; chip-time ms)':58s} {sum(v[0] for v in agg.values()) / reps:9.3f} {total / reps:9.3f}", file=sys.stderr)

        def mfma_entry(k, ms_, n, fl, nb, chip):
            ach = fl / (ms_ * 1e-3) / 1e12
            peak = PEAK_TFLOPS[args.dtype]
            cus = chip / ms_
            e = {"kernel": k, "bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4)}
            if cus < 0.999:      # sized for a share of the chip by design: the fraction of what THOSE CUs can deliver beside it
                e.update({"cus_share": round(cus, 3), "frac_of_cus_used": round(ach / (peak * cus), 4)})
            return e

        k, (ms_, n, fl, nb, chip) = ranked[0]
        traffic, traffic_src = pmc_traffic(k)
        common = {"traffic": traffic, "traffic_unit": "bytes/launch (HBM, PMC)", "traffic_source": traffic_src,
                  "launch_timing": "HIP events around every launch of a single-stream replay of the plan (the production step "
                                   "overlaps weight-gradient kernels on a side stream); kernels ranked by chip-time = duration x "
                                   "share of the CUs the launch is sized for",
                  "launches_per_step": n // 2, "avg_launch_ms": round(ms_ / n, 4), "share_of_plan_time": round(chip / total, 3)}
        if fl:
            result["roofline"] = {**mfma_entry(k, ms_, n, fl, nb, chip), **common, "algorithmic_bytes_per_launch": int(nb / n),
                                  "algorithmic_gflop_per_launch": round(fl / n / 1e9, 3)}
        else:
            gbs = nb / (ms_ * 1e-3) / 1e9 if nb else None      # algorithmic bytes (each operand once) / launch time
            result["roofline"] = {"bound": "hbm", "kernel": k, "achieved": round(gbs, 1) if gbs else None, "peak": 8000.0, "unit": "GB/s",
                                  "frac": round(gbs / 8000.0, 4) if gbs else None, **common,
                                  "algorithmic_bytes_per_launch": int(nb / n) if nb else None}
        # the kernels behind the dominant one (same measurement): the step is not one kernel
        also = []
        for k2, (ms2, n2, fl2, nb2, chip2) in ranked[1:8]:
            t2, _ = pmc_traffic(k2)          # HBM bytes per launch from the same PMC file as the dominant kernel's (None: not in it)
            extra = {"launches_per_step": n2 // 2, "share_of_plan_time": round(chip2 / total, 3), "avg_launch_ms": round(ms2 / n2, 4),
                     "traffic": t2, "algorithmic_bytes_per_launch": int(nb2 / n2) if nb2 else None}
            if fl2:
                also.append({**mfma_entry(k2, ms2, n2, fl2, nb2, chip2), **extra})
            elif nb2:
                a2 = nb2 / (ms2 * 1e-3) / 1e9
                also.append({"kernel": k2, "bound": "hbm", "achieved": round(a2, 1), "unit": "GB/s", "frac": round(a2 / 8000.0, 4), **extra})
        result["roofline"]["next_kernels"] = also
        result["roofline"]["plan_kernel_ms_per_step"] = round(sum(v[0] for v in agg.values()) / 2, 3)
        result["roofline"]["plan_chip_ms_per_step"] = round(total / 2, 3)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        log("per-kernel profile done; timing the CPU oracle baseline")
        if args.model in CPU_CONFIGS:
            result["cpu_baseline"] = cpu_baseline(args.model, args.size)
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(result) + "\n").encode())
    if use_dp:
        if world > 1:
            dist.barrier()      # rank 0 has been profiling on its own: tear the communicator down together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
