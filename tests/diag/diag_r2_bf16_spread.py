"""How far apart do EQUALLY VALID arithmetic orders leave the R2AttU_Net Dice of tests/test_gpu_train_bf16.py?  Runs
diag_r2_gpu_traj.py (HIP fp32 and bf16, Dice after the given steps, default 20 / 32 / 48) once per combination of the plan-level fusion switches:
each combination computes the same function with roundings in different places (a gradient summed before or after it is rounded to
bf16, a dot product with or without a fused multiply-add).   python tests/diag/diag_r2_bf16_spread.py"""
import itertools
import os
import subprocess
import sys

here = os.path.dirname(os.path.abspath(__file__))
switches = ("MI355_FUSE_GATE_BWD", "MI355_FUSE_RESIDUAL")      # (R2AttU_Net has no layer MI355_FUSE_POOL / _POOL_BWD apply to)
for combo in itertools.product("10", repeat=len(switches)):
    env = dict(os.environ, **dict(zip(switches, combo)))
    r = subprocess.run([sys.executable, os.path.join(here, "diag_r2_gpu_traj.py")] + (sys.argv[1:] or ["20", "32", "48"]), env=env, capture_output=True, text=True)
    tag = " ".join(f"{s[11:]}={v}" for s, v in zip(switches, combo))
    for line in r.stdout.splitlines():
        if line.startswith("HIP"):
            print(f"[{tag}] {line}", flush=True)
    if r.returncode:
        print(f"[{tag}] failed: {r.stderr[-400:]}", flush=True)
