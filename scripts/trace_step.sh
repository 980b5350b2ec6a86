#!/bin/bash
# kernel trace (start / end timestamps) of the production two-stream schedule: which launches overlap, and what it costs them
set -e -o pipefail
tag=${1:-trace}
out=$PWD/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out/kt -- python3 bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-profile > $out/bench.json 2> $out/kt.log
f=$(find $out/kt -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $out/${tag}_step_timeline.txt <<'PY'
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last full step: find the last AdamW launch and the one before it
idx = [i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"]]
a, b = idx[-2] + 1, idx[-1] + 1
step = rows[a:b]
t0 = int(step[0]["Start_Timestamp"])
def short(n):
    n = re.sub(r"\(.*", "", n).replace("void ", "")
    return n[:58]
print(f"# one production step: {len(step)} dispatches, {(int(step[-1]['End_Timestamp']) - t0) / 1e6:.3f} ms; columns: start us, duration us, queue, kernel")
for r in step:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s / 1e3:10.1f} {(e - s) / 1e3:8.1f}  q{r['Queue_Id']:>3s}  {short(r['Kernel_Name'])}")
PY
tail -3 $out/bench.json | cut -c1-200
wc -l $out/${tag}_step_timeline.txt
