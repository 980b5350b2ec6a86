"""effective clock per kernel = GRBM_GUI_ACTIVE / 8 / duration (MI355X_MICROARCH.md, DVFS give-back).
The counter also runs over the dispatch's launch / drain time, which the kernel timestamps exclude: the quotient READS HIGH on
dispatches shorter than about 0.3 ms (it has shown 2.7-3.5 GHz on 50-us streaming kernels of a chip that tops out at 2.4) and
is within 3 % of the in-kernel clock only from about 10 ms up.  Rows are marked accordingly; quote clocks from long dispatches
(scripts/clock_long.sh: the same kernels on batches large enough for 1-10 ms launches)."""
import csv, glob, sys, collections, re
d = sys.argv[1]
dur = {}
for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r['Dispatch_Id']] = (re.sub(r"\(.*", "", r['Kernel_Name'])[:60], int(r['End_Timestamp']) - int(r['Start_Timestamp']))
agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != 'GRBM_GUI_ACTIVE': continue
        k, ns = dur[r['Dispatch_Id']]
        a = agg[k]; a[0] += float(r['Counter_Value']); a[1] += ns; a[2] += 1
for k, (c, ns, n) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:10]:
    us = ns / n / 1e3
    note = "" if us >= 3000 else ("  (~: dispatch < 3 ms, reads a few % high)" if us >= 300 else "  (UNRELIABLE: dispatch < 0.3 ms, reads high)")
    print(f"{k:60s} n={n:4d} avg {us:8.1f} us  clock {c/8/ns:6.3f} GHz{note}")
