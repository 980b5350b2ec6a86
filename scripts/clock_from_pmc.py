"""effective clock per kernel = GRBM_GUI_ACTIVE / 8 / duration (MI355X_MICROARCH.md, DVFS give-back)."""
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
    print(f"{k:60s} n={n:4d} avg {ns/n/1e3:8.1f} us  clock {c/8/ns:6.3f} GHz")
