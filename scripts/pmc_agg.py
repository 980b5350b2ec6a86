"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel: sum of each counter + dispatch count."""
import csv, sys, collections, glob, re
files = sys.argv[1:]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for pat in files:
    for f in glob.glob(pat, recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"\(.*", "", r["Kernel_Name"])[:70]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == sorted(agg[k])[0]: cnt[k] += 1
names = sorted({c for v in agg.values() for c in v})
print("kernel".ljust(70), "n".rjust(6), " ".join(n[-22:].rjust(22) for n in names))
key = names[0]
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", kv[1].get(key, 0)))[:18]:
    print(k.ljust(70), str(cnt[k]).rjust(6), " ".join(f"{v.get(n, 0):22.4g}" for n in names))
