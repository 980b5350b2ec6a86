"""HBM-side bytes per launch of each conv kernel from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; KiB per
dispatch) -> profiles/<round>_pmc_traffic.json, the file bench.py reads for roofline.traffic.  FETCH_SIZE is doubled:
gfx950 tallies the 128-B requests of wide coalesced reads at 64 B (MI355X_MICROARCH.md, HBM section).
usage: pmc_traffic.py <fetch csv glob> <write csv glob> <out.json>"""
import collections, csv, glob, json, re, sys

def demangle(name):
    """Itanium names of the conv kernel templates (`_Z<len><name>I<args>E...`; ROCm 7.2 ships no demangler that knows
    DF16b = __bf16): -> name<bf16,128,8,32>; anything else is returned unchanged."""
    m = re.match(r"_Z(\d+)", name)
    if not m:
        return name
    n = int(m.group(1))
    base, rest = name[m.end():m.end() + n], name[m.end() + n:]
    if not rest.startswith("I"):
        return base
    args, i = [], 1
    while i < len(rest) and rest[i] != "E":
        if rest.startswith("DF16b", i):
            args.append("bf16"); i += 5
        elif rest.startswith("DF16_", i):
            args.append("f16"); i += 5
        elif rest[i] == "f":
            args.append("f32"); i += 1
        elif rest[i] == "L":
            mm = re.match(r"L[a-z](n?\d+)E", rest[i:])
            if not mm:
                return name
            args.append(mm.group(1)); i += mm.end()
        else:
            return name
    return f"{base}<{','.join(args)}>"


# streaming kernels are templates over an Op type: bench.py tags them by their launcher
OP_TAGS = (("BnBwdReduceOp", "mi355_bn_bwd_reduce"), ("BnBwdApplyOp", "mi355_bn_bwd_apply"), ("BnActOp", "mi355_bn_act"),
           ("bn_act_pool2_kernel", "mi355_bn_act_pool2"), ("BnStatsOp", "mi355_bn_stats"),
           ("BnBwdReducePool2Op", "mi355_bn_bwd_reduce_pool2"), ("BnBwdApplyPool2Op", "mi355_bn_bwd_apply_pool2"),
           ("GateBnBwdReduceOp", "mi355_gate_bn_bwd_reduce"), ("GateBnBwdApplyOp", "mi355_gate_bn_bwd_apply"))


def per_kernel(pattern, counter):
    tot, cnt = collections.defaultdict(float), collections.Counter()
    for f in glob.glob(pattern, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            raw = r["Kernel_Name"]
            op = [t for o, t in OP_TAGS if o in raw]
            k = op[0] if op else re.sub(r"\(.*", "", demangle(raw)).replace("void ", "").strip()
            k = k.replace("__hip_bfloat16", "bf16").replace("__bf16", "bf16").replace("_Float16", "f16").replace(" ", "")
            tot[k] += float(r["Counter_Value"])
            cnt[k] += 1
    return tot, cnt


fetch, nf = per_kernel(sys.argv[1], "FETCH_SIZE")
write, nw = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes of bench.py --steps 2 "
                 "--warmup 1 --no-cpu-baseline --no-profile), counters in KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md "
                 "(gfx950 reports 1/2 of wide coalesced reads)", "kernels": {}}
for k in sorted(fetch, key=lambda k: -(2 * fetch[k] + write.get(k, 0))):
    if "conv" not in k and "wgrad" not in k and not k.startswith("mi355_bn_"):
        continue
    f = 2 * fetch[k] * 1024 / nf[k] / 1e6
    w = write.get(k, 0) * 1024 / max(nw.get(k, 1), 1) / 1e6
    # bench.py tags drop the element type of the templated kernels
    tag = re.sub(r"<bf16,|<f16,", "<", k) if ("halo" in k or "dma" in k or "stream" in k or "ws_kernel" in k) else k
    tag = tag.replace("<bf16>", "").replace("<f16>", "")
    if tag.startswith("wgrad3x3_halo_kernel"):      # (both W16 instantiations of the nine-tap weight gradient: one bench tag)
        tag = "wgrad3x3_halo_kernel"
    if tag.startswith("wgrad3x3_halo8_kernel"):     # (... and both W32 instantiations of its eight-wave form)
        tag = "wgrad3x3_halo8_kernel"
    e = {"launches_profiled": nf[k], "fetch_MB_per_launch": round(f, 1), "write_MB_per_launch": round(w, 1),
         "hbm_MB_per_launch": round(f + w, 1)}
    if tag in out["kernels"]:                      # two instantiations under one bench tag: launch-weighted mean
        o = out["kernels"][tag]
        n0, n1 = o["launches_profiled"], e["launches_profiled"]
        e = {"launches_profiled": n0 + n1, **{q: round((o[q] * n0 + e[q] * n1) / (n0 + n1), 1)
                                              for q in ("fetch_MB_per_launch", "write_MB_per_launch", "hbm_MB_per_launch")}}
    out["kernels"][tag] = e
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1)[:1500])
