#!/usr/bin/env python3
"""tools/sweep_gen_pmc.py <dir> -- join tools/sweep_gen.py's timing lines (<dir>/points.jsonl) with the per-dispatch
counters of three rocprofv3 --pmc passes of the same script (<dir>/pmc{1,2,3}): the k-th group of 7 spmm_flat_kernel
dispatches belongs to the k-th point; the 5 timed ones are averaged.  Prints one line per point."""
import csv
import glob
import json
import sys

d = sys.argv[1]
pts = [json.loads(ln) for ln in open(f"{d}/points.jsonl") if ln.startswith("{")]
ctr = {}
for sub in ("pmc1", "pmc2", "pmc3"):
    for f in glob.glob(f"{d}/{sub}/*/*_counter_collection.csv"):
        per = {}
        for r in csv.DictReader(open(f)):
            if "spmm_flat_kernel" in r["Kernel_Name"]:
                per.setdefault(r["Counter_Name"], {}).setdefault(int(r["Dispatch_Id"]), 0.0)
                per[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
        for name, disp in per.items():
            ctr[name] = [v for _, v in sorted(disp.items())]
out = []
for i, p in enumerate(pts):
    row = dict(p)
    sl = slice(7 * i + 2, 7 * i + 7)
    g = {n: (sum(v[sl]) / max(1, len(v[sl])) if len(v) >= 7 * (i + 1) else None) for n, v in ctr.items()}
    if g.get("TCC_HIT_sum") is not None:
        row["l2_hit"] = round(g["TCC_HIT_sum"] / (g["TCC_HIT_sum"] + g["TCC_MISS_sum"]), 3)
    if g.get("FETCH_SIZE") is not None and g.get("WRITE_SIZE") is not None:
        row["traffic_MB"] = round((2 * g["FETCH_SIZE"] + g["WRITE_SIZE"]) * 1024 / 1e6, 1)  # KiB; FETCH doubled (gfx950)
        row["traffic_over_alg"] = round(row["traffic_MB"] / p["b_alg_MB"], 2)
    out.append(row)
    print(json.dumps(row))
json.dump(out, open(f"{d}/sweep_joined.json", "w"), indent=1)
