#!/usr/bin/env python3
"""Summarise a tools/pmc.sh output directory: per-kernel average duration and counters per launch."""
import collections
import csv
import glob
import json
import re
import sys

out = sys.argv[1]


def kname(full):
    """'void flex::(anonymous namespace)::spmm_flat_kernel<16, true, 4, 4>(flex::PlanView, ...)' -> 'spmm_flat_kernel<16, true, 4, 4>'"""
    m = re.search(r"flex::(?:\(anonymous namespace\)::)?(\w+(?:<[^>]*>)?)", full)
    return m.group(1) if m else full


res = {}
for f in glob.glob(f"{out}/kt/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "flex::" in r["Name"]:
            name = kname(r["Name"])
            res.setdefault(name, {})["calls"] = int(r["Calls"])
            res[name]["avg_us"] = float(r["AverageNs"]) / 1e3
for d in ("pmc1", "pmc2", "pmc3"):
    for f in glob.glob(f"{out}/{d}/*/*_counter_collection.csv"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "flex::" in r["Kernel_Name"]:
                name = kname(r["Kernel_Name"])
                agg[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (name, c), v in agg.items():
            res.setdefault(name, {})[c] = sum(v) / len(v)
for name, d in res.items():
    if "TCC_HIT_sum" in d:
        d["l2_hit_rate"] = d["TCC_HIT_sum"] / max(1.0, d["TCC_HIT_sum"] + d["TCC_MISS_sum"])
    if "FETCH_SIZE" in d:  # KB; gfx950 tallies 128-B requests at 64 B for wide reads (MI355X_MICROARCH HBM)
        d["fetch_MB_raw"] = d["FETCH_SIZE"] / 1e3
        d["fetch_MB_x2"] = 2 * d["FETCH_SIZE"] / 1e3
    if "WRITE_SIZE" in d:
        d["write_MB"] = d["WRITE_SIZE"] / 1e3
# measured B reuse u (≙ flex.cu:5513-5528) when the bench line with n, nnz, k is in kt.log
try:
    cfg = None
    for line in open(f"{out}/kt.log"):
        if line.startswith("{") and '"config"' in line:
            cfg = json.loads(line)["config"]
    for name, d in res.items():
        if cfg and "fetch_MB_x2" in d and name.startswith("spmm_"):
            nnz, n, k = cfg["nnz"], cfg["n"], cfg["k"]
            d["u_l2"] = 4.0 * nnz * k / max(1.0, d["fetch_MB_x2"] * 1e6 - 8.0 * nnz - 4.0 * (n + 1))
except (OSError, KeyError, ValueError):
    pass
print(json.dumps(res, indent=1))
json.dump(res, open(f"{out}/summary.json", "w"), indent=1)
