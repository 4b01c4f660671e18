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
    # both counters are in KiB (counter_defs.yaml: .../1024; a 2 GiB copy reads FETCH_SIZE = 1 048 58x and WRITE_SIZE = 2 097 152:
    # tools/probe_counters.py); gfx950 tallies 128-B requests at 64 B for wide reads (MI355X_MICROARCH HBM)
    if "FETCH_SIZE" in d:
        d["fetch_MB_raw"] = d["FETCH_SIZE"] * 1024 / 1e6
        d["fetch_MB_x2"] = 2 * d["FETCH_SIZE"] * 1024 / 1e6
    if "WRITE_SIZE" in d:
        d["write_MB"] = d["WRITE_SIZE"] * 1024 / 1e6
# roofline of every SpMM kernel of the launch from THIS directory alone (verdict r02 item 6): algorithmic bytes of the workload
# (4(m+1) + 8 nnz + 4 n k + 4 m k, from the bench line in kt.log), the kernel's average duration from the kernel
# trace, the HBM-side traffic from the PMC passes (2 x FETCH_SIZE + WRITE_SIZE: the gfx950 correction of MI355X_MICROARCH.md)
# and the measured B reuse u (≙ flex.cu:5513-5528).  With several kernels per step (row blocks + flat rest, fix-up, MFMA tiles)
# the step's figures are in "step".
HBM_PEAK = 8.0e12
try:
    cfg = None
    for line in open(f"{out}/kt.log"):
        if line.startswith("{") and '"config"' in line:
            cfg = json.loads(line)["config"]
    if cfg:
        nnz, n, k = cfg["nnz"], cfg["n"], cfg["k"]
        b_alg = 4.0 * (n + 1) + 8.0 * nnz + 8.0 * n * k
        step = {"b_alg": int(b_alg), "us": 0.0, "traffic_bytes": 0.0}
        calls = [d.get("calls", 0) for name, d in res.items() if name.startswith("spmm_")]
        per_step = max(calls) if calls else 1
        for name, d in res.items():
            if not name.startswith("spmm_") or "avg_us" not in d:
                continue
            share = d.get("calls", per_step) / per_step  # the stamped twin of the imbalance report runs once per process
            d["b_alg"] = int(b_alg)
            d["frac"] = round(b_alg / (d["avg_us"] * 1e-6) / HBM_PEAK, 4)
            if "fetch_MB_x2" in d and "write_MB" in d:
                d["traffic_bytes"] = int((d["fetch_MB_x2"] + d["write_MB"]) * 1e6)
                d["traffic_over_b_alg"] = round(d["traffic_bytes"] / b_alg, 2)
                d["traffic_GBps"] = round(d["traffic_bytes"] / (d["avg_us"] * 1e-6) / 1e9, 1)
                d["u_l2"] = round(4.0 * nnz * k / max(1.0, d["fetch_MB_x2"] * 1e6 - 8.0 * nnz - 4.0 * (n + 1)), 3)
            if share > 0.5 and ", true>" not in name:  # kernels that run every step (not the once-per-process stamped twin)
                step["us"] += d["avg_us"]
                step["traffic_bytes"] += d.get("traffic_bytes", 0)
        if step["us"] > 0:
            step["frac"] = round(b_alg / (step["us"] * 1e-6) / HBM_PEAK, 4)
            step["traffic_over_b_alg"] = round(step["traffic_bytes"] / b_alg, 2) if step["traffic_bytes"] else None
            step["traffic_bytes"] = int(step["traffic_bytes"])
            step["workload"] = cfg.get("workload")
            res["step"] = step
except (OSError, KeyError, ValueError):
    pass
print(json.dumps(res, indent=1))
json.dump(res, open(f"{out}/summary.json", "w"), indent=1)
