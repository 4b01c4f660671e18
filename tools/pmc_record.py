#!/usr/bin/env python3
"""tools/pmc_record.py <gpurun_out/tag> <workload> <k> <order> -- after tools/pmc.sh: copy the summaries into profiles/
(rNN_<workload>_k<k>_*) and record the launch's HBM-side bytes with the fingerprint of the sources they were measured on
(bench.py reports `roofline.traffic` only while that fingerprint matches the running sources)."""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

out, workload, k, order = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
rnd = os.environ.get("FLEX_ROUND", "r03")
summ = json.load(open(os.path.join(out, "summary.json")))
main = max((n for n in summ if n.startswith("spmm_") and "FETCH_SIZE" in summ[n]), key=lambda n: summ[n].get("avg_us", 0) * summ[n].get("calls", 1))
d = summ[main]
traffic = int(2 * d["FETCH_SIZE"] * 1024 + d["WRITE_SIZE"] * 1024)  # KiB -> B; FETCH_SIZE doubled (gfx950: 128-B requests tallied at 64 B)
tag = f"{rnd}_{workload}_k{k}"
shutil.copy(os.path.join(out, "summary.json"), os.path.join(ROOT, "profiles", f"{tag}_pmc_summary.json"))
for f in glob.glob(os.path.join(out, "kt", "*", "*_kernel_stats.csv")):
    shutil.copy(f, os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"))
p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
t = json.load(open(p)) if os.path.exists(p) else {}
t[f"{workload}_k{k}_{order}_n1"] = {"bytes": traffic, "source_hash": bench.traffic_source_hash(), "kernel": main,
                                   "avg_us": round(d.get("avg_us", 0.0), 3), "l2_hit_rate": round(d.get("l2_hit_rate", 0.0), 4),
                                   "profiled": rnd}
json.dump(t, open(p, "w"), indent=1)
print(tag, main, "traffic", traffic, "avg_us", d.get("avg_us"), "hit", d.get("l2_hit_rate"))
