#!/usr/bin/env python3
"""tools/pmc_by_variant.py <counter_collection.csv> <kernel substring> <dispatches per variant> -- mean of every counter per
consecutive group of dispatches of one kernel (tools/probe_blocks.py launches each variant 3 x (3 + reps) times)."""
import collections
import csv
import sys

path, sub, per = sys.argv[1], sys.argv[2], int(sys.argv[3])
rows = [r for r in csv.DictReader(open(path)) if sub in r["Kernel_Name"]]
by_counter = collections.defaultdict(list)
for r in rows:
    by_counter[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
for c, v in by_counter.items():
    v.sort()
    vals = [x for _, x in v]
    groups = [vals[i:i + per] for i in range(0, len(vals), per)]
    print(c, " | ".join(f"{sum(g) / len(g):.4g} (n={len(g)})" for g in groups))
