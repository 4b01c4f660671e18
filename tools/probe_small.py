#!/usr/bin/env python3
"""tools/probe_small.py [graph.csv] -- the launch floor on a small graph (pubmed.csv: 19 717 rows, 108 365 nonzeros): launch time over
the chunk budget, the row-splitting threshold and the two forms of the split-row sum, k = 32 and 128 (verdict r02, item 7)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flex_amd  # noqa: E402
from tools._timing import timeit  # noqa: E402

path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "pubmed.csv")
a = flex_amd.csv_load(path)
for k in (32, 128):
    B = torch.rand((a.n, k), device="cuda") * 2 - 1
    C = torch.empty((a.m, k), device="cuda")
    base = flex_amd.Plan(a, k, order=flex_amd.FLEX_ORDER_CLUSTER)
    t0 = timeit(base, B, C, 200, rounds=5, warm=20)
    i = base.info()
    print(f"k={k} rule: {t0:6.2f} us  chunks {i['n_chunks']} slots {i['n_slots']} split rows {i['n_split_rows']} budget {base.tuning()['chunk_records']} long_row {base.tuning()['long_row']}", flush=True)
    for budget in (16, 24, 32, 48, 64, 96, 128):
        for long_row in (0, 4 * budget, 100000):
            for split in (2, 1):
                knobs = {"chunk_records": budget, "split_rows": split}
                if long_row:
                    knobs["long_row"] = long_row
                p = flex_amd.Plan(a, k, order=flex_amd.FLEX_ORDER_CLUSTER, tuning=knobs)
                t = timeit(p, B, C, 200, rounds=5, warm=20)
                i = p.info()
                print(f"k={k} budget {budget:4d} long_row {long_row or budget:6d} split_rows={split}: {t:6.2f} us  chunks {i['n_chunks']:5d} slots {i['n_slots']:5d} split rows {i['n_split_rows']:4d}", flush=True)
                p.destroy()
