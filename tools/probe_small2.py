#!/usr/bin/env python3
"""second sweep on pubmed.csv: column-tile width and XCD-slice padding on a graph that does not fill the chip"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flex_amd  # noqa: E402
from tools._timing import timeit  # noqa: E402

a = flex_amd.csv_load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "pubmed.csv"))
for k in (128, 32):
    B = torch.rand((a.n, k), device="cuda") * 2 - 1
    C = torch.empty((a.m, k), device="cuda")
    for lanes in ((32, 16, 8) if k == 128 else (8,)):
        for budget in (48, 64, 96, 128, 192):
            for xcd in (0, 2):
                knobs = {"chunk_records": budget, "long_row": 4 * budget, "lanes_per_nz": lanes, "xcd_slices": xcd}
                p = flex_amd.Plan(a, k, order=flex_amd.FLEX_ORDER_CLUSTER, tuning=knobs)
                t = timeit(p, B, C, 200, rounds=5, warm=20)
                i = p.info()
                print(f"k={k} G={lanes:2d} budget {budget:4d} xcd_slices={xcd}: {t:6.2f} us  chunks {i['n_chunks']:5d} slots {i['n_slots']:5d} split rows {i['n_split_rows']:4d}", flush=True)
                p.destroy()
