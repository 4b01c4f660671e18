#!/usr/bin/env python3
"""tools/probe_bound.py [workload k ...] -- the all-L2-hit time of the product kernel (verdict r02 item 1).

The matrix is put in community order once (flex_order_cluster + flex_perm_csr) and planned in that order, so every variant below
has the SAME schedule, chunks, row lengths and record stream; only the B row a record points at changes: col % T folds B onto a
table of T rows -- T = 1024 (one column tile of it is L2-resident on every XCD), 8192, 65536 (Infinity-Cache-resident) and n
(unfolded: the real launch).  Printed: launch time and the rate at which the demanded B bytes (records x 4k) move."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flex_amd  # noqa: E402
import tools._knobs  # noqa: E402,F401  (FLEX_* environment knobs -> plan descriptor)
from tools._timing import timeit  # noqa: E402

args = sys.argv[1:] or ["reddit", "128", "amazon", "128"]
for name, k in zip(args[0::2], (int(x) for x in args[1::2])):
    a = flex_amd.synth_graph(name)
    rank = flex_amd.order_cluster(a)
    _, ap = flex_amd.perm_csr(a, rank)
    del a
    B = torch.rand((ap.n, k), device="cuda") * 2 - 1
    C = torch.empty((ap.m, k), device="cuda")
    for T in (1024, 8192, 65536, ap.n):
        b = ap if T >= ap.n else flex_amd.HostCsr(ap.rowPtr, (ap.col % T).astype(np.uint32), ap.vals, n=ap.n)
        p = flex_amd.Plan(b, k)
        i = p.info()
        t = timeit(p, B, C, 10 if ap.nnz > 1e8 else 30)
        p.destroy()
        print(f"{name} k={k} G={i['lanes_per_nz']} B folded to {T:8d} rows ({T * k * 4 / 1e6:8.1f} MB, one tile {T * 16 * i['lanes_per_nz'] / 1e6:7.2f} MB): "
              f"{t:8.1f} us   demanded B bytes {ap.nnz * k * 4 / 1e9:6.1f} GB at {ap.nnz * k * 4 / t / 1e6:6.1f} TB/s", flush=True)
