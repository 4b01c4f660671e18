#!/usr/bin/env python3
"""k not a multiple of 32: dense storage (ld = k) against rows padded to whole cache lines (ld = 128)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flex_amd  # noqa: E402
import tools._knobs  # noqa: E402,F401  (FLEX_* environment knobs -> plan descriptor)

for name in ("flickr", "reddit"):
    a = flex_amd.synth_graph(name)
    for k, ld in ((100, 100), (100, 128), (104, 104), (104, 128), (128, 128), (44, 44), (44, 64), (64, 64)):
        B = torch.rand((a.n, ld), device="cuda") * 2 - 1
        C = torch.empty((a.m, ld), device="cuda")
        p = flex_amd.Plan(a, k, order=2, ldb=ld, ldc=ld)
        s = torch.cuda.current_stream().cuda_stream
        best = 1e9
        for rnd in range(3):
            for _ in range(5):
                p.spmm(B.data_ptr(), C.data_ptr(), s)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                p.spmm(B.data_ptr(), C.data_ptr(), s)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 30 * 1e3)
        print(f"{name:7s} k={k:4d} ld={ld:4d} t={best:8.1f} us GFLOPS={2 * a.nnz * k / best / 1e3:8.1f}", flush=True)
