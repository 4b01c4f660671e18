#!/usr/bin/env python3
"""What one rank of an N-GPU weak-scaling run sees, measured on ONE GPU: shard r of the flickr x N graph with the
full (N x larger) B resident.  Predicts per-GPU step time at N = 1, 2, 4, 8 (no collective is on the data path)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flex_amd
import tools._knobs  # noqa: F401  (FLEX_* environment knobs -> plan descriptor)
name = sys.argv[1] if len(sys.argv) > 1 else "flickr"
strong = len(sys.argv) > 2 and sys.argv[2] == "strong"  # fixed graph, rows/N per GPU (the north_star's Amazon run)
k = int(sys.argv[3]) if len(sys.argv) > 3 else 128
a1 = flex_amd.synth_graph(name) if strong else None
for N in (1, 2, 4, 8):
    a = a1 if strong else flex_amd.synth_graph(name, scale=N)
    B = torch.rand((a.n, k), device="cuda") * 2 - 1
    ts = []
    for r in sorted({0, N // 2, N - 1}):
        if N == 1:
            p = flex_amd.Plan(a, k, order=flex_amd.FLEX_ORDER_CLUSTER); rows = a.m; nnz = a.nnz
        else:
            sh = flex_amd.make_shard(a, k, r, N, order="cluster"); p = sh.plan(k, 0); rows = sh.r1 - sh.r0; nnz = sh.nnz
        C = torch.empty((rows, k), device="cuda")
        s = torch.cuda.current_stream().cuda_stream
        best = 1e9
        for rnd in range(3):
            for _ in range(5): p.spmm(B.data_ptr(), C.data_ptr(), s)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): p.spmm(B.data_ptr(), C.data_ptr(), s)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 50 * 1e3)
        ts.append((r, rows, nnz, best))
    worst = max(t[3] for t in ts)
    print(f"{name} k={k} {'strong /' if strong else 'x'}{N}: n={a.n} nnz={a.nnz} B={a.n*k*4/1e6:.0f} MB; shards " +
          ", ".join(f"r{r}: {rows} rows {nnz} nnz {t:.1f}us" for r, rows, nnz, t in ts) +
          f"  -> predicted aggregate {2*a.nnz*k/worst/1e3:.0f} GFLOPS")
