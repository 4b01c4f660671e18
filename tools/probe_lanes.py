#!/usr/bin/env python3
"""Where does a narrower column tile (fewer lanes per record, several k-tiles per launch) start to pay?
Reddit-shaped generator at varying average degree / n, k=128, cluster schedule; each G in its own
subprocess (FLEX_LANES is read at plan time).  Usage: python tools/probe_lanes.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, torch
sys.path.insert(0, %r)
import flex_amd
import tools._knobs  # noqa: F401  (FLEX_* environment knobs -> plan descriptor)
n, deg, k = int(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3])
p = flex_amd.synth_preset("reddit")
a = flex_amd.synth_graph(n=n, nnz=n + 2 * int(n * (deg - 1) / 2), alpha=p.alpha, community=p.community, p_in=p.p_in, p_near=p.p_near,
                         near_window=p.near_window, seed=7)
B = torch.rand((a.n, k), device="cuda") * 2 - 1
C = torch.empty((a.m, k), device="cuda")
pl = flex_amd.Plan(a, k, order=2)
s = torch.cuda.current_stream().cuda_stream
best = 1e9
for rnd in range(3):
    for _ in range(3): pl.spmm(B.data_ptr(), C.data_ptr(), s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): pl.spmm(B.data_ptr(), C.data_ptr(), s)
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
print(f"n={n:8d} deg={a.nnz/a.n:6.1f} k={k} G={pl.info()['lanes_per_nz']:2d} t={best:9.1f} us GFLOPS={2*a.nnz*k/best/1e3:8.1f}", flush=True)
''' % ROOT
for n, deg in ((232965, 12), (232965, 24), (232965, 36), (232965, 48), (232965, 64), (232965, 100), (60000, 100), (60000, 32), (1000000, 32), (1000000, 64)):
    for lanes in (32, 16, 8):
        env = dict(os.environ, FLEX_LANES=str(lanes))
        subprocess.run([sys.executable, "-c", code, str(n), str(deg), "128"], check=True, env=env, timeout=300)
