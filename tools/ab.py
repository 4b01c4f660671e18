#!/usr/bin/env python3
"""A/B two builds of the library on one workload (separate processes would add variance: each build is
loaded in its own subprocess but timed with the same script; use for coarse differences only)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, torch
sys.path.insert(0, %r)
from flex_amd import binding
binding._SO = sys.argv[1]
import flex_amd
import tools._knobs  # noqa: F401  (FLEX_* environment knobs -> plan descriptor)
name, k, order = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
a = flex_amd.synth_graph(name)
B = torch.rand((a.n, k), device="cuda") * 2 - 1
C = torch.empty((a.m, k), device="cuda")
p = flex_amd.Plan(a, k, order=order)
s = torch.cuda.current_stream().cuda_stream
best = 1e9
for rnd in range(5):
    for _ in range(5): p.spmm(B.data_ptr(), C.data_ptr(), s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): p.spmm(B.data_ptr(), C.data_ptr(), s)
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 50 * 1e3)
print(f"{os.path.basename(sys.argv[1]):28s} {name} k={k} order={order} wave_nnz={os.environ.get('FLEX_WAVE_NNZ','dflt')} t={best:8.1f} us  GFLOPS={2*a.nnz*k/best/1e3:8.1f}")
''' % ROOT
libs = sys.argv[1].split(",")
name, k, order = sys.argv[2], sys.argv[3], sys.argv[4]
for rep in range(2):
    for lib in libs:
        subprocess.run([sys.executable, "-c", code, os.path.join(ROOT, "flex_amd", "lib", lib), name, k, order], check=True)
