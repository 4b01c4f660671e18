#!/usr/bin/env python3
"""Timing-only ablations (outputs are wrong by design): which part of the flickr launch is not the B gather?"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, %r)
from flex_amd import binding
binding._SO = sys.argv[1]
import flex_amd
import tools._knobs  # noqa: F401  (FLEX_* environment knobs -> plan descriptor)
name, k, fold = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
a = flex_amd.synth_graph(name)
if fold: a = flex_amd.HostCsr(a.rowPtr, (a.col %% fold).astype(np.uint32), a.vals, n=a.n)
B = torch.rand((a.n, k), device="cuda") * 2 - 1
C = torch.empty((a.m, k), device="cuda")
p = flex_amd.Plan(a, k, order=2)
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
print(f"{os.path.basename(sys.argv[1]):26s} {name} k={k} fold={fold:6d}: {best:7.1f} us")
''' % ROOT
for lib in sys.argv[1].split(","):
    for fold in (1024, 0):
        subprocess.run([sys.executable, "-c", code, os.path.join(ROOT, "flex_amd", "lib", lib), sys.argv[2], sys.argv[3], str(fold)], check=True, timeout=90)
