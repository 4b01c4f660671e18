#!/usr/bin/env python3
"""Is the SpMM kernel memory-bound?  Fold the column indices of the flickr stand-in onto a table of
T rows (col % T): T=1024 (512 KB of B: L2-resident), 8192 (4 MB), 65536 (32 MB: Infinity Cache), n (the real thing)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flex_amd  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "flickr"
k = int(sys.argv[2]) if len(sys.argv) > 2 else 128
a = flex_amd.synth_graph(name)
B = torch.rand((a.n, k), device="cuda") * 2 - 1
C = torch.empty((a.m, k), device="cuda")
s = torch.cuda.current_stream().cuda_stream
for T in (1024, 8192, 65536, a.n):
    b = flex_amd.HostCsr(a.rowPtr, (a.col % T).astype(np.uint32), a.vals, n=a.n)
    for order in (0, 2):
        p = flex_amd.Plan(b if T < a.n else a, k, order=order)
        best = 1e9
        for rnd in range(3):
            for _ in range(5):
                p.spmm(B.data_ptr(), C.data_ptr(), s)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                p.spmm(B.data_ptr(), C.data_ptr(), s)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 50 * 1e3)
        print(f"{name} k={k} B rows folded to {T:7d} ({T*k*4/1e6:7.1f} MB) order={'clu' if order else 'nat'}: {best:8.1f} us  gather {a.nnz*k*4/best/1e3:8.0f} GB/s")
