#!/usr/bin/env python3
"""One very long row among many short ones: how does the split-row machinery scale with the number of pieces?"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flex_amd  # noqa: E402
import tools._knobs  # noqa: E402,F401  (FLEX_* environment knobs -> plan descriptor)
import oracle  # noqa: E402  (tool, not product: checks the result)

n, k = 1 << 20, 128
for hub in (10_000, 100_000, 1_000_000):
    deg = np.full(n, 4, dtype=np.int64)
    deg[7] = hub
    rp = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(deg, out=rp[1:])
    rng = np.random.default_rng(1)
    col = rng.integers(0, n, size=rp[-1], dtype=np.int64).astype(np.uint32)
    col[rp[7]:rp[8]] = rng.permutation(n)[:hub].astype(np.uint32)
    vals = rng.uniform(-1, 1, size=rp[-1]).astype(np.float32)
    a = flex_amd.HostCsr(rp.astype(np.uint32), col, vals, n=n)
    B = torch.rand((n, k), device="cuda") * 2 - 1
    C = torch.empty((n, k), device="cuda")
    p = flex_amd.Plan(a, k)
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        p.spmm(B.data_ptr(), C.data_ptr(), s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        p.spmm(B.data_ptr(), C.data_ptr(), s)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    Bh = B.cpu().numpy()
    gold_hub = (vals[rp[7]:rp[8], None].astype(np.float64) * Bh[col[rp[7]:rp[8]]].astype(np.float64)).sum(axis=0)
    err = np.abs(C[7].cpu().numpy() - gold_hub).max()
    info = p.info()
    print(f"hub={hub:8d}: {us:9.1f} us  pieces={info['n_partials']:6d} chunks={info['n_chunks']}  hub-row max err {err:.2e} (tol {4*1.19e-7*hub:.2e})", flush=True)
