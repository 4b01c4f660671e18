#!/usr/bin/env python3
"""tools/probe_rmat.py [scale=18] [edge factor=16] [k=128] -- a graph WITHOUT communities (R-MAT a/b/c = 0.57/0.19/0.19, symmetrised,
self loops, random relabel) under every schedule: what the community order and the guard of its second stage do there."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flex_amd  # noqa: E402
import tools._knobs  # noqa: E402,F401  (FLEX_* environment knobs -> plan descriptor)

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 18
ef = int(sys.argv[2]) if len(sys.argv) > 2 else 16
k = int(sys.argv[3]) if len(sys.argv) > 3 else 128
rng = np.random.default_rng(1)
n, m = 1 << scale, ef << scale
r = np.zeros(m, np.int64)
c = np.zeros(m, np.int64)
for lvl in range(scale):
    u = rng.random(m)
    r |= (u >= 0.76).astype(np.int64) << lvl
    c |= (((u >= 0.57) & (u < 0.76)) | (u >= 0.95)).astype(np.int64) << lvl
perm = rng.permutation(n)
key = np.unique(np.concatenate([perm[r] * n + perm[c], perm[c] * n + perm[r], np.arange(n) * (n + 1)]))
rr, cc = key // n, key % n
rp = np.zeros(n + 1, dtype=np.int64)
np.cumsum(np.bincount(rr, minlength=n), out=rp[1:])
a = flex_amd.HostCsr(rp.astype(np.uint32), cc.astype(np.uint32), rng.uniform(-1, 1, len(cc)).astype(np.float32), n=n)
print(f"R-MAT scale {scale}: n={a.n} nnz={a.nnz} max degree {int(np.diff(rp).max())}", flush=True)
B = torch.rand((a.n, k), device="cuda") * 2 - 1
C = torch.empty((a.m, k), device="cuda")
s = torch.cuda.current_stream().cuda_stream
for tag, order, env in (("natural", 0, {}), ("rcm", flex_amd.FLEX_ORDER_RCM, {}), ("cluster, walk only", flex_amd.FLEX_ORDER_CLUSTER, {"FLEX_CLUSTER_NO_REFINE": "1"}),
                        ("cluster (moves guarded)", flex_amd.FLEX_ORDER_CLUSTER, {})):
    os.environ.pop("FLEX_CLUSTER_NO_REFINE", None)
    os.environ.update(env)
    t0 = time.time()
    p = flex_amd.Plan(a, k, order=order)
    tp = time.time() - t0
    best = 1e9
    for _ in range(3):
        for _ in range(3):
            p.spmm(B.data_ptr(), C.data_ptr(), s)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            p.spmm(B.data_ptr(), C.data_ptr(), s)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 30 * 1e3)
    print(f"  {tag}: {best:.1f} us = {2 * a.nnz * k / best * 1e-3:.0f} GFLOPS (plan {tp:.2f} s)", flush=True)
    p.destroy()
