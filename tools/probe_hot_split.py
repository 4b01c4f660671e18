#!/usr/bin/env python3
"""tools/probe_hot_split.py [workload k ...] -- what could LDS-level reuse of B buy?  (verdict r02, item 2)

The matrix is put in community order (flex_order_cluster + flex_perm_csr); a ROW BLOCK is R consecutive rows of that
order.  A column is HOT in a row block when at least `thr` of the block's nonzeros use it: those are the B rows a
workgroup that owns the block could stage in LDS once and use `u` times.  Printed per (R, thr): the share of the
nonzeros that are hot, u = hot nonzeros / hot (block, column) pairs, the mean number of hot columns per block (LDS
rows needed), and the launch time of the flat kernel on (a) the whole matrix, (b) the matrix WITHOUT its hot nonzeros
-- the part any hybrid would still send through the L2 gather path, i.e. a lower bound of the hybrid's time -- and
(c) the hot nonzeros alone."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flex_amd  # noqa: E402
import tools._knobs  # noqa: E402,F401  (FLEX_* environment knobs -> plan descriptor)
from tools._timing import timeit  # noqa: E402


def sub_csr(ap, keep):
    rows = np.repeat(np.arange(ap.m, dtype=np.int64), np.diff(ap.rowPtr.astype(np.int64)))
    cnt = np.bincount(rows[keep], minlength=ap.m)
    rp = np.zeros(ap.m + 1, dtype=np.int64)
    np.cumsum(cnt, out=rp[1:])
    return flex_amd.HostCsr(rp.astype(np.uint32), ap.col[keep], ap.vals[keep], n=ap.n)


args = sys.argv[1:] or ["reddit", "128", "amazon", "128"]
for name, k in zip(args[0::2], (int(x) for x in args[1::2])):
    t0 = time.time()
    a = flex_amd.synth_graph(name)
    rank = flex_amd.order_cluster(a)
    _, ap = flex_amd.perm_csr(a, rank)
    del a
    print(f"{name}: n={ap.m} nnz={ap.nnz} generated + ordered in {time.time() - t0:.1f} s", flush=True)
    B = torch.rand((ap.n, k), device="cuda") * 2 - 1
    C = torch.empty((ap.m, k), device="cuda")
    reps = 10 if ap.nnz > 1e8 else 30
    p = flex_amd.Plan(ap, k)
    t_full = timeit(p, B, C, reps)
    G = p.info()["lanes_per_nz"]
    p.destroy()
    print(f"{name} k={k} G={G}: whole matrix (natural order of the permuted loader) {t_full:.1f} us", flush=True)
    rows = np.repeat(np.arange(ap.m, dtype=np.uint64), np.diff(ap.rowPtr.astype(np.int64)))
    for R in (256, 512, 1024):
        key = ((rows // np.uint64(R)) << np.uint64(32)) | ap.col.astype(np.uint64)
        uniq, inv, cnt = np.unique(key, return_inverse=True, return_counts=True)
        uses = cnt[inv]  # per nonzero: how many nonzeros of its row block use its column
        nblk = (ap.m + R - 1) // R
        del key, inv
        for thr in (2, 4, 8):
            hot = uses >= thr
            n_hot = int(hot.sum())
            pairs = int((cnt >= thr).sum())
            share = n_hot / ap.nnz
            line = (f"{name} k={k} R={R:5d} thr={thr}: hot nnz {100 * share:5.1f} %  u={n_hot / max(pairs, 1):6.2f}  "
                    f"hot cols/block {pairs / nblk:8.1f} (= {pairs / nblk * 16 * G / 1024:7.1f} KiB of LDS at one {4 * G}-column tile)")
            if thr == 4 or (R == 512 and thr == 2):
                cold = sub_csr(ap, ~hot)
                pc = flex_amd.Plan(cold, k)
                t_cold = timeit(pc, B, C, reps)
                pc.destroy()
                hotm = sub_csr(ap, hot)
                ph = flex_amd.Plan(hotm, k)
                t_hot = timeit(ph, B, C, reps)
                ph.destroy()
                line += f" | flat kernel: cold part alone {t_cold:8.1f} us, hot part alone {t_hot:8.1f} us (whole {t_full:.1f})"
                del cold, hotm
            print(line, flush=True)
        del uniq, cnt, uses
