#!/usr/bin/env python3
"""tools/probe_tile_group.py [workload k ...] -- grouped tile order of the flat kernel: every XCD's slice of the schedule walked group by
group, all column tiles of a group back to back (a group's records re-read from the Infinity Cache), over the group size."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flex_amd  # noqa: E402
from tools._timing import timeit  # noqa: E402

args = sys.argv[1:] or ["amazon", "128", "reddit", "128"]
for name, k in zip(args[0::2], (int(x) for x in args[1::2])):
    a = flex_amd.synth_graph(name)
    vo, ap = flex_amd.perm_csr(a, flex_amd.order_cluster(a))
    del a
    B = torch.rand((ap.n, k), device="cuda") * 2 - 1
    C = torch.empty((ap.m, k), device="cuda")
    reps = 10 if ap.nnz > 1e8 else 30
    p = flex_amd.Plan(ap, k, vo_mp=vo)
    ref_t = timeit(p, B, C, reps)
    ref = C.clone()
    print(f"{name} k={k} G={p.info()['lanes_per_nz']}: one pass per tile {ref_t:.1f} us", flush=True)
    for tg in (64, 256, 1024, 4096, 16384):
        for nt in (2, 1):
            q = flex_amd.Plan(ap, k, vo_mp=vo, tuning={"tile_group": tg, "rec_nt": nt})
            t = timeit(q, B, C, reps)
            same = bool(torch.equal(C, ref))
            print(f"{name} k={k} tile_group={tg:6d} rec_nt={'on' if nt == 1 else 'off'}: {t:8.1f} us ({ref_t / t:.3f}x) same bits: {same}", flush=True)
            q.destroy()
