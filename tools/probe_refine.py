#!/usr/bin/env python3
"""tools/probe_refine.py [workload k ...] -- the community order with and without its second stage (vertex moves between
stretches of the order, cluster.cpp) and over the stretch length: launch time and planning time per variant, in one process.
FLEX_CLUSTER_NO_REFINE / FLEX_CLUSTER_STRETCH are read at plan time."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flex_amd  # noqa: E402
import tools._knobs  # noqa: E402,F401  (FLEX_* environment knobs -> plan descriptor)


def timeit(p, B, C, n):
    s = torch.cuda.current_stream().cuda_stream
    best = 1e9
    for _ in range(3):
        for _ in range(3):
            p.spmm(B.data_ptr(), C.data_ptr(), s)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            p.spmm(B.data_ptr(), C.data_ptr(), s)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


args = sys.argv[1:] or ["flickr", "128", "yelp", "128", "reddit", "128", "reddit", "32", "amazon", "128"]
for name, k in zip(args[0::2], (int(x) for x in args[1::2])):
    a = flex_amd.synth_graph(name)
    B = torch.rand((a.n, k), device="cuda") * 2 - 1
    C = torch.empty((a.m, k), device="cuda")
    res = []
    for st in ("none", "256", "512", "1024", "2048", "none"):
        os.environ.pop("FLEX_CLUSTER_NO_REFINE", None)
        os.environ.pop("FLEX_CLUSTER_STRETCH", None)
        if st == "none":
            os.environ["FLEX_CLUSTER_NO_REFINE"] = "1"
        else:
            os.environ["FLEX_CLUSTER_STRETCH"] = st
        if a.nnz > 1e8 and st in ("256", "2048"):
            continue
        t0 = time.time()
        p = flex_amd.Plan(a, k, order=flex_amd.FLEX_ORDER_CLUSTER)
        tp = time.time() - t0
        res.append(f"{st}: {timeit(p, B, C, 10 if a.nnz > 1e8 else 30):.1f} us (plan {tp:.2f} s)")
        p.destroy()
    print(name, k, " | ".join(res), flush=True)
