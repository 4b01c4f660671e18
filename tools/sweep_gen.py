#!/usr/bin/env python3
"""tools/sweep_gen.py [out.json] [workload] -- sensitivity of the Reddit-shape (or, second argument, Amazon-shape) measurements to the stand-in generator's structure
parameters (share of uniformly random edges, community size, width of the "near" ring).  Every point: the preset's n, nnz and
degree law, k = 128, cluster schedule; 2 warm-up + 5 timed launches (always 7 dispatches of spmm_flat_kernel per point, so
that a rocprofv3 --pmc pass of this same script can be cut into points by dispatch order: tools/sweep_gen_pmc.py)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flex_amd  # noqa: E402
import tools._knobs  # noqa: E402,F401  (FLEX_* environment knobs -> plan descriptor)

K = 128
POINTS = [  # (label, community, p_in, p_near, window)
    ("random 0.00", 2048, 0.75, 0.25, 8), ("random 0.05", 2048, 0.70, 0.25, 8), ("random 0.10", 2048, 0.65, 0.25, 8),
    ("random 0.15 (preset)", 2048, 0.60, 0.25, 8), ("random 0.20", 2048, 0.55, 0.25, 8), ("random 0.40", 2048, 0.35, 0.25, 8),
    ("community 512", 512, 0.60, 0.25, 8), ("community 8192", 8192, 0.60, 0.25, 8), ("near ring +-2", 2048, 0.60, 0.25, 2),
    ("near ring +-32", 2048, 0.60, 0.25, 32),
]
WL = sys.argv[2] if len(sys.argv) > 2 else "reddit"
if WL == "amazon":  # the headline shape: fewer points (each is 264 M nonzeros), its preset community is 4096
    POINTS = [("random 0.00", 4096, 0.75, 0.25, 8), ("random 0.15 (preset)", 4096, 0.60, 0.25, 8), ("random 0.40", 4096, 0.35, 0.25, 8),
              ("community 1024", 1024, 0.60, 0.25, 8), ("community 16384", 16384, 0.60, 0.25, 8), ("near ring +-2", 4096, 0.60, 0.25, 2)]
sp = flex_amd.synth_preset(WL)
res = []
for label, comm, p_in, p_near, win in POINTS:
    a = flex_amd.synth_graph(n=sp.n, nnz=sp.nnz, alpha=sp.alpha, community=comm, p_in=p_in, p_near=p_near, near_window=win,
                             shuffle=True, gcn_norm=bool(sp.gcn_norm), seed=sp.seed)
    B = torch.rand((a.n, K), device="cuda") * 2 - 1
    C = torch.empty((a.m, K), device="cuda")
    p = flex_amd.Plan(a, K, order=flex_amd.FLEX_ORDER_CLUSTER | flex_amd.FLEX_PLAN_STATS)
    del a.col, a.vals  # the plan has copied what it needs
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        p.spmm(B.data_ptr(), C.data_ptr(), s)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        p.spmm(B.data_ptr(), C.data_ptr(), s)
    e1.record()
    torch.cuda.synchronize()
    st, info = p.stats(), p.info()
    us = e0.elapsed_time(e1) * 1e3 / 5
    b_alg = 4.0 * (a.m + 1) + 8.0 * a.nnz + 8.0 * a.n * K
    row = {"point": label, "community": comm, "p_in": p_in, "p_near": p_near, "window": win, "nnz": a.nnz, "us": round(us, 1),
           "gflops": round(2e-3 * a.nnz * K / us, 1), "frac_8TBs": round(b_alg / us / 8e6, 4), "G": info["lanes_per_nz"],
           "b_reuse_xcd": round(st["reuse_xcd"], 2), "l2_model_MB": round(st["l2_bytes"] / 1e6, 1), "b_alg_MB": round(b_alg / 1e6, 1),
           "tile_nnz_pct_10": round(st["tile_nnz_pct_10"], 2), "tile_nnz_pct_25": round(st["tile_nnz_pct_25"], 2),
           "tile_mean_fill": round(st["tile_mean_fill"], 5), "mfma_tiles": info["n_tiles"], "flat_dispatches": 7}
    res.append(row)
    print(json.dumps(row), flush=True)
    p.destroy()
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
