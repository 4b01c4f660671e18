#!/usr/bin/env python3
"""Block-density detector (north_star: "MFMA only where RCM/Gorder reordering yields dense block-sparse tiles").

For an ordering (rows AND columns relabelled by it) report how the nonzeros distribute over T x T tiles:
the fraction of nonzeros that sit in tiles with fill >= f.  An fp32 MFMA tile path (v_mfma_f32_32x32x2_f32
runs at the VALU rate, 157 TFLOP/s dense) does `1/fill` times the useful flops, and only pays by re-using
the tile's T gathered B rows T times; against the gather kernel's measured 6-8 TFLOP/s it needs fill >~ 0.1-0.25.
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flex_amd

T = 32
graphs = sys.argv[1].split(",") if len(sys.argv) > 1 else ["pubmed-file", "ppi", "flickr", "reddit"]
print(f"tile {T}x{T}; columns: fraction of nonzeros in tiles with fill >= f")
print(f"{'graph':18s} {'order':8s} {'tiles':>9s} {'mean fill':>9s} " + " ".join(f"f>={f:<5}" for f in (0.05, 0.1, 0.25, 0.5)))
for name in graphs:
    a = flex_amd.csv_load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "pubmed.csv")) if name == "pubmed-file" \
        else flex_amd.synth_graph(name)
    rows = np.repeat(np.arange(a.m, dtype=np.int64), np.diff(a.rowPtr.astype(np.int64)))
    for oname, fn in (("natural", None), ("rcm", flex_amd.order_rcm), ("gorder", flex_amd.order_gorder), ("cluster", flex_amd.order_cluster)):
        if oname == "gorder" and a.nnz > 5e6:
            continue
        rank = np.arange(a.n, dtype=np.int64) if fn is None else fn(a).astype(np.int64)
        key = (rank[rows] // T) * ((a.n + T - 1) // T) + rank[a.col] // T
        _, cnt = np.unique(key, return_counts=True)
        fill = cnt / float(T * T)
        shares = [cnt[fill >= f].sum() / a.nnz for f in (0.05, 0.1, 0.25, 0.5)]
        print(f"{name:18s} {oname:8s} {len(cnt):9d} {fill.mean():9.4f} " + " ".join(f"{s:7.3f}" for s in shares))
