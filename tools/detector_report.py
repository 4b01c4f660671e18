#!/usr/bin/env python3
"""tools/detector_report.py -- the block-density detector's verdict (flex_plan_stats) for every BASELINE graph:
share of the nonzeros in 32x32 tiles (schedule coordinates) of fill >= 0.10 / 0.25 / 0.50, and what the default rule routes
to the MFMA kernel.  k = 128, cluster schedule (and RCM / Gorder where they are affordable)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flex_amd  # noqa: E402
import tools._knobs  # noqa: E402,F401  (FLEX_* environment knobs -> plan descriptor)

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "pubmed.csv")
print(f"{'graph':28s} {'order':8s} {'mean fill':>9s} {'>=0.10':>8s} {'>=0.25':>8s} {'>=0.50':>8s} {'MFMA tiles':>10s} {'routed nnz':>10s}")
for name in ("pubmed.csv", "wiki-vote", "soc-sign-epinions", "ppi", "flickr", "yelp", "reddit", "amazon"):
    a = flex_amd.csv_load(GOLDEN) if name == "pubmed.csv" else flex_amd.synth_graph(name)
    orders = [("cluster", flex_amd.FLEX_ORDER_CLUSTER)]
    if a.nnz < 30_000_000:
        orders.append(("rcm", flex_amd.FLEX_ORDER_RCM))
    if a.nnz < 2_000_000 and name not in ("wiki-vote", "soc-sign-epinions"):  # Gorder refuses isolated vertices
        orders.append(("gorder", flex_amd.FLEX_ORDER_GORDER))
    for oname, o in orders:
        p = flex_amd.Plan(a, 128, order=o | flex_amd.FLEX_PLAN_STATS)
        st = p.stats()
        print(f"{name:28s} {oname:8s} {st['tile_mean_fill']:9.5f} {st['tile_nnz_pct_10']:7.2f}% {st['tile_nnz_pct_25']:7.2f}% "
              f"{st['tile_nnz_pct_50']:7.2f}% {st['mfma_tiles']:10d} {st['mfma_nnz_pct']:9.2f}%", flush=True)
        p.destroy()
