#!/usr/bin/env python3
"""tools/probe_blocks.py [workload k ...] -- the hot-block path (the matrix split: LDS-staged B panels for the nonzeros with reuse
inside a block of rows, the flat kernel for the rest) against the flat kernel alone on one workload: launch time, what the block
image looks like (hot share, u, panels, padding), over the knobs given as BLOCK_SWEEP ("rounds:panel_rows:thr:cap[:ablate],...";
default a small grid).  ablate 2 = no panel work (what the flat part + staging cost), 3 = neither staging nor work."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flex_amd  # noqa: E402
if os.environ.get("PROBE_LIB"):  # an experiment build of the library (make -C flex_amd/csrc block_variants)
    from flex_amd import binding
    binding._SO = os.path.join(os.path.dirname(binding._SO), os.environ["PROBE_LIB"])
from tools._timing import timeit  # noqa: E402

args = sys.argv[1:] or ["reddit", "128"]
grid = os.environ.get("BLOCK_SWEEP", "8:0:2:0,8:0:3:0,4:0:2:0,8:0:2:0:2,8:0:2:0:3")  # panel 0 = the build's panel size;  # a 5th field = block_ablate (timing only)
for name, k in zip(args[0::2], (int(x) for x in args[1::2])):
    gen = {kv.split("=")[0]: float(kv.split("=")[1]) for kv in os.environ.get("GEN", "").split(",") if kv}  # e.g. GEN=p_in=0.75,p_near=0.25
    if gen:
        sp = flex_amd.synth_preset(name)
        a = flex_amd.synth_graph(n=sp.n, nnz=sp.nnz, alpha=sp.alpha, community=int(gen.get("community", sp.community)), p_in=gen.get("p_in", sp.p_in),
                                 p_near=gen.get("p_near", sp.p_near), near_window=int(gen.get("near_window", sp.near_window)), shuffle=True,
                                 gcn_norm=bool(sp.gcn_norm), seed=sp.seed)
        name += " " + os.environ["GEN"]
    else:
        a = flex_amd.synth_graph(name)
    rank = flex_amd.order_cluster(a)
    vo, ap = flex_amd.perm_csr(a, rank)  # planned as a reordered loader: the ordering is computed once for all variants
    del a
    B = torch.rand((ap.n, k), device="cuda") * 2 - 1
    C = torch.empty((ap.m, k), device="cuda")
    reps = 10 if ap.nnz > 1e8 else 30
    p = flex_amd.Plan(ap, k, vo_mp=vo, tuning={"blocks": 2})  # the flat kernel, whatever the planner's rule would pick
    t_flat = timeit(p, B, C, reps)
    ref = C.clone()
    p.destroy()
    print(f"{name} k={k}: flat kernel {t_flat:.1f} us", flush=True)
    for spec in grid.split(","):
        rounds, prow, thr, cap, abl = (list(int(x) for x in spec.split(":")) + [0])[:5]
        t0 = time.time()
        pb = flex_amd.Plan(ap, k, vo_mp=vo, tuning={"blocks": 1, "block_rounds": rounds, "block_panel_rows": prow, "block_thr": thr, "block_cap": cap, "block_ablate": abl})
        tp = time.time() - t0
        i = pb.info()
        t = timeit(pb, B, C, reps)
        err = float((C - ref).abs().max())
        print(f"{name} k={k} rounds={rounds} panel={pb.tuning()['block_panel_rows']} thr={thr} cap={pb.tuning()['block_cap']}{' ABLATE=' + str(abl) if abl else ''}: {t:8.1f} us ({t_flat / t:.2f}x)  blocks {i['n_blocks']} rows {100 * i['block_rows'] / ap.m:.1f} % "
              f"hot {100 * i['block_hot_nnz'] / ap.nnz:.1f} % of nnz, u={i['block_hot_nnz'] / max(i['block_hot_cols'], 1):.2f} "
              f"panels/block {i['block_panels'] / max(i['n_blocks'], 1):.1f} pad {100 * (i['block_records'] / max(i['block_hot_nnz'], 1) - 1):.1f} % flat records {i['n_records']} plan {tp:.1f} s  max|diff vs flat| {err:.2e}", flush=True)
        pb.destroy()
