#!/usr/bin/env python3
"""tools/probe_fixup.py [workload k ...] -- split rows summed inside the launch (relaxed sc1 hand-off) against the two-launch
form (spmm_fixup_kernel after the main kernel): launch time of both on the same box, alternating, best of 3 x n."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flex_amd  # noqa: E402
import tools._knobs  # noqa: E402,F401  (FLEX_* environment knobs -> plan descriptor)
from tools._timing import timeit  # noqa: E402

args = sys.argv[1:] or ["reddit", "128", "amazon", "128", "flickr", "128", "reddit", "32"]
for name, k in zip(args[0::2], (int(x) for x in args[1::2])):
    a = flex_amd.synth_graph(name)
    B = torch.rand((a.n, k), device="cuda") * 2 - 1
    C = torch.empty((a.m, k), device="cuda")
    plans = {}
    for mode in ("1", "2"):
        os.environ["FLEX_FUSED_FIXUP"] = mode
        plans[mode] = flex_amd.Plan(a, k, order=flex_amd.FLEX_ORDER_CLUSTER)
    os.environ.pop("FLEX_FUSED_FIXUP")
    res = {"1": [], "2": []}
    for rnd in range(3):
        for mode in ("1", "2"):
            res[mode].append(timeit(plans[mode], B, C, 10 if a.nnz > 1e8 else 30))
    i = plans["1"].info()
    print(f"{name} k={k}: split rows {i['n_split_rows']} partials {i['n_partials']} | in-launch "
          + "/".join(f"{t:.1f}" for t in res["1"]) + " us | two-launch " + "/".join(f"{t:.1f}" for t in res["2"]) + " us", flush=True)
