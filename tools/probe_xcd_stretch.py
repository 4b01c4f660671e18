#!/usr/bin/env python3
"""tools/probe_xcd_stretch.py [workload k …] -- the eighths of the schedule against stretches dealt to the XCDs in turn
(tuning.xcd_slices = 3, xcd_stretch = workgroups per stretch): launch time (best of 3 rounds, alternating) and, with the in-run
counters, HBM-side bytes and L2 hit rate per variant.  STRETCHES=64,256,... picks the sweep."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flex_amd import counters  # noqa: E402

counters.init()
import torch  # noqa: E402

import flex_amd  # noqa: E402
from tools._timing import timeit  # noqa: E402

sync = torch.cuda.synchronize
torch.zeros(1, device="cuda")
stretches = [int(x) for x in os.environ.get("STRETCHES", "16,64,256,1024,4096").split(",")]
args = sys.argv[1:] or ["amazon", "128", "reddit", "128"]
for name, k in zip(args[0::2], (int(x) for x in args[1::2])):
    a = flex_amd.synth_graph(name)
    B = torch.rand((a.n, k), device="cuda") * 2 - 1
    C = torch.empty((a.m, k), device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    variants = [("eighths", {})] + [(f"dealt {st}", {"xcd_slices": 3, "xcd_stretch": st}) for st in stretches] + [("round-robin", {"xcd_slices": 2})]
    plans = [(label, flex_amd.Plan(a, k, order=flex_amd.FLEX_ORDER_CLUSTER, tuning=dict(t, blocks=2))) for label, t in variants]
    n = 10 if a.nnz > 1e8 else 30
    best = {label: 1e9 for label, _ in plans}
    for _ in range(3):
        for label, p in plans:
            best[label] = min(best[label], timeit(p, B, C, n, rounds=1))
    ref = None
    for label, p in plans:
        def steps():
            for _ in range(n):
                p.spmm(B.data_ptr(), C.data_ptr(), s)
        t = counters.traffic(steps, sync=sync, launches=n)
        l2 = counters.count(steps, counters.L2_PASS, sync=sync)
        Ch = C.clone()
        ref = Ch if ref is None else ref
        print(json.dumps({"workload": name, "k": k, "slices": label, "us": round(best[label], 1), "traffic_GB": round(t["traffic_bytes"] / 1e9, 3),
                          "traffic_TBps": round(t["traffic_bytes"] / best[label] / 1e6, 2),
                          "l2_hit": round(l2["TCC_HIT_sum"] / max(1.0, l2["TCC_HIT_sum"] + l2["TCC_MISS_sum"]), 4),
                          "slots": p.info()["n_slots"], "chunks": p.info()["n_chunks"], "same_result": bool(torch.equal(Ch, ref))}), flush=True)
    for _, p in plans:
        p.destroy()
