#!/usr/bin/env python3
"""tools/probe_edge_classes.py [workload k] -- what the launch's bytes are made of: the stand-in generator's three classes of edges
(inside a community / within the ring of +-8 communities / uniformly random) switched off one at a time, same n, nnz and degree law,
flat kernel (BLOCKS=1: the row-block route), community schedule; launch time and the in-run counters per point."""
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
name, k = (sys.argv[1], int(sys.argv[2])) if len(sys.argv) > 2 else ("amazon", 128)
sp = flex_amd.synth_preset(name, 1)
rnd = 1.0 - sp.p_in - sp.p_near
points = [("preset", sp.p_in, sp.p_near), ("no random edges", sp.p_in + rnd, sp.p_near), ("no ring edges", sp.p_in + sp.p_near, 0.0),
          ("community edges only", 1.0, 0.0)]
for label, p_in, p_near in points:
    a = flex_amd.synth_graph(n=sp.n, nnz=sp.nnz, alpha=sp.alpha, community=sp.community, p_in=p_in, p_near=p_near, near_window=sp.near_window,
                             seed=sp.seed)
    B = torch.rand((a.n, k), device="cuda") * 2 - 1
    C = torch.empty((a.m, k), device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    route = int(os.environ.get("BLOCKS", "2"))  # 2: flat kernel (default here), 1: the row-block route, 0: the planner's rule
    p = flex_amd.Plan(a, k, order=flex_amd.FLEX_ORDER_CLUSTER, tuning={"blocks": route})
    n = 10 if a.nnz > 1e8 else 30
    us = timeit(p, B, C, n)

    def steps():
        for _ in range(n):
            p.spmm(B.data_ptr(), C.data_ptr(), s)
    t = counters.traffic(steps, sync=sync, launches=n)
    l2 = counters.count(steps, counters.L2_PASS, sync=sync)
    b_alg = 4.0 * (a.m + 1) + 8.0 * a.nnz + 8.0 * a.n * k
    print(json.dumps({"workload": name, "k": k, "route": {0: "rule", 1: "row blocks", 2: "flat"}[route], "blocks": p.info().get("n_blocks", 0), "edges": label, "p_in": round(p_in, 3), "p_near": round(p_near, 3), "random": round(1 - p_in - p_near, 3),
                      "nnz": a.nnz, "us": round(us, 1), "traffic_GB": round(t["traffic_bytes"] / 1e9, 2), "over_b_alg": round(t["traffic_bytes"] / b_alg, 2),
                      "traffic_TBps": round(t["traffic_bytes"] / us / 1e6, 2),
                      "l2_hit": round(l2["TCC_HIT_sum"] / max(1.0, l2["TCC_HIT_sum"] + l2["TCC_MISS_sum"]), 4),
                      "u_l2": round(4.0 * a.nnz * k / max(1.0, t["read_bytes"] - 8.0 * a.nnz - 4.0 * (a.m + 1)), 2)}), flush=True)
    p.destroy()
    del a, B, C
