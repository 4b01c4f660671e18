#!/usr/bin/env python3
"""tools/probe_variant.py <workload> <k> [knob=value,... ...] -- one workload under the default plan and under each given set of
plan-time knobs (fields of flex_plan_tuning), with PROBE_LIB=<file in flex_amd/lib> an experiment build of the library: launch time
(best of 3 alternating rounds) and the in-run counters (HBM-side bytes, rate, L2 hit rate) per variant, results compared."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flex_amd import counters  # noqa: E402

counters.init()
import torch  # noqa: E402

import flex_amd  # noqa: E402
from flex_amd import binding  # noqa: E402
from tools._timing import timeit  # noqa: E402

if os.environ.get("PROBE_LIB"):
    binding._SO = os.path.join(os.path.dirname(binding._SO), os.environ["PROBE_LIB"])
sync = torch.cuda.synchronize
torch.zeros(1, device="cuda")
name, k = sys.argv[1], int(sys.argv[2])
sets = [{}] + [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in a.split(",")) for a in sys.argv[3:]]
gen = dict(kv.split("=") for kv in os.environ.get("GEN", "").split(",") if kv)
a = flex_amd.synth_graph(name, **{g: float(v) for g, v in gen.items()}) if gen else flex_amd.synth_graph(name)
B = torch.rand((a.n, k), device="cuda") * 2 - 1
C = torch.empty((a.m, k), device="cuda")
s = torch.cuda.current_stream().cuda_stream
plans = [(json.dumps(t), flex_amd.Plan(a, k, order=flex_amd.FLEX_ORDER_CLUSTER, tuning=t)) for t in sets]
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
    err = float((Ch - ref).abs().max())
    print(json.dumps({"lib": os.environ.get("PROBE_LIB", "libflex_spmm.so"), "workload": name, "k": k, "tuning": label, "us": round(best[label], 1),
                      "traffic_GB": round(t["traffic_bytes"] / 1e9, 3), "read_GB": round(t["read_bytes"] / 1e9, 3),
                      "traffic_TBps": round(t["traffic_bytes"] / best[label] / 1e6, 2),
                      "l2_hit": round(l2["TCC_HIT_sum"] / max(1.0, l2["TCC_HIT_sum"] + l2["TCC_MISS_sum"]), 4),
                      "l2_req": int(l2["TCC_REQ_sum"] / n), "max_abs_diff_vs_first": err}), flush=True)
