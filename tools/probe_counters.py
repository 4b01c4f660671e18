#!/usr/bin/env python3
"""tools/probe_counters.py [workload k ...] -- the in-run counters (flex_amd/counters.py, libflex_counters.so) checked two ways:
(1) calibration on known byte counts in this engine's access widths: a 2 GiB device copy (16 B per lane: read 2 GiB, write
2 GiB) and a read-only stream; (2) the SpMM launch of a workload against the bytes the committed rocprofv3 passes recorded
for the same plan (profiles/pmc_traffic.json).  Prints one JSON line per check."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flex_amd import counters  # noqa: E402

counters.init()  # before torch touches the card
print("init ok", file=sys.stderr, flush=True)

import torch  # noqa: E402

import flex_amd  # noqa: E402

print("torch imported", file=sys.stderr, flush=True)
sync = torch.cuda.synchronize
torch.zeros(1, device="cuda")  # the runtime comes up here, and the profiler with it
print(json.dumps({"gpus_listed": counters.devices()}), flush=True)

gib = 2
src = torch.empty(gib << 28, dtype=torch.float32, device="cuda").normal_()
dst = torch.empty_like(src)
reps = 5


def copies():
    for _ in range(reps):
        dst.copy_(src)


for _ in range(2):  # the second round shows the repeatability
    t = counters.traffic(copies, sync=sync, launches=reps)
    print(json.dumps({"check": "copy", "bytes_read": gib << 30, "bytes_written": gib << 30,
                      "fetch_x2_over_read": round(t["read_bytes"] / (gib << 30), 4), "write_over_written": round(t["write_bytes"] / (gib << 30), 4),
                      **{k: round(v) for k, v in t.items()}}), flush=True)


def sums():
    for _ in range(reps):
        src.sum()


t = counters.traffic(sums, sync=sync, launches=reps)
print(json.dumps({"check": "read-only stream (torch sum)", "bytes_read": gib << 30,
                  "fetch_x2_over_read": round(t["read_bytes"] / (gib << 30), 4), "write_bytes": round(t["write_bytes"])}), flush=True)
del src, dst

recorded = {}
try:
    recorded = json.load(open(os.path.join(os.path.dirname(__file__), "..", "profiles", "pmc_traffic.json")))
except OSError:
    pass
args = sys.argv[1:] or ["reddit", "128", "flickr", "128", "amazon", "128"]
for name, k in zip(args[0::2], (int(x) for x in args[1::2])):
    a = flex_amd.synth_graph(name)
    B = torch.rand((a.n, k), device="cuda") * 2 - 1
    C = torch.empty((a.m, k), device="cuda")
    plan = flex_amd.Plan(a, k, order=flex_amd.FLEX_ORDER_CLUSTER)
    n = 10 if a.nnz > 1e8 else 30
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        plan.spmm(B.data_ptr(), C.data_ptr(), s)

    def steps():
        for _ in range(n):
            plan.spmm(B.data_ptr(), C.data_ptr(), s)

    t = counters.traffic(steps, sync=sync, launches=n)
    l2 = counters.count(steps, counters.L2_PASS, sync=sync)
    b_alg = 4.0 * (a.m + 1) + 8.0 * a.nnz + 4.0 * a.n * k + 4.0 * a.m * k
    rec = recorded.get(f"{name}_k{k}_cluster_n1", {})
    print(json.dumps({"check": f"{name} k={k}", "traffic_bytes": round(t["traffic_bytes"]), "read_bytes": round(t["read_bytes"]),
                      "write_bytes": round(t["write_bytes"]), "traffic_over_b_alg": round(t["traffic_bytes"] / b_alg, 2),
                      "l2_hit_rate": round(l2["TCC_HIT_sum"] / max(1.0, l2["TCC_HIT_sum"] + l2["TCC_MISS_sum"]), 4),
                      "u_l2": round(4.0 * a.nnz * k / max(1.0, t["read_bytes"] - 8.0 * a.nnz - 4.0 * (a.m + 1)), 3),
                      "rocprofv3_recorded": rec.get("bytes")}), flush=True)
