#!/usr/bin/env python3
"""tools/probe_counters_overhead.py [--attach] -- does the in-run counter tool (flex_amd/counters.py), once attached to the
process, change what the launches cost while NO pass is open?  Device time per launch (HIP events) and host time per launch
call, on the smallest and a mid-size workload; run once with and once without --attach on the same box."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
attach = "--attach" in sys.argv
if attach:
    from flex_amd import counters
    counters.init()
import torch  # noqa: E402

import flex_amd  # noqa: E402
from tools._timing import timeit  # noqa: E402

torch.zeros(1, device="cuda")
for name, k in (("pubmed.csv", 32), ("pubmed.csv", 128), ("flickr", 128), ("reddit", 128)):
    a = flex_amd.csv_load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", name)) if name.endswith(".csv") else flex_amd.synth_graph(name)
    B = torch.rand((a.n, k), device="cuda") * 2 - 1
    C = torch.empty((a.m, k), device="cuda")
    plan = flex_amd.Plan(a, k, order=flex_amd.FLEX_ORDER_CLUSTER)
    n = 2000 if a.nnz < 1e6 else 100
    dev_us = timeit(plan, B, C, n)
    s = torch.cuda.current_stream().cuda_stream
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        plan.spmm(B.data_ptr(), C.data_ptr(), s)
    host_us = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    print(json.dumps({"attached": attach, "gpus_listed": counters.devices() if attach else None, "workload": name, "k": k,
                      "device_us_per_launch": round(dev_us, 2), "host_us_per_launch_call": round(host_us, 2)}), flush=True)
