#!/usr/bin/env python3
"""tools/plan_on_host.py workload k [knob=value ...] -- create a plan WITHOUT a GPU (tests/hostsim: the library's host code, "device
memory" = malloc) and print what the planner says about it (FLEX_PLAN_TIMING lines on stderr, the plan's info).  For looking at the
hot-block split -- candidates, what was left cold and why -- at the full BASELINE sizes in the container."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("FLEX_PLAN_TIMING", "1")
import hostsim  # noqa: E402
from flex_amd import binding  # noqa: E402

so = hostsim.build(extra_flags=("-O2",), out=os.path.join(ROOT, "tests", "hostsim", "_build", "libflex_hostsim_O2.so"))
binding._SO, binding._lib = so, None
import flex_amd  # noqa: E402

name, k = sys.argv[1], int(sys.argv[2])
knobs = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in sys.argv[3:]}
t0 = time.time()
a = flex_amd.synth_graph(name)
rank = flex_amd.order_cluster(a)
vo, ap = flex_amd.perm_csr(a, rank)
print(f"{name}: n={a.m} nnz={a.nnz} generated + ordered in {time.time() - t0:.1f} s", flush=True)
del a
for variant in (knobs,) if knobs else ({"blocks": 1},):
    t0 = time.time()
    p = flex_amd.Plan(ap, k, vo_mp=vo, tuning=variant)
    i = p.info()
    print(f"{variant}: planned in {time.time() - t0:.1f} s; flat records {i['n_records']} ({100 * i['n_records'] / ap.nnz:.1f} % of nnz), G={i['lanes_per_nz']}, "
          f"hot {100 * i['block_hot_nnz'] / ap.nnz:.1f} %, blocks {i['n_blocks']}, panels/block {i['block_panels'] / max(i['n_blocks'], 1):.1f}, "
          f"records/hot {i['block_records'] / max(i['block_hot_nnz'], 1):.2f}", flush=True)
    p.destroy()
