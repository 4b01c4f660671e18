#!/usr/bin/env python3
"""tools/probe_2d.py <workload> [k] -- the 2-D (column-panel) schedule against the 1-D one on one graph:
launch time, plan shape and agreement of the results, over panel size / minimum run / column-tile width.
Environment knobs are read at plan time, so every variant is a fresh plan in this one process."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flex_amd  # noqa: E402
import tools._knobs  # noqa: E402,F401  (FLEX_* environment knobs -> plan descriptor)

wl = sys.argv[1] if len(sys.argv) > 1 else "reddit"
k = int(sys.argv[2]) if len(sys.argv) > 2 else 128
variants = sys.argv[3:] or None
shuffle = int(os.environ.get("PROBE_SHUFFLE", "1"))
a = flex_amd.synth_graph(wl, shuffle=bool(shuffle))
t0 = time.perf_counter()
vo, ap = flex_amd.perm_csr(a, flex_amd.order_cluster(a))  # the community order ONCE; every variant plans the permuted matrix
print(f"order+perm {time.perf_counter() - t0:.1f}s", flush=True)
g = torch.Generator(device="cuda")
g.manual_seed(1)
B = torch.rand((a.n, k), generator=g, device="cuda") * 2 - 1
C = torch.empty((a.m, k), device="cuda")
stream = torch.cuda.current_stream().cuda_stream
deg = torch.from_numpy(np.diff(a.rowPtr.astype(np.int64))).cuda().clamp(min=1).float().unsqueeze(1)
steps = 5 if a.nnz > 100_000_000 else 30


def run(env):
    for key in [k_ for k_ in os.environ if k_.startswith("FLEX_") and k_ != "FLEX_HOST_THREADS"]:  # every plan-time knob, whatever the variant before set
        os.environ.pop(key, None)
    os.environ.update({k_: str(v) for k_, v in env.items()})
    t0 = time.perf_counter()
    p = flex_amd.Plan(ap, k, vo_mp=vo)
    t_plan = time.perf_counter() - t0
    for _ in range(2):
        p.spmm(B.data_ptr(), C.data_ptr(), stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        p.spmm(B.data_ptr(), C.data_ptr(), stream)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / steps
    info = p.info()
    out = C.clone()
    p.destroy()
    return us, info, t_plan, out


base_us, info, tp, ref = run({"FLEX_2D": 2})
print(f"{wl} k={k} n={a.n} nnz={a.nnz}  1-D: {base_us:9.1f} us  G={info['lanes_per_nz']} chunks={info['n_chunks']} tasks={info['n_tasks']} "
      f"partials={info['n_partials']} plan={tp:.2f}s", flush=True)
grid = []
if variants:
    for v in variants:
        grid.append(dict(kv.split("=") for kv in v.split(",")))
else:
    for kb in (1024, 2048, 3072):
        for seg in (3, 5, 8):
            grid.append({"FLEX_2D": 1, "FLEX_PANEL_KB": kb, "FLEX_SEG_MIN": seg})
for env in grid:
    env.setdefault("FLEX_2D", 1)
    us, info, tp, out = run(env)
    err = ((out - ref).abs() / (ref.abs().clamp(min=1.0) * deg)).max().item() / np.finfo(np.float32).eps
    print(f"  {env}: {us:9.1f} us ({base_us / us:5.2f}x)  G={info['lanes_per_nz']} P={info['panel_rows']} chunks={info['n_chunks']} tasks={info['n_tasks']} "
          f"partials={info['n_partials']} MB={info['device_bytes'] >> 20} plan={tp:.2f}s  max|d|/(eps*deg)={err:.2f}", flush=True)
