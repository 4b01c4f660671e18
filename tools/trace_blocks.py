#!/usr/bin/env python3
"""tools/trace_blocks.py <workload> <k> [rounds:panel:thr:cap] -- where the consumer waves of the row-block kernel spend their
cycles (diagnostic -DFLEX_TRACE build: libflex_spmm_trace.so, `make -C flex_amd/csrc trace`; the product carries no stamps).
Per wave: prologue, cold phase, panel phases, waiting at barriers, window refills (inside the phases), epilogue."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flex_amd import binding  # noqa: E402

binding._SO = os.path.join(os.path.dirname(binding._SO), "libflex_spmm_trace.so")
import flex_amd  # noqa: E402

name, k = sys.argv[1], int(sys.argv[2])
rounds, prow, thr, cap, abl = (list(int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "4:480:3:0").split(":")) + [0])[:5]
gen = {kv.split("=")[0]: float(kv.split("=")[1]) for kv in os.environ.get("GEN", "").split(",") if kv}
if gen:
    sp = flex_amd.synth_preset(name)
    a = flex_amd.synth_graph(n=sp.n, nnz=sp.nnz, alpha=sp.alpha, community=int(gen.get("community", sp.community)), p_in=gen.get("p_in", sp.p_in),
                             p_near=gen.get("p_near", sp.p_near), near_window=int(gen.get("near_window", sp.near_window)), shuffle=True,
                             gcn_norm=bool(sp.gcn_norm), seed=sp.seed)
else:
    a = flex_amd.synth_graph(name)
p = flex_amd.Plan(a, k, order=flex_amd.FLEX_ORDER_CLUSTER, tuning={"blocks": 1, "block_rounds": rounds, "block_panel_rows": prow, "block_thr": thr, "block_cap": cap, "block_ablate": abl})
i = p.info()
nb, ktiles = i["n_blocks"], (k + 31) // 32
B = torch.rand((a.n, k), device="cuda") * 2 - 1
C = torch.empty((a.m, k), device="cuda")
log = torch.zeros((ktiles * nb * 15, 8), dtype=torch.int64, device="cuda")
L = flex_amd.lib()
L.flex_debug_set_trace.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
s = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    p.spmm(B.data_ptr(), C.data_ptr(), s)
L.flex_debug_set_trace(p._h, log.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
p.spmm(B.data_ptr(), C.data_ptr(), s)
e1.record()
torch.cuda.synchronize()
t = log.cpu().numpy().astype(np.float64)
tot = t[:, 6].sum()
names = ["prologue", "cold phase", "panel phases", "barrier wait", "(window refills)", "epilogue"]
print(f"{name} k={k}: {nb} blocks x {ktiles} tiles, launch {e0.elapsed_time(e1) * 1e3:.1f} us (stamped build), hot {100 * i['block_hot_nnz'] / max(i['block_nnz'], 1):.1f} % "
      f"pad {100 * (i['block_records'] / max(i['block_nnz'], 1) - 1):.1f} % panels/block {i['block_panels'] / nb:.1f}")
for j, nm in enumerate(names):
    print(f"  {nm:18s} {100 * t[:, j].sum() / tot:5.1f} % of the consumer waves' cycles")
wg = t[:, 6].reshape(ktiles * nb, 15)
print(f"  workgroup lifetime: mean {wg.max(1).mean() / 100:.1f} us of the 100 MHz clock?  (cycles: mean {wg.max(1).mean():.0f}, max {wg.max(1).max():.0f}); steps per wave mean {t[:, 7].mean():.0f}")
print(f"  cycles per step per wave: {t[:, 6].sum() / max(t[:, 7].sum(), 1):.0f}; cold+hot only: {(t[:, 1].sum() + t[:, 2].sum()) / max(t[:, 7].sum(), 1):.0f}")
