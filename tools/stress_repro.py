#!/usr/bin/env python3
"""tools/stress_repro.py <workload> [launches] -- bit-reproducibility of the in-launch combination of row pieces at full size:
the same plan launched N times (other traffic in between), every C compared bit for bit with the first.  FLEX_2D=1 in the
environment stresses the many-pieces-per-chunk path."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flex_amd  # noqa: E402
import tools._knobs  # noqa: E402,F401  (FLEX_* environment knobs -> plan descriptor)

wl = sys.argv[1] if len(sys.argv) > 1 else "reddit"
n_launch = int(sys.argv[2]) if len(sys.argv) > 2 else 30
a = flex_amd.synth_graph(wl)
k = 128
B = torch.rand((a.n, k), device="cuda") * 2 - 1
p = flex_amd.Plan(a, k, order=flex_amd.FLEX_ORDER_CLUSTER)
info = p.info()
ref = p(B).clone()
C = torch.empty_like(ref)
filler = torch.empty(256 << 20, device="cuda")
bad = 0
for it in range(n_launch):
    C.fill_(float("nan"))
    if it % 2:
        filler.add_(1.0)
    p(B, out=C)
    torch.cuda.synchronize()
    if not torch.equal(C, ref):
        bad += 1
        print(f"launch {it}: {int((C != ref).sum())} elements differ", flush=True)
print(f"{wl}: two_d={info['two_d']} split_rows={info['n_split_rows']} partials={info['n_partials']} launches={n_launch} mismatching launches={bad}")
sys.exit(1 if bad else 0)
