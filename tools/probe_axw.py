#!/usr/bin/env python3
"""GEMM half of the A*X*W layer in isolation: time per order for a shape, library chosen by FLEX_AXW_LIB (ablations)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flex_amd  # noqa: E402
import tools._knobs  # noqa: E402,F401  (FLEX_* environment knobs -> plan descriptor)
from flex_amd import axw  # noqa: E402

name, dim, c = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
a = flex_amd.synth_graph(name)
h = axw.Axw(a, dim, c)
X = torch.rand((a.n, dim), device="cuda") * 2 - 1
W = torch.rand((dim, c), device="cuda")
for order in (axw.FLEX_AXW_A_XW, axw.FLEX_AXW_AX_W):
    best = (1e9, 1e9)
    for _ in range(15):
        _, (g, s) = h.run(X, W, order, timed=True)
        best = min(best, (g, s))
    print(f"{name} dim={dim} c={c} order={order}: gemm {best[0]*1e3:7.1f} us  spmm {best[1]*1e3:7.1f} us", flush=True)
