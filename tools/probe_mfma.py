#!/usr/bin/env python3
"""tools/probe_mfma.py [k] -- the MFMA dense-tile route against the vector kernel on block-dense inputs
(block-diagonal blocks of a given fill + random noise), over fill and routing threshold."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flex_amd  # noqa: E402
import tools._knobs  # noqa: E402,F401  (FLEX_* environment knobs -> plan descriptor)

k = int(sys.argv[1]) if len(sys.argv) > 1 else 128


def block_dense_graph(n, block, fill, noise_deg, seed):
    rng = np.random.default_rng(seed)
    nb = n // block
    n = nb * block
    mask = rng.random((nb, block, block)) < fill
    b, r, c = np.nonzero(mask)
    rows = [b * block + r, np.repeat(np.arange(n), noise_deg)]
    cols = [b * block + c, rng.integers(0, n, size=n * noise_deg)]
    r, c = np.concatenate(rows), np.concatenate(cols)
    key = np.unique(r.astype(np.int64) * n + c)
    r, c = key // n, key % n
    rp = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(r, minlength=n), out=rp[1:])
    vals = rng.uniform(-1, 1, size=len(c)).astype(np.float32)
    return flex_amd.HostCsr(rp.astype(np.uint32), c.astype(np.uint32), vals, n=n)


def timed(a, B, C, env):
    for key in ("FLEX_MFMA", "FLEX_MFMA_FILL"):
        os.environ.pop(key, None)
    os.environ.update({k_: str(v) for k_, v in env.items()})
    p = flex_amd.Plan(a, k, order=flex_amd.FLEX_ORDER_NATURAL)
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        p.spmm(B.data_ptr(), C.data_ptr(), s)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        p.spmm(B.data_ptr(), C.data_ptr(), s)
    e1.record()
    torch.cuda.synchronize()
    info = p.info()
    out = C.clone()
    p.destroy()
    return e0.elapsed_time(e1) * 1e3 / 20, info, out


# PROBE_CASES="64:0.3:8,..." restricts the cases (block:fill:noise), PROBE_THR="10" the routing thresholds (for profiler passes)
CASES = [(64, 0.9, 8), (64, 0.6, 8), (64, 0.3, 8), (64, 0.15, 8), (128, 0.3, 8), (32, 0.5, 16)]
if os.environ.get("PROBE_CASES"):
    CASES = [(int(c.split(":")[0]), float(c.split(":")[1]), int(c.split(":")[2])) for c in os.environ["PROBE_CASES"].split(",")]
THRS = [int(t) for t in os.environ.get("PROBE_THR", "10,25,50").split(",")]
for block, fill, noise in CASES:
    a = block_dense_graph(200_000, block, fill, noise, seed=1)
    B = torch.rand((a.n, k), device="cuda") * 2 - 1
    C = torch.empty((a.m, k), device="cuda")
    v_us, _, ref = timed(a, B, C, {"FLEX_MFMA": 2})
    line = f"block {block} fill {fill} noise {noise}: n={a.n} nnz={a.nnz}  vector {v_us:8.1f} us ({2e-3 * a.nnz * k / v_us:7.0f} GFLOPS)"
    for thr in THRS:
        us, info, out = timed(a, B, C, {"FLEX_MFMA": 1, "FLEX_MFMA_FILL": thr})
        err = (out - ref).abs().max().item()
        line += f" | thr {thr}%: {us:8.1f} us ({v_us / us:4.2f}x, {100.0 * info['tile_nnz'] / a.nnz:4.1f}% of nnz in {info['n_tiles']} tiles, |d|={err:.1e})"
    print(line, flush=True)
