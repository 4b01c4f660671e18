#!/usr/bin/env python3
"""Per-wave timeline of one SpMM launch (diagnostic build libflex_spmm_trace.so): which XCD ran
which part of the schedule, when each XCD finished, how long waves live."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import flex_amd  # noqa: E402
import tools._knobs  # noqa: E402,F401  (FLEX_* environment knobs -> plan descriptor)
from flex_amd import binding  # noqa: E402

binding._SO = os.path.join(ROOT, "flex_amd", "lib", "libflex_spmm_trace.so")


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "flickr"
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    order = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    a = flex_amd.synth_graph(name)
    B = torch.rand((a.n, k), device="cuda") * 2 - 1
    C = torch.empty((a.m, k), device="cuda")
    p = flex_amd.Plan(a, k, order=order)
    info = p.info()
    nw = info['n_slots']
    log = torch.zeros((nw, 12), dtype=torch.int64, device="cuda")
    L = flex_amd.lib()
    L.flex_debug_set_trace.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        p.spmm(B.data_ptr(), C.data_ptr(), s)
    torch.cuda.synchronize()
    L.flex_debug_set_trace(p._h, log.data_ptr())
    p.spmm(B.data_ptr(), C.data_ptr(), s)
    torch.cuda.synchronize()
    t = log.cpu().numpy()
    t = t[(t[:, 1] != 0) & (t[:, 3] > 0)]  # drop the empty slots that pad the XCD slices
    xcc, t0, t1, nrec, nch, home = t[:, 0], t[:, 1], t[:, 2], t[:, 3], t[:, 4], t[:, 5]
    base = t0.min()
    tick = 0.01  # s_memrealtime: 100 MHz -> 0.01 us per tick
    print(f"{name} k={k} order={order} chunks={info['n_chunks']} waves={len(t)} span={(t1.max()-base)*tick:.1f}us")
    for x in range(8):
        m = xcc == x
        if not m.any():
            continue
        print(f" xcc{x}: waves={m.sum():6d} chunks={nch[m].sum():6d} recs={nrec[m].sum():8d} first_start={(t0[m].min()-base)*tick:6.1f} "
              f"last_start={(t0[m].max()-base)*tick:6.1f} first_exit={(t1[m].min()-base)*tick:6.1f} last_exit={(t1[m].max()-base)*tick:6.1f}")
    print(" chunks per wave: mean %.2f max %d; waves with 0 chunks: %d" % (nch.mean(), nch.max(), (nch == 0).sum()))
    ph = t[:, 6:11].sum(axis=0) * tick
    tot = ((t1 - t0) * tick).sum()
    names = ["descriptors", "records->LDS", "gather wait", "fma+flush", "(unused)"]
    print(" wave-time shares (stamped build, waits drained at every stamp): " +
          ", ".join(f"{n} {100*v/tot:.1f}%" for n, v in zip(names, ph)) + f"; total {tot/len(t):.1f} us/wave")
    span = (t1.max() - base) * tick
    step = max(1.0, span / 25)
    edges = np.arange(0, span + step, step)
    live = [int(np.sum(((t0 - base) * tick < e + step) & ((t1 - base) * tick > e))) for e in edges[:-1]]
    print(f" live waves per {step:.1f}us bucket:", live)
    life = (t1 - t0) * tick
    print(" wave life us: mean %.2f p50 %.2f p90 %.2f p99 %.2f max %.2f" % (life.mean(), *np.percentile(life, [50, 90, 99, 100])))
    # per-CU table (≙ the reference's per-SM timing, flex.cu:27-79, 5087-5126): HW_ID = wave[3:0] simd[5:4] cu[11:8] sh[12] se[14:13]
    hw = t[:, 11]
    cu_key = xcc * 4096 + ((hw >> 13) & 3) * 1024 + ((hw >> 12) & 1) * 512 + ((hw >> 8) & 15)
    keys = np.unique(cu_key)
    busy_us = np.array([life[cu_key == kk].sum() for kk in keys])
    recs_cu = np.array([nrec[cu_key == kk].sum() for kk in keys])
    last_cu = np.array([((t1[cu_key == kk].max() - base) * tick) for kk in keys])
    print(f" CUs seen: {len(keys)}; wave-time per CU us: min {busy_us.min():.0f} mean {busy_us.mean():.0f} max {busy_us.max():.0f} "
          f"({100 * busy_us.max() / busy_us.mean() - 100:.0f}% imb); records per CU: min {recs_cu.min()} mean {recs_cu.mean():.0f} "
          f"max {recs_cu.max()} ({100 * recs_cu.max() / recs_cu.mean() - 100:.0f}% imb); last exit per CU us: "
          f"min {last_cu.min():.1f} mean {last_cu.mean():.1f} max {last_cu.max():.1f}")
    per_xcc = [int(np.sum((keys // 4096) == x)) for x in range(8)]
    print(f" CUs per XCC: {per_xcc}")
    busy = t[nch > 0]
    ex = (busy[:, 2] - base) * tick
    print(" exit time of busy waves us: p1 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % tuple(np.percentile(ex, [1, 50, 90, 99, 100])))


if __name__ == "__main__":
    main()
