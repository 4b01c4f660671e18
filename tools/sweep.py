#!/usr/bin/env python3
"""Tuning sweep (GPU box): plan-time knobs x workloads, interleaved timing in ONE process
(cdna guide rule 24).  Not part of the product or the tests."""
import itertools
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flex_amd  # noqa: E402
import tools._knobs  # noqa: E402,F401  (FLEX_* environment knobs -> plan descriptor)


def timeit(plan, B, C, reps):
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(5):
        plan.spmm(B.data_ptr(), C.data_ptr(), s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        plan.spmm(B.data_ptr(), C.data_ptr(), s)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3  # us


def main():
    which = sys.argv[1].split(",") if len(sys.argv) > 1 else ["flickr"]
    ks = [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["128"])]
    wave_nnz = [int(x) for x in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["64", "128", "256", "512", "1024"])]
    for name in which:
        shuffle = True
        if name.endswith("-noshuf"):
            name, shuffle = name[:-7], False
        t = time.time()
        a = flex_amd.synth_graph(name, shuffle=shuffle)
        print(f"# {name} shuffle={shuffle} n={a.n} nnz={a.nnz} gen {time.time()-t:.1f}s", flush=True)
        for k in ks:
            B = torch.rand((a.n, k), device="cuda") * 2 - 1
            C = torch.empty((a.m, k), device="cuda")
            plans = {}
            variants = [int(x) for x in os.environ.get("SWEEP_KERNELS", "1,2").split(",")]
            remaps = [int(x) for x in os.environ.get("SWEEP_REMAP", "1,2").split(",")]
            orders = [int(x) for x in os.environ.get("SWEEP_ORDERS", "0,1").split(",")]
            ldsx = [int(x) for x in os.environ.get("SWEEP_LDS", "1").split(",")]
            for wn, order, remap, kv, lx in itertools.product(wave_nnz, orders, remaps, variants, ldsx):
                os.environ["FLEX_WAVE_NNZ"] = str(wn)
                os.environ["FLEX_XCD_REMAP"] = str(remap)
                os.environ["FLEX_KERNEL"] = str(kv)
                os.environ["FLEX_LDS_EXTRA"] = str(lx)
                plans[(wn, order, remap, kv, lx)] = flex_amd.Plan(a, k, order=order)
            reps = 20 if a.nnz > 5e6 else 100
            res = {key: [] for key in plans}
            for rnd in range(3):
                for key, p in plans.items():
                    res[key].append(timeit(p, B, C, reps))
            balg = 4 * (a.m + 1) + 8 * a.nnz + 8 * a.n * k
            for key, v in sorted(res.items()):
                us = min(v)
                i = plans[key].info()
                print(f"{name:8s} k={k:4d} wave_nnz={key[0]:5d} order={('nat','rcm','clu')[key[1]]} remap={'on' if key[2]==1 else 'off'} kern={'flat' if key[3]==1 else 'row'} ldsx={key[4]:6d} "
                      f"chunks={i['n_chunks']:7d} split={i['n_split_rows']:6d} t={us:9.1f}us med={np.median(v):9.1f} "
                      f"GFLOPS={2*a.nnz*k/us/1e3:9.1f} Balg={balg/us/1e3:8.1f}GB/s gather={a.nnz*k*4/us/1e3:8.1f}GB/s", flush=True)
            for p in plans.values():
                p.destroy()


if __name__ == "__main__":
    main()
