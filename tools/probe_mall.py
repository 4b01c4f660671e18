#!/usr/bin/env python3
"""Streaming read / copy rate of flex_hbm_probe as a function of buffer size: L2-resident (<= 32 MiB over the
8 XCDs), Infinity-Cache-resident (<= 256 MiB) and HBM.  Usage: python tools/probe_mall.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402  (HIP runtime first)
import flex_amd  # noqa: E402
import tools._knobs  # noqa: E402,F401  (FLEX_* environment knobs -> plan descriptor)

for mib in (8, 16, 32, 64, 96, 128, 192, 256, 384, 512, 1024, 2048, 4096):
    reps = 40 if mib <= 512 else 10
    r, t = flex_amd.hbm_probe(0, mib=mib, reps=reps), flex_amd.hbm_probe(0, mib=mib, reps=reps, temporal=True)
    print(f"{mib:5d} MiB  nt: read {r['read_GBps']:8.1f} copy(r+w) {r['copy_GBps']:8.1f}   "
          f"temporal: read {t['read_GBps']:8.1f} copy(r+w) {t['copy_GBps']:8.1f}  GB/s", flush=True)
