"""ctypes binding of libflex_counters.so (include/flex_counters.h): the card's memory counters read inside the run.

≙ the NPerf_* calls of the reference's run() (flex.cu:4583-4656) and the L1<->L2 / DRAM / `u` columns it prints per table
row (flex.cu:5237).  `init()` must come before the process makes its first HIP call (in Python: before
`torch.cuda.is_available()` / the first `.cuda()`, and before the first `flex_amd.lib()` call that touches the card).
"""
from __future__ import annotations

import ctypes as C
import os

_SO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libflex_counters.so")
_lib = None

# the gfx950 corrections of MI355X_MICROARCH "HBM" (the same ones tools/pmc_summary.py applies to the rocprofv3 passes): both
# counters are in KiB, and FETCH_SIZE tallies a 128-byte request as 64 bytes -- double it before comparing with a byte count
FETCH_BYTES_PER_UNIT = 2 * 1024
WRITE_BYTES_PER_UNIT = 1024
# FETCH_SIZE takes 3 of the TCC block's 4 counter slots and WRITE_SIZE 2: one pass each
TRAFFIC_PASSES = (("FETCH_SIZE",), ("WRITE_SIZE",))
L2_PASS = ("TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum", "TCC_READ_sum")  # the TCC block's four slots
L2_REQUEST_BYTES = 128  # a TCC request is one 128-byte line (amazon k=128: 1.09e9 requests per launch for 138 GB of gather demand)
# wave-level instruction counts (≙ the reference's "Per Mult / Num Insns" columns, flex.cu:5350-5420): one SQ pass
SQ_PASS = ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS", "SQ_INSTS_SMEM")


class CountersError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise CountersError(f"{_SO} is missing: run __graft_entry__.build() (make -C flex_amd/csrc)")
        L = C.CDLL(_SO, mode=C.RTLD_GLOBAL)  # rocprofiler-register looks rocprofiler_configure up over the whole process
        L.flex_counters_begin.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.c_int]
        L.flex_counters_end.argtypes = [C.POINTER(C.c_double)]
        L.flex_counters_error.restype = C.c_char_p
        _lib = L
    return _lib


def _check(rc: int, what: str):
    if rc != 0:
        raise CountersError(f"{what} failed ({rc}): {lib().flex_counters_error().decode()}")


def init():
    """Asks for the profiler, which comes up with the runtime.  Call before the first HIP call of the process."""
    _check(lib().flex_counters_init(), "flex_counters_init")


def devices() -> int:
    """GPUs the profiler lists; 0 until the runtime has initialised (or when init came too late)."""
    return lib().flex_counters_devices()


def begin(names, device: int = 0):
    arr = (C.c_char_p * len(names))(*[n.encode() for n in names])
    _check(lib().flex_counters_begin(device, arr, len(names)), f"flex_counters_begin({', '.join(names)})")


def end(n: int):
    out = (C.c_double * n)()
    _check(lib().flex_counters_end(out), "flex_counters_end")
    return list(out)


def count(fn, names, device: int = 0, sync=None) -> dict:
    """Runs fn() between begin and end; `sync` (e.g. torch.cuda.synchronize) is called before each, so that the pass sees
    exactly fn's device work.  Returns {name: sum over all instances}."""
    if sync:
        sync()
    begin(names, device)
    try:
        fn()
        if sync:
            sync()
    finally:
        vals = end(len(names))
    return dict(zip(names, vals))


def traffic(fn, device: int = 0, sync=None, launches: int = 1) -> dict:
    """HBM-side bytes of fn() (which makes `launches` launches), per launch: one pass for FETCH_SIZE, one for WRITE_SIZE, with
    the gfx950 corrections applied.  `traffic_bytes` = 2*FETCH_SIZE + WRITE_SIZE in bytes, the figure bench.py's
    roofline.traffic carries."""
    f = count(fn, TRAFFIC_PASSES[0], device, sync)["FETCH_SIZE"]
    w = count(fn, TRAFFIC_PASSES[1], device, sync)["WRITE_SIZE"]
    rd, wr = f * FETCH_BYTES_PER_UNIT / launches, w * WRITE_BYTES_PER_UNIT / launches
    return {"fetch_size_raw": f / launches, "write_size_raw": w / launches, "read_bytes": rd, "write_bytes": wr, "traffic_bytes": rd + wr}
