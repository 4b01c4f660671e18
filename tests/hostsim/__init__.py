"""Test infrastructure: libflex_hostsim.so = the library's HOST code (parsers, orderings, generator, planner) linked against
tests/hostsim/shim.cpp instead of the kernels, with "device memory" malloc'ed on the host.  It cannot compute an SpMM (every
launcher reports FLEX_ERR_UNSUPPORTED); it exists so that plans can be created and their image checked without a GPU."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
_CSRC = os.path.join(_ROOT, "flex_amd", "csrc")
_OUT = os.path.join(_HERE, "_build", "libflex_hostsim.so")
HOST_SOURCES = ["plan.cpp", "plan_build.cpp", "block_plan.cpp", "plan_check.cpp", "dense_tiles.cpp", "ingest.cpp", "reorder.cpp", "cluster.cpp", "rabbit.cpp", "gorder.cpp", "shard.cpp", "synth.cpp"]


def build(extra_flags=(), out=_OUT):
    srcs = [os.path.join(_CSRC, f) for f in HOST_SOURCES] + [os.path.join(_HERE, "shim.cpp")]
    deps = srcs + [os.path.join(_CSRC, h) for h in ("internal.h", "plan.h", "host_parallel.h")] + [os.path.join(_ROOT, "include", "flex_spmm.h")]
    if os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps):
        return out
    os.makedirs(os.path.dirname(out), exist_ok=True)
    cmd = ["g++", "-std=c++20", "-O1", "-fPIC", "-shared", "-DFLEX_HOSTSIM", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + _CSRC,
           "-Wl,-Bsymbolic-functions", "-o", out] + list(extra_flags) + srcs + ["-lpthread", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"]  # the HIP calls the shim does not replace are never reached
    subprocess.check_call(cmd)
    return out
