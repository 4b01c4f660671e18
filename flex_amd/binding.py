"""ctypes binding of include/flex_spmm.h (the C ABI of libflex_spmm.so).

Device buffers are raw pointers (``tensor.data_ptr()``), streams are raw hipStream_t
values (``torch.cuda.current_stream().cuda_stream``): torch is plumbing here, the
kernels live in the shared library.  There is no fallback: if the library is missing
or a call fails, FlexError is raised.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
_SO = os.path.join(_HERE, "lib", "libflex_spmm.so")

FLEX_ORDER_NATURAL = 0
FLEX_ORDER_RCM = 1
FLEX_ORDER_CLUSTER = 2
FLEX_ORDER_GORDER = 3
FLEX_PLAN_STATS = 0x100
FLEX_PLAN_AUTOTUNE = 0x200
FLEX_PLAN_ROW_RANGE = 0x1000
FLEX_PLAN_XCD_INTERLEAVE = 0x2000


class FlexError(RuntimeError):
    pass


class _Csr(C.Structure):  # flex_csr
    _fields_ = [("m", C.c_int32), ("n", C.c_int32), ("nnz", C.c_int64),
                ("rowPtr", C.c_void_p), ("col", C.c_void_p), ("vals", C.c_void_p)]


class _HostCsr(C.Structure):  # flex_host_csr
    _fields_ = [("m", C.c_int32), ("n", C.c_int32), ("nnz", C.c_int64),
                ("rowPtr", C.POINTER(C.c_uint32)), ("col", C.POINTER(C.c_uint32)),
                ("vals", C.POINTER(C.c_float)),
                ("uni_nb", C.c_int64), ("n_edges_one_way", C.c_int64),
                ("n_edges_asymmetric", C.c_int64),
                ("n_nodes_z_out", C.c_int32), ("n_nodes_z_in", C.c_int32),
                ("n_nodes_z_deg", C.c_int32), ("is_directed", C.c_int32), ("c", C.c_int32)]


class _PlanInfo(C.Structure):  # flex_plan_info
    _fields_ = [("m", C.c_int32), ("n", C.c_int32), ("k", C.c_int32), ("device", C.c_int32),
                ("nnz", C.c_int64), ("n_tasks", C.c_int64), ("n_chunks", C.c_int64),
                ("n_split_rows", C.c_int64), ("n_partials", C.c_int64),
                ("device_bytes", C.c_int64), ("lanes_per_nz", C.c_int32), ("order", C.c_int32),
                ("plan_ms", C.c_double), ("n_slots", C.c_int64), ("two_d", C.c_int32), ("panel_rows", C.c_int32),
                ("n_tiles", C.c_int64), ("tile_nnz", C.c_int64), ("n_records", C.c_int64),
                ("n_blocks", C.c_int64), ("block_rows", C.c_int64), ("block_nnz", C.c_int64), ("block_hot_nnz", C.c_int64),
                ("block_hot_cols", C.c_int64), ("block_panels", C.c_int64), ("block_records", C.c_int64),
                ("n_bundles", C.c_int64), ("bundle_rows", C.c_int64)]


class _PlanStats(C.Structure):  # flex_plan_stats
    _fields_ = [("records", C.c_int64), ("cols_wave", C.c_int64), ("cols_wg", C.c_int64),
                ("cols_xcd", C.c_int64), ("reuse_wave", C.c_double), ("reuse_wg", C.c_double),
                ("reuse_xcd", C.c_double), ("gather_bytes", C.c_double), ("l2_bytes", C.c_double),
                ("chunk_rec_max", C.c_int64), ("chunk_rec_mean", C.c_double),
                ("chunk_imb_pct", C.c_double), ("xcd_imb_pct", C.c_double),
                ("split_nnz_pct", C.c_double), ("pad_pct", C.c_double), ("n_workgroups", C.c_int64),
                ("tile_nnz_pct_10", C.c_double), ("tile_nnz_pct_25", C.c_double), ("tile_nnz_pct_50", C.c_double),
                ("tile_mean_fill", C.c_double), ("mfma_tiles", C.c_int64), ("mfma_nnz_pct", C.c_double),
                ("lds_hot_pct_2", C.c_double), ("lds_hot_pct_4", C.c_double), ("lds_u_2", C.c_double), ("lds_u_4", C.c_double)]


class _ClusterTuning(C.Structure):  # flex_cluster_tuning
    _fields_ = [(f, C.c_int32) for f in ("batch", "no_refine", "stretch", "sweeps", "stride")]


class _PlanTuning(C.Structure):  # flex_plan_tuning: every field 0 = the planner's rule
    _fields_ = [(f, C.c_int32) for f in (
        "lanes_per_nz", "chunk_records", "long_row", "piece_records", "row_cost", "xcd_slices", "xcd_balance",
        "chunk_cost", "task_cost", "split_rows", "rec_nt", "unroll", "two_d", "panel_kb", "seg_min", "mfma",
        "mfma_fill_pct", "lds_extra", "host_threads")] + [("cluster", _ClusterTuning)] + [(f, C.c_int32) for f in (
        "blocks", "block_rounds", "block_panel_rows", "block_thr", "block_cap", "block_ablate", "tile_group", "xcd_stretch", "far_first", "bundle", "bundle_len")] + [("reserved", C.c_int32 * 5)]


TUNING_FIELDS = tuple(f for f, _ in _PlanTuning._fields_ if f not in ("cluster", "reserved"))
CLUSTER_TUNING_FIELDS = tuple(f for f, _ in _ClusterTuning._fields_)


# Knobs applied to every Plan() of this PROCESS that does not name them itself: a convenience of this binding for tests and
# tools whose plans are created inside helpers (the library has no such state: it only sees the descriptor it is handed).
DEFAULT_TUNING: dict = {}


def _tuning(d) -> "_PlanTuning | None":
    """dict -> flex_plan_tuning; keys are its field names, cluster knobs as cluster_<field> (cluster_no_refine=1 ...)."""
    if not d:
        return None
    t = _PlanTuning()
    for key, val in d.items():
        if key in TUNING_FIELDS:
            setattr(t, key, int(val))
        elif key.startswith("cluster_") and key[8:] in CLUSTER_TUNING_FIELDS:
            setattr(t.cluster, key[8:], int(val))
        else:
            raise FlexError(f"unknown tuning knob {key!r} (flex_plan_tuning has: {', '.join(TUNING_FIELDS)}, cluster_*)")
    return t


class _PlanDesc(C.Structure):  # flex_plan_desc
    _fields_ = [("struct_size", C.c_size_t), ("A", C.POINTER(_Csr)), ("k", C.c_int), ("ldb", C.c_int), ("ldc", C.c_int),
                ("device", C.c_int), ("flags", C.c_uint), ("row_begin", C.c_int64), ("row_end", C.c_int64),
                ("col_map", C.c_void_p), ("row_map", C.c_void_p), ("tuning", C.POINTER(_PlanTuning))]


class _KernelInfo(C.Structure):  # flex_kernel_info
    _fields_ = [("vgprs", C.c_int32), ("sgprs", C.c_int32), ("lds_bytes", C.c_int32), ("scratch_bytes", C.c_int32),
                ("threads_per_block", C.c_int32), ("waves_per_cu", C.c_int32)]


class _Imbalance(C.Structure):  # flex_imbalance
    _fields_ = [("waves", C.c_int64), ("cus_seen", C.c_int32), ("xcds_seen", C.c_int32), ("span_us", C.c_double),
                ("cu_busy_imb_pct", C.c_double), ("cu_end_spread_pct", C.c_double), ("xcd_busy_imb_pct", C.c_double),
                ("xcd_end_spread_pct", C.c_double), ("wave_us_mean", C.c_double), ("wave_us_max", C.c_double)]


class _SynthParams(C.Structure):  # flex_synth_params
    _fields_ = [("n", C.c_int64), ("nnz", C.c_int64), ("alpha", C.c_double),
                ("community", C.c_int64), ("p_in", C.c_double), ("p_near", C.c_double),
                ("near_window", C.c_int32), ("shuffle", C.c_int32), ("gcn_norm", C.c_int32),
                ("directed", C.c_int32), ("seed", C.c_uint64)]


# every symbol include/flex_spmm.h declares (tests/test_abi.py checks the header against this)
SYMBOLS = [
    "flex_plan_create", "flex_plan_create_ex", "flex_plan_create_ld", "flex_plan_create_mapped", "flex_plan_create_rows", "flex_spmm",
    "flex_plan_destroy", "flex_plan_measure_imbalance", "flex_plan_get_info", "flex_plan_get_stats", "flex_plan_get_tuning", "flex_set_host_threads", "flex_order_cluster_ex", "flex_plan_self_check", "flex_plan_kernel_info", "flex_hbm_probe", "flex_gather_rows", "flex_csv_load", "flex_mtx_load",
    "flex_csv_save", "flex_csr_save_bin", "flex_csr_load_bin", "flex_csr_fingerprint", "flex_perm_save", "flex_perm_load",
    "flex_host_csr_free", "flex_fill_dense_rand", "flex_order_rcm", "flex_order_cluster", "flex_order_gorder", "flex_perm_csr",
    "flex_order_deg", "flex_order_dfs", "flex_order_rabbit", "flex_shard_rows", "flex_synth_graph", "flex_synth_preset", "flex_strerror", "flex_last_hip_error",
    "flex_last_hip_error_string", "flex_abi_version",
]

_lib = None


def lib_path() -> str:
    return _SO


def build(force: bool = False) -> str:
    """Compile libflex_spmm.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    if force and os.path.exists(_SO):
        os.remove(_SO)
    subprocess.check_call(["make", "-s", "-C", _CSRC, "all"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise FlexError(f"{_SO} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(there is no CPU fallback)")
        # torch bundles its own libamdhip64.so; if ours were loaded first, libflex_spmm.so would bind
        # to /opt/rocm's copy and the process would hold TWO HIP runtimes (the second one sees no
        # device, and pointers/streams of one are foreign to the other).  Import torch first so the
        # library resolves to the runtime that owns the tensors it is handed.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(_SO)
        vp, i64, i32, u32 = C.c_void_p, C.c_int64, C.c_int, C.c_uint
        L.flex_plan_create.argtypes = [C.POINTER(vp), C.POINTER(_Csr), i32, i32, u32]
        L.flex_plan_create_ex.argtypes = [C.POINTER(vp), C.POINTER(_PlanDesc)]
        L.flex_plan_create_ld.argtypes = [C.POINTER(vp), C.POINTER(_Csr), i32, i32, i32, i32, u32]
        L.flex_plan_create_mapped.argtypes = [C.POINTER(vp), C.POINTER(_Csr), vp, i32, i32, u32]
        L.flex_plan_create_rows.argtypes = [C.POINTER(vp), C.POINTER(_Csr), i64, i64, vp, i32, i32, u32]
        L.flex_spmm.argtypes = [vp, vp, vp, vp]
        L.flex_plan_destroy.argtypes = [vp]
        L.flex_plan_get_info.argtypes = [vp, C.POINTER(_PlanInfo)]
        L.flex_plan_get_stats.argtypes = [vp, C.POINTER(_PlanStats)]
        L.flex_plan_get_tuning.argtypes = [vp, C.POINTER(_PlanTuning)]
        L.flex_set_host_threads.argtypes = [i32]
        L.flex_order_cluster_ex.argtypes = [C.POINTER(_Csr), C.POINTER(_ClusterTuning), vp]
        L.flex_plan_self_check.argtypes = [vp]
        L.flex_plan_measure_imbalance.argtypes = [vp, vp, vp, vp, C.POINTER(_Imbalance)]
        L.flex_plan_kernel_info.argtypes = [vp, C.POINTER(_KernelInfo)]
        L.flex_gather_rows.argtypes = [vp, vp, vp, i64, i32, vp]
        L.flex_hbm_probe.argtypes = [i32, i64, i32, i32, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.flex_csv_load.argtypes = [C.c_char_p, C.POINTER(_HostCsr)]
        L.flex_mtx_load.argtypes = [C.c_char_p, i32, C.POINTER(_HostCsr)]
        L.flex_csv_save.argtypes = [C.c_char_p, C.POINTER(_Csr)]
        L.flex_csr_fingerprint.argtypes = [C.POINTER(_Csr)]
        L.flex_csr_fingerprint.restype = C.c_uint64
        L.flex_perm_save.argtypes = [C.c_char_p, vp, i64, C.c_uint64]
        L.flex_perm_load.argtypes = [C.c_char_p, vp, i64, C.c_uint64]
        L.flex_csr_save_bin.argtypes = [C.c_char_p, C.POINTER(_Csr)]
        L.flex_csr_load_bin.argtypes = [C.c_char_p, C.POINTER(_HostCsr)]
        L.flex_host_csr_free.argtypes = [C.POINTER(_HostCsr)]
        L.flex_host_csr_free.restype = None
        L.flex_fill_dense_rand.argtypes = [vp, i64, i32]
        L.flex_order_rcm.argtypes = [C.POINTER(_Csr), vp]
        L.flex_order_cluster.argtypes = [C.POINTER(_Csr), vp]
        L.flex_order_gorder.argtypes = [C.POINTER(_Csr), u32, vp]
        L.flex_order_dfs.argtypes = [C.POINTER(_Csr), vp]
        L.flex_order_rabbit.argtypes = [C.POINTER(_Csr), i32, vp]
        L.flex_order_deg.argtypes = [C.POINTER(_Csr), i32, vp]
        L.flex_synth_preset.argtypes = [C.c_char_p, i32, C.POINTER(_SynthParams)]
        L.flex_perm_csr.argtypes = [C.POINTER(_Csr), vp, vp, vp, vp, vp]
        L.flex_shard_rows.argtypes = [C.POINTER(_Csr), i32, i32, vp]
        L.flex_synth_graph.argtypes = [C.POINTER(_SynthParams), C.POINTER(_HostCsr)]
        L.flex_strerror.argtypes = [i32]
        L.flex_strerror.restype = C.c_char_p
        L.flex_last_hip_error.restype = i32
        L.flex_last_hip_error_string.restype = C.c_char_p
        L.flex_abi_version.restype = i32
        _lib = L
    return _lib


def _check(rc: int, what: str):
    if rc != 0:
        L = lib()
        msg = L.flex_strerror(rc).decode()
        if rc == -3:
            msg += f" [hip {L.flex_last_hip_error()}: {L.flex_last_hip_error_string().decode()}]"
        raise FlexError(f"{what}: {msg} ({rc})")


class HostCsr:
    """Host CSR (numpy, uint32 indices / float32 values) + the DataLoader statistics."""

    def __init__(self, rowPtr, col, vals, n=None, **stats):
        self.rowPtr = np.ascontiguousarray(rowPtr, dtype=np.uint32)
        self.col = np.ascontiguousarray(col, dtype=np.uint32)
        self.vals = np.ascontiguousarray(vals, dtype=np.float32)
        self.m = len(self.rowPtr) - 1
        self.n = self.m if n is None else int(n)
        self.nnz = len(self.col)
        self.__dict__.update(stats)

    def view(self) -> _Csr:
        return _Csr(self.m, self.n, self.nnz, self.rowPtr.ctypes.data, self.col.ctypes.data,
                    self.vals.ctypes.data)


def _take(s: _HostCsr) -> HostCsr:
    n = s.n
    try:
        rp = np.ctypeslib.as_array(s.rowPtr, shape=(s.m + 1,)).copy()
        col = np.ctypeslib.as_array(s.col, shape=(max(s.nnz, 1),))[: s.nnz].copy()
        vals = np.ctypeslib.as_array(s.vals, shape=(max(s.nnz, 1),))[: s.nnz].copy()
        stats = {k: getattr(s, k) for k in ("uni_nb", "n_edges_one_way", "n_edges_asymmetric",
                                           "n_nodes_z_out", "n_nodes_z_in", "n_nodes_z_deg",
                                           "is_directed", "c")}
    finally:
        lib().flex_host_csr_free(C.byref(s))
    return HostCsr(rp, col, vals, n=n, **stats)


def csv_load(path: str) -> HostCsr:
    s = _HostCsr()
    _check(lib().flex_csv_load(os.fsencode(path), C.byref(s)), f"flex_csv_load({path})")
    return _take(s)


def mtx_load(path: str, sort_columns: bool = True) -> HostCsr:
    s = _HostCsr()
    _check(lib().flex_mtx_load(os.fsencode(path), int(sort_columns), C.byref(s)), f"flex_mtx_load({path})")
    return _take(s)


def csv_save(path: str, a: HostCsr):
    v = a.view()
    _check(lib().flex_csv_save(os.fsencode(path), C.byref(v)), f"flex_csv_save({path})")


def csr_save_bin(path: str, a: HostCsr):
    v = a.view()
    _check(lib().flex_csr_save_bin(os.fsencode(path), C.byref(v)), f"flex_csr_save_bin({path})")


def csr_load_bin(path: str) -> HostCsr:
    s = _HostCsr()
    _check(lib().flex_csr_load_bin(os.fsencode(path), C.byref(s)), f"flex_csr_load_bin({path})")
    return _take(s)


def hbm_probe(device: int = 0, mib: int = 2048, reps: int = 10, temporal: bool = False) -> dict:
    """Measured GB/s of a read-only stream and of a copy (read + write bytes) on `device`."""
    r, c = C.c_double(), C.c_double()
    _check(lib().flex_hbm_probe(device, mib << 20, reps, int(temporal), C.byref(r), C.byref(c)), "flex_hbm_probe")
    return {"read_GBps": r.value, "copy_GBps": c.value}


def csr_fingerprint(a: HostCsr) -> int:
    v = a.view()
    return int(lib().flex_csr_fingerprint(C.byref(v)))


def perm_save(path: str, rank, fingerprint: int):
    r = np.ascontiguousarray(rank, dtype=np.uint32)
    _check(lib().flex_perm_save(os.fsencode(path), r.ctypes.data, len(r), fingerprint), f"flex_perm_save({path})")


def perm_load(path: str, n: int, fingerprint: int) -> np.ndarray:
    rank = np.empty(max(n, 1), dtype=np.uint32)
    _check(lib().flex_perm_load(os.fsencode(path), rank.ctypes.data, n, fingerprint), f"flex_perm_load({path})")
    return rank[:n]


def fill_dense_rand(n: int, k: int) -> np.ndarray:
    B = np.empty((n, k), dtype=np.float32)
    _check(lib().flex_fill_dense_rand(B.ctypes.data, n, k), "flex_fill_dense_rand")
    return B


def order_rcm(a: HostCsr) -> np.ndarray:
    rank = np.empty(max(a.m, 1), dtype=np.uint32)
    v = a.view()
    _check(lib().flex_order_rcm(C.byref(v), rank.ctypes.data), "flex_order_rcm")
    return rank[: a.m]


def order_cluster(a: HostCsr, **knobs) -> np.ndarray:
    """The engine's community order (agglomeration + vertex moves, cluster.cpp): rank[old] = new.
    knobs: fields of flex_cluster_tuning (batch, no_refine, stretch, sweeps, stride)."""
    rank = np.empty(max(a.m, 1), dtype=np.uint32)
    v = a.view()
    t = _ClusterTuning()
    for key, val in knobs.items():
        if key not in CLUSTER_TUNING_FIELDS:
            raise FlexError(f"unknown cluster knob {key!r}")
        setattr(t, key, int(val))
    _check(lib().flex_order_cluster_ex(C.byref(v), C.byref(t), rank.ctypes.data), "flex_order_cluster_ex")
    return rank[: a.m]


def set_host_threads(n: int) -> int:
    """Process-wide cap on the planner's / orderings' / generator's worker threads (0 = core count); returns the old value."""
    return int(lib().flex_set_host_threads(int(n)))


def perm_csr(a: HostCsr, rank: np.ndarray):
    """DataLoaderRcm body: returns (vo_mp, permuted HostCsr)."""
    rank = np.ascontiguousarray(rank, dtype=np.uint32)
    vo = np.empty(max(a.m, 1), dtype=np.int32)
    rp2 = np.empty(a.m + 1, dtype=np.uint32)
    c2 = np.empty(max(a.nnz, 1), dtype=np.uint32)
    v2 = np.empty(max(a.nnz, 1), dtype=np.float32)
    v = a.view()
    _check(lib().flex_perm_csr(C.byref(v), rank.ctypes.data, vo.ctypes.data, rp2.ctypes.data,
                               c2.ctypes.data, v2.ctypes.data), "flex_perm_csr")
    return vo[: a.m], HostCsr(rp2, c2[: a.nnz], v2[: a.nnz], n=a.n)


def shard_rows(a: HostCsr, k: int, nparts: int) -> np.ndarray:
    bounds = np.empty(nparts + 1, dtype=np.int64)
    v = a.view()
    _check(lib().flex_shard_rows(C.byref(v), k, nparts, bounds.ctypes.data), "flex_shard_rows")
    return bounds


SYNTH_PRESETS = ("amazon", "flickr", "ppi", "pubmed", "reddit", "soc-sign-epinions", "wiki-vote", "yelp")


def synth_preset(name: str, scale: int = 1) -> _SynthParams:
    """Generator parameters of the stand-in for a README / SuiteSparse graph (flex_synth_preset)."""
    p = _SynthParams()
    _check(lib().flex_synth_preset(name.lower().encode(), int(scale), C.byref(p)), f"flex_synth_preset({name})")
    return p


def synth_graph(name: str | None = None, *, scale: int = 1, n=None, nnz=None, alpha=2.1, community=0, p_in=0.0,
                p_near=0.0, near_window=8, shuffle=True, gcn_norm=True, directed=False, seed=0xF1E0) -> HostCsr:
    if name is not None:
        p = synth_preset(name, scale)
        p.shuffle = int(bool(shuffle))
    else:
        p = _SynthParams(int(n), int(nnz), float(alpha), int(community), float(p_in), float(p_near),
                         int(near_window), int(bool(shuffle)), int(bool(gcn_norm)), int(bool(directed)), int(seed))
    s = _HostCsr()
    _check(lib().flex_synth_graph(C.byref(p), C.byref(s)), f"flex_synth_graph({name or n})")
    return _take(s)


def order_gorder(a: HostCsr, window: int = 3) -> np.ndarray:
    rank = np.empty(max(a.m, 1), dtype=np.uint32)
    v = a.view()
    _check(lib().flex_order_gorder(C.byref(v), int(window), rank.ctypes.data), "flex_order_gorder")
    return rank[: a.m]


def order_dfs(a: HostCsr) -> np.ndarray:
    rank = np.empty(max(a.m, 1), dtype=np.uint32)
    v = a.view()
    _check(lib().flex_order_dfs(C.byref(v), rank.ctypes.data), "flex_order_dfs")
    return rank[: a.m]


def order_rabbit(a: HostCsr, is_directed: bool | None = None) -> np.ndarray:
    """flex_order_rabbit: the reference's Rabbit order (DataLoader.cu:455-655); is_directed defaults to the loader's statistic."""
    if is_directed is None:
        is_directed = bool(getattr(a, "is_directed", 0))
    rank = np.empty(max(a.m, 1), dtype=np.uint32)
    v = a.view()
    _check(lib().flex_order_rabbit(C.byref(v), int(bool(is_directed)), rank.ctypes.data), "flex_order_rabbit")
    return rank[: a.m]


def order_deg(a: HostCsr, descending: bool = True) -> np.ndarray:
    rank = np.empty(max(a.m, 1), dtype=np.uint32)
    v = a.view()
    _check(lib().flex_order_deg(C.byref(v), int(descending), rank.ctypes.data), "flex_order_deg")
    return rank[: a.m]


class Plan:
    """flex_plan handle (≙ Mat after csr2_DiagTiling + alpha_transfer)."""

    def __init__(self, a: HostCsr, k: int, device: int = 0, order: int = FLEX_ORDER_NATURAL,
                 vo_mp=None, rows=None, col_map=None, ldb: int | None = None, ldc: int | None = None, tuning: dict | None = None):
        """tuning: plan-time knobs as a dict of flex_plan_tuning fields (0 / absent = the planner's rule), e.g.
        {"lanes_per_nz": 16, "split_rows": 1, "cluster_no_refine": 1}."""
        self._h = C.c_void_p()
        self._keep = (a, vo_mp, col_map)
        v = a.view()
        L = lib()
        tn = _tuning({**DEFAULT_TUNING, **(tuning or {})})
        if tn is not None or ((ldb is not None or ldc is not None) and (rows is not None or vo_mp is not None)):
            # a combination the named entry points do not cover: the general one
            cm = None if col_map is None else np.ascontiguousarray(col_map, dtype=np.int32)
            vm = None if vo_mp is None else np.ascontiguousarray(vo_mp, dtype=np.int32)
            self._keep = (a, cm, vm)
            d = _PlanDesc(C.sizeof(_PlanDesc), C.pointer(v), k, ldb or 0, ldc or 0, device,
                          order | (0 if rows is None else FLEX_PLAN_ROW_RANGE),
                          0 if rows is None else int(rows[0]), 0 if rows is None else int(rows[1]),
                          (cm if cm is not None else vm).ctypes.data if (cm is not None or vm is not None) else None,
                          None if vm is None else vm.ctypes.data, None if tn is None else C.pointer(tn))
            rc = L.flex_plan_create_ex(C.byref(self._h), C.byref(d))
        elif rows is not None:
            cm = None if col_map is None else np.ascontiguousarray(col_map, dtype=np.int32)
            self._keep = (a, cm)
            rc = L.flex_plan_create_rows(C.byref(self._h), C.byref(v), int(rows[0]), int(rows[1]),
                                         None if cm is None else cm.ctypes.data, k, device, order)
        elif vo_mp is not None:
            vm = np.ascontiguousarray(vo_mp, dtype=np.int32)
            rc = L.flex_plan_create_mapped(C.byref(self._h), C.byref(v), vm.ctypes.data, k, device, order)
        elif ldb is not None or ldc is not None:
            rc = L.flex_plan_create_ld(C.byref(self._h), C.byref(v), k, ldb or k, ldc or k, device, order)
        else:
            rc = L.flex_plan_create(C.byref(self._h), C.byref(v), k, device, order)
        _check(rc, "flex_plan_create")
        self.k = k
        self._keep = None  # the plan copies what it needs

    def info(self) -> dict:
        i = _PlanInfo()
        _check(lib().flex_plan_get_info(self._h, C.byref(i)), "flex_plan_get_info")
        return {f: getattr(i, f) for f, _ in _PlanInfo._fields_}

    def stats(self) -> dict:
        """flex_plan_stats (≙ alpha_stats_collect + B-Re1/B-Re2); the plan must be made with FLEX_PLAN_STATS."""
        st = _PlanStats()
        _check(lib().flex_plan_get_stats(self._h, C.byref(st)), "flex_plan_get_stats")
        return {f: getattr(st, f) for f, _ in _PlanStats._fields_}

    def tuning(self) -> dict:
        """flex_plan_get_tuning: the knobs this plan was built with, rules resolved."""
        t = _PlanTuning()
        _check(lib().flex_plan_get_tuning(self._h, C.byref(t)), "flex_plan_get_tuning")
        d = {f: getattr(t, f) for f in TUNING_FIELDS}
        d.update({"cluster_" + f: getattr(t.cluster, f) for f in CLUSTER_TUNING_FIELDS})
        return d

    def kernel_info(self) -> dict:
        ki = _KernelInfo()
        _check(lib().flex_plan_kernel_info(self._h, C.byref(ki)), "flex_plan_kernel_info")
        return {f: getattr(ki, f) for f, _ in _KernelInfo._fields_}

    def measure_imbalance(self, dB_ptr: int, dC_ptr: int, stream: int = 0) -> dict:
        """flex_plan_measure_imbalance (≙ the per-SM "Imb" column, flex.cu:5087-5126): one stamped launch, per-CU / per-XCD busy imbalance."""
        im = _Imbalance()
        _check(lib().flex_plan_measure_imbalance(self._h, dB_ptr, dC_ptr, stream, C.byref(im)), "flex_plan_measure_imbalance")
        return {f: getattr(im, f) for f, _ in _Imbalance._fields_}

    def self_check(self):
        """flex_plan_self_check: the device image of the plan is a partition of the work (raises FlexError if not)."""
        _check(lib().flex_plan_self_check(self._h), "flex_plan_self_check")

    def spmm(self, dB_ptr: int, dC_ptr: int, stream: int = 0):
        _check(lib().flex_spmm(self._h, dB_ptr, dC_ptr, stream), "flex_spmm")

    def __call__(self, B, out=None):
        """torch convenience: B is a cuda float32 [n,k] tensor; returns C [m,k]."""
        import torch
        i = self.info()
        assert B.is_cuda and B.dtype == torch.float32 and B.is_contiguous() and tuple(B.shape) == (i["n"], i["k"])
        if out is None:
            out = torch.empty((i["m"], i["k"]), dtype=torch.float32, device=B.device)
        self.spmm(B.data_ptr(), out.data_ptr(), torch.cuda.current_stream(B.device).cuda_stream)
        return out

    def destroy(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().flex_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def gather_rows(dst_ptr: int, src_ptr: int, idx_ptr: int, n: int, k: int, stream: int = 0):
    _check(lib().flex_gather_rows(dst_ptr, src_ptr, idx_ptr, n, k, stream), "flex_gather_rows")
