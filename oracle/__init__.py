"""ctypes front-end to the CPU oracle (oracle/flex_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under flex_amd/ imports this package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = [os.path.join(_HERE, f) for f in ("flex_oracle.c", "flex_oracle.h", "Makefile")]
    stale = force or not os.path.exists(_SO) or any(
        os.path.getmtime(s) > os.path.getmtime(_SO) for s in src)
    if stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return _SO


class _Csr(C.Structure):
    _fields_ = [("m", C.c_int64), ("n", C.c_int64), ("nnz", C.c_int64),
                ("rowPtr", C.POINTER(C.c_uint32)), ("col", C.POINTER(C.c_uint32)),
                ("vals", C.POINTER(C.c_float)),
                ("uni_nb", C.c_int64), ("n_edges_one_way", C.c_int64),
                ("n_edges_asymmetric", C.c_int64),
                ("n_nodes_z_out", C.c_int32), ("n_nodes_z_in", C.c_int32),
                ("n_nodes_z_deg", C.c_int32), ("is_directed", C.c_int32), ("c", C.c_int32)]


def lib():
    global _lib
    if _lib is None:
        try:
            build()
        except Exception:
            if not os.path.exists(_SO):
                raise
        L = C.CDLL(_SO)
        u32p, f32p, u64p, i32p = (C.POINTER(C.c_uint32), C.POINTER(C.c_float),
                                  C.POINTER(C.c_uint64), C.POINTER(C.c_int32))
        L.oracle_csv_load.argtypes = [C.c_char_p, C.POINTER(_Csr), C.c_int]
        L.oracle_csv_load.restype = C.c_int
        L.oracle_csr_free.argtypes = [C.POINTER(_Csr)]
        L.oracle_gen_B.argtypes = [C.c_int64, C.c_int, f32p, C.c_int]
        L.oracle_spmm.argtypes = [C.c_int64, u32p, u32p, f32p, f32p, f32p, C.c_int]
        L.oracle_spmm_mt.argtypes = [C.c_int64, u32p, u32p, f32p, f32p, f32p, C.c_int, C.c_int]
        L.oracle_rescheck.argtypes = [f32p, f32p, u32p, C.c_int64, C.c_int,
                                      C.POINTER(C.c_double), i32p, C.POINTER(C.c_int64)]
        L.oracle_rescheck.restype = C.c_int64
        L.oracle_order_rcm.argtypes = [C.c_int64, u32p, u32p, u64p]
        L.oracle_order_rcm.restype = C.c_int
        L.oracle_order_gorder.argtypes = [C.c_int64, u32p, u32p, C.c_uint64, u64p]
        L.oracle_order_gorder.restype = C.c_int
        L.oracle_order_dfs.argtypes = [C.c_int64, u32p, u32p, u64p]
        L.oracle_order_dfs.restype = C.c_int
        L.oracle_order_rabbit.argtypes = [C.c_int64, u32p, u32p, C.c_int, u64p]
        L.oracle_order_rabbit.restype = C.c_int
        L.oracle_mtx_load.argtypes = [C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                      C.POINTER(u32p), C.POINTER(u32p), C.POINTER(f32p)]
        L.oracle_mtx_load.restype = C.c_int
        L.oracle_free.argtypes = [C.c_void_p]
        L.oracle_perm_csr.argtypes = [C.c_int64, u32p, u32p, f32p, u64p, i32p, u32p, u32p, f32p]
        _lib = L
    return _lib


def _p(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class Csr:
    """Host CSR with the reference DataLoader's statistics."""

    def __init__(self, rowPtr, col, vals, **stats):
        self.rowPtr, self.col, self.vals = _u32(rowPtr), _u32(col), _f32(vals)
        self.m = self.n = len(self.rowPtr) - 1
        self.nnz = len(self.col)
        self.__dict__.update(stats)


def csv_load(path: str, reset_rand: bool = True) -> Csr:
    s = _Csr()
    rc = lib().oracle_csv_load(os.fsencode(path), C.byref(s), int(reset_rand))
    if rc != 0:
        raise ValueError(f"oracle_csv_load({path}) failed: {rc}")
    try:
        rp = np.ctypeslib.as_array(s.rowPtr, shape=(s.m + 1,)).copy()
        col = np.ctypeslib.as_array(s.col, shape=(max(s.nnz, 1),))[: s.nnz].copy()
        vals = np.ctypeslib.as_array(s.vals, shape=(max(s.nnz, 1),))[: s.nnz].copy()
        stats = {k: getattr(s, k) for k in ("uni_nb", "n_edges_one_way", "n_edges_asymmetric",
                                           "n_nodes_z_out", "n_nodes_z_in", "n_nodes_z_deg",
                                           "is_directed", "c")}
    finally:
        lib().oracle_csr_free(C.byref(s))
    return Csr(rp, col, vals, **stats)


def gen_B(n: int, k: int, reset_rand: bool = True) -> np.ndarray:
    B = np.empty((n, k), dtype=np.float32)
    lib().oracle_gen_B(n, k, _p(B, C.c_float), int(reset_rand))
    return B


def spmm(rowPtr, col, vals, B, nthreads: int = 1) -> np.ndarray:
    rowPtr, col, vals, B = _u32(rowPtr), _u32(col), _f32(vals), _f32(B)
    m, k = len(rowPtr) - 1, B.shape[1]
    Cm = np.empty((m, k), dtype=np.float32)
    if nthreads > 1:
        lib().oracle_spmm_mt(m, _p(rowPtr, C.c_uint32), _p(col, C.c_uint32), _p(vals, C.c_float),
                             _p(B, C.c_float), _p(Cm, C.c_float), k, nthreads)
    else:
        lib().oracle_spmm(m, _p(rowPtr, C.c_uint32), _p(col, C.c_uint32), _p(vals, C.c_float),
                          _p(B, C.c_float), _p(Cm, C.c_float), k)
    return Cm


def rescheck(gold, res, orig_rowPtr):
    """resCheck (flex.cu:4154-4213): returns (mismatches, max_err, nnz of that row, #gold zeros)."""
    gold, res, rp = _f32(gold), _f32(res), _u32(orig_rowPtr)
    m, k = gold.shape
    assert res.shape == gold.shape and len(rp) == m + 1
    me, mn, gz = C.c_double(), C.c_int32(), C.c_int64()
    cnt = lib().oracle_rescheck(_p(gold, C.c_float), _p(res, C.c_float), _p(rp, C.c_uint32), m, k,
                                C.byref(me), C.byref(mn), C.byref(gz))
    return int(cnt), me.value, mn.value, gz.value


def order_rcm(rowPtr, col) -> np.ndarray:
    rowPtr, col = _u32(rowPtr), _u32(col)
    n = len(rowPtr) - 1
    rank = np.empty(max(n, 1), dtype=np.uint64)
    rc = lib().oracle_order_rcm(n, _p(rowPtr, C.c_uint32), _p(col, C.c_uint32), _p(rank, C.c_uint64))
    if rc:
        raise RuntimeError(f"oracle_order_rcm failed: {rc}")
    return rank[:n]


def order_gorder(rowPtr, col, window: int = 3) -> np.ndarray:
    rowPtr, col = _u32(rowPtr), _u32(col)
    n = len(rowPtr) - 1
    rank = np.empty(max(n, 1), dtype=np.uint64)
    rc = lib().oracle_order_gorder(n, _p(rowPtr, C.c_uint32), _p(col, C.c_uint32), window, _p(rank, C.c_uint64))
    if rc:
        raise RuntimeError(f"oracle_order_gorder failed: {rc}")
    return rank[:n]


def order_rabbit(rowPtr, col, is_directed: bool) -> np.ndarray:
    """DataLoaderRabbit (DataLoader.cu:455-655), rank[old] = new."""
    rowPtr, col = _u32(rowPtr), _u32(col)
    n = len(rowPtr) - 1
    rank = np.empty(max(n, 1), dtype=np.uint64)
    rc = lib().oracle_order_rabbit(n, _p(rowPtr, C.c_uint32), _p(col, C.c_uint32), int(bool(is_directed)), _p(rank, C.c_uint64))
    if rc:
        raise RuntimeError(f"oracle_order_rabbit failed: {rc}")
    return rank[:n]


def order_dfs(rowPtr, col) -> np.ndarray:
    rowPtr, col = _u32(rowPtr), _u32(col)
    n = len(rowPtr) - 1
    rank = np.empty(max(n, 1), dtype=np.uint64)
    rc = lib().oracle_order_dfs(n, _p(rowPtr, C.c_uint32), _p(col, C.c_uint32), _p(rank, C.c_uint64))
    if rc:
        raise RuntimeError(f"oracle_order_dfs failed: {rc}")
    return rank[:n]


def mtx_load(path: str):
    """mtx2csr.cc's mmio_allinone: returns (m, n, rowPtr, col, vals) in the reference's unsorted layout."""
    m, n, nnz = C.c_int64(), C.c_int64(), C.c_int64()
    rp, col, val = C.POINTER(C.c_uint32)(), C.POINTER(C.c_uint32)(), C.POINTER(C.c_float)()
    rc = lib().oracle_mtx_load(os.fsencode(path), C.byref(m), C.byref(n), C.byref(nnz), C.byref(rp), C.byref(col), C.byref(val))
    if rc:
        raise ValueError(f"oracle_mtx_load({path}) failed: {rc}")
    try:
        out = (m.value, n.value, np.ctypeslib.as_array(rp, shape=(m.value + 1,)).copy(),
               np.ctypeslib.as_array(col, shape=(max(nnz.value, 1),))[: nnz.value].copy(),
               np.ctypeslib.as_array(val, shape=(max(nnz.value, 1),))[: nnz.value].copy())
    finally:
        for p in (rp, col, val):
            lib().oracle_free(C.cast(p, C.c_void_p))
    return out


def perm_csr(rowPtr, col, vals, rank):
    """DataLoaderRcm body: returns (vo_mp, rowPtr2, col2, vals2)."""
    rowPtr, col, vals = _u32(rowPtr), _u32(col), _f32(vals)
    rank = np.ascontiguousarray(rank, dtype=np.uint64)
    n = len(rowPtr) - 1
    vo = np.empty(max(n, 1), dtype=np.int32)
    rp2 = np.empty(n + 1, dtype=np.uint32)
    c2 = np.empty(max(len(col), 1), dtype=np.uint32)
    v2 = np.empty(max(len(col), 1), dtype=np.float32)
    lib().oracle_perm_csr(n, _p(rowPtr, C.c_uint32), _p(col, C.c_uint32), _p(vals, C.c_float),
                          _p(rank, C.c_uint64), _p(vo, C.c_int32), _p(rp2, C.c_uint32),
                          _p(c2, C.c_uint32), _p(v2, C.c_float))
    return vo[:n], rp2, c2[: len(col)], v2[: len(col)]
