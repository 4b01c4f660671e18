"""Test helpers: seeded random CSR matrices and oracle-vs-HIP comparison."""
import numpy as np

import oracle
from flex_amd import HostCsr


def random_csr(m, n, avg_deg, seed, long_rows=(), empty_frac=0.1, sorted_cols=True):
    """Random CSR with some empty rows and optional very long rows {row: nnz}."""
    rng = np.random.default_rng(seed)
    deg = rng.poisson(avg_deg, size=m).astype(np.int64)
    deg[rng.random(m) < empty_frac] = 0
    for r, d in dict(long_rows).items():
        deg[r] = d
    deg = np.minimum(deg, n)
    rp = np.zeros(m + 1, dtype=np.int64)
    np.cumsum(deg, out=rp[1:])
    col = np.empty(rp[-1], dtype=np.uint32)
    for r in range(m):
        d = deg[r]
        if d == 0:
            continue
        if d * 4 > n:
            c = rng.permutation(n)[:d]
        else:
            c = np.unique(rng.integers(0, n, size=2 * d + 8))
            rng.shuffle(c)
            c = c[:d]
            while len(c) < d:  # extremely unlikely
                c = np.unique(np.concatenate([c, rng.integers(0, n, size=d)]))[:d]
        col[rp[r]:rp[r + 1]] = np.sort(c) if sorted_cols else c
    vals = rng.uniform(-1, 1, size=rp[-1]).astype(np.float32)
    return HostCsr(rp.astype(np.uint32), col, vals, n=n)


def random_B(n, k, seed):
    return np.random.default_rng(seed).uniform(-1, 1, size=(n, k)).astype(np.float32)


def assert_matches_oracle(a, B, C_hip, nthreads=4):
    """resCheck (flex.cu:4154-4213) of the HIP result against the CPU oracle: zero mismatches."""
    gold = oracle.spmm(a.rowPtr, a.col, a.vals, B, nthreads=nthreads)
    cnt, max_err, me_nnz, _ = oracle.rescheck(gold, C_hip, a.rowPtr)
    assert cnt == 0, f"{cnt} elements beyond 4*eps*row_nnz (max err {max_err:g} on a row of {me_nnz} nnz)"
    return gold, max_err


def vendor_spmm(a, k, B):
    """hipSPARSE CSR_ALG3 row-major SpMM through libflex_vendor.so (≙ cuSpmm, flex.cu:5717-5804): returns C (numpy)."""
    import ctypes as C
    import os

    import torch

    import flex_amd
    V = C.CDLL(os.path.join(os.path.dirname(flex_amd.lib_path()), "libflex_vendor.so"))
    V.flex_vendor_spmm_create.argtypes = [C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_int64, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    V.flex_vendor_spmm_run.argtypes = [C.c_void_p, C.c_void_p]
    V.flex_vendor_spmm_destroy.argtypes = [C.c_void_p]
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()  # noqa: E731
    rp, col, val = dev(a.rowPtr.astype(np.int32)), dev(a.col.astype(np.int32)), dev(a.vals)
    Bd = dev(B)
    Cd = torch.zeros((a.m, k), device="cuda")
    h = C.c_void_p()
    assert V.flex_vendor_spmm_create(C.byref(h), a.m, a.n, a.nnz, rp.data_ptr(), col.data_ptr(), val.data_ptr(), k,
                                     Bd.data_ptr(), Cd.data_ptr()) == 0
    assert V.flex_vendor_spmm_run(h, torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    V.flex_vendor_spmm_destroy(h)
    return Cd.cpu().numpy()
