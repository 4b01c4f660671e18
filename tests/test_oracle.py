"""CPU tests of the oracle itself: the survey-recorded numbers (not reference-held fixtures: parity stays unpinned),
the regression vectors, scipy / literal-loop cross-checks."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

import oracle
from conftest import GOLDEN


def _bandwidth(rp, col):
    rows = np.repeat(np.arange(len(rp) - 1), np.diff(rp.astype(np.int64)))
    return int(np.abs(rows - col.astype(np.int64)).max())


def test_pubmed_ingest_matches_reference_facts(pubmed):
    # README.md:15 and SURVEY.md section 4 (probed on the reference's host half)
    assert (pubmed.m, pubmed.nnz) == (19717, 108365)
    deg = np.diff(pubmed.rowPtr.astype(np.int64))
    assert deg.min() == 2 and deg.max() == 172
    assert not pubmed.is_directed and pubmed.n_edges_one_way == 0
    assert pubmed.n_nodes_z_deg == pubmed.n_nodes_z_in == pubmed.n_nodes_z_out == 0
    assert pubmed.c == 3  # DataLoader.cu:69-70
    # columns strictly sorted, diagonal present on every row
    for r in (0, 1, 100, 19716):
        cols = pubmed.col[pubmed.rowPtr[r]:pubmed.rowPtr[r + 1]]
        assert np.all(np.diff(cols.astype(np.int64)) > 0) and r in cols


def test_a_mat_ingest_trailing_comma(a_mat):
    # data/a_mat.csv: 48x48, 280 nnz, values 1/2, directed, trailing comma on the value line
    assert (a_mat.m, a_mat.nnz) == (48, 280)
    assert set(np.unique(a_mat.vals).tolist()) <= {1.0, 2.0}
    assert a_mat.is_directed
    assert a_mat.c == 100  # unknown name -> default, DataLoader.cu:81-84


def test_rand_B_matches_reference_prefix():
    # cpuX first values from the reference's host half, glibc rand() seed 1
    B = oracle.gen_B(4, 1)
    assert np.allclose(B.ravel(), [0.680375, -0.211234, 0.566198, 0.59688], atol=5e-7)


def test_pubmed_k32_checksum_matches_reference(pubmed):
    # SURVEY.md 8(c) / BASELINE.md section 3: fp64 sum of the fp32 C = 666.878358
    B = oracle.gen_B(pubmed.n, 32)
    Cm = oracle.spmm(pubmed.rowPtr, pubmed.col, pubmed.vals, B)
    assert abs(float(Cm.astype(np.float64).sum()) - 666.878358) < 5e-7


@pytest.mark.parametrize("k", [32, 128])
def test_pubmed_golden(pubmed, golden, k):
    B = oracle.gen_B(pubmed.n, k)
    assert np.array_equal(B.ravel()[:64], golden[f"pubmed_k{k}_B_prefix"])
    Cm = oracle.spmm(pubmed.rowPtr, pubmed.col, pubmed.vals, B)
    assert np.array_equal(Cm.ravel()[golden[f"pubmed_k{k}_idx"]], golden[f"pubmed_k{k}_C_at_idx"])
    assert float(Cm.astype(np.float64).sum()) == float(golden[f"pubmed_k{k}_C_sum"])
    assert np.array_equal(Cm.astype(np.float64).sum(axis=1), golden[f"pubmed_k{k}_C_rowsum"])


def test_a_mat_golden(a_mat, golden):
    B = oracle.gen_B(a_mat.n, 8)
    assert np.array_equal(B, golden["a_mat_k8_B"])
    assert np.array_equal(oracle.spmm(a_mat.rowPtr, a_mat.col, a_mat.vals, B), golden["a_mat_k8_C"])


def test_spmm_vs_scipy_fp64(pubmed):
    B = oracle.gen_B(pubmed.n, 16)
    Cm = oracle.spmm(pubmed.rowPtr, pubmed.col, pubmed.vals, B)
    A = sp.csr_matrix((pubmed.vals.astype(np.float64), pubmed.col.astype(np.int64),
                       pubmed.rowPtr.astype(np.int64)), shape=(pubmed.m, pubmed.n))
    ref = (A @ B.astype(np.float64)).astype(np.float32)
    cnt, max_err, _, zeros = oracle.rescheck(ref, Cm, pubmed.rowPtr)
    assert cnt == 0 and max_err < 1e-6 and zeros == 0


def test_spmm_mt_bit_identical(pubmed):
    B = oracle.gen_B(pubmed.n, 32)
    a = oracle.spmm(pubmed.rowPtr, pubmed.col, pubmed.vals, B)
    b = oracle.spmm(pubmed.rowPtr, pubmed.col, pubmed.vals, B, nthreads=4)
    assert np.array_equal(a, b)


def test_spmm_literal_python_loop(a_mat):
    # the 8-line loop of aspt/sspmm_128.cu:1415-1422, literally, on the toy matrix
    k = 4
    B = oracle.gen_B(a_mat.n, k)
    gold = np.zeros((a_mat.m, k), dtype=np.float32)
    rows = np.repeat(np.arange(a_mat.m), np.diff(a_mat.rowPtr.astype(np.int64)))
    for i in range(a_mat.nnz):
        for j in range(k):
            prod = np.float32(B[a_mat.col[i], j] * a_mat.vals[i])
            gold[rows[i], j] = np.float32(gold[rows[i], j] + prod)
    assert np.array_equal(gold, oracle.spmm(a_mat.rowPtr, a_mat.col, a_mat.vals, B))


def test_empty_rows_and_empty_matrix():
    rp = np.array([0, 0, 2, 2, 3], dtype=np.uint32)
    col = np.array([1, 3, 0], dtype=np.uint32)
    val = np.array([2.0, -1.0, 0.5], dtype=np.float32)
    B = np.arange(8, dtype=np.float32).reshape(4, 2)
    Cm = oracle.spmm(rp, col, val, B)
    assert np.array_equal(Cm, [[0, 0], [2 * 2 - 6, 2 * 3 - 7], [0, 0], [0, 0.5]])
    Z = oracle.spmm(np.zeros(4, np.uint32), np.zeros(0, np.uint32), np.zeros(0, np.float32),
                    np.ones((3, 2), np.float32))
    assert np.array_equal(Z, np.zeros((3, 2), np.float32))


def test_rescheck_semantics():
    rp = np.array([0, 1, 101], dtype=np.uint32)
    gold = np.array([[0.5, 2.0], [0.25, 4.0]], dtype=np.float32)
    eps = np.finfo(np.float32).eps
    res = gold.copy()
    assert oracle.rescheck(gold, res, rp)[0] == 0
    res[0, 0] += 8 * eps          # |g|<1: absolute error 8 eps > tol 4 eps (1 nz)
    res[1, 1] *= 1 + 100 * eps    # |g|>=1: relative error 100 eps < tol 400 eps (100 nz)
    cnt, max_err, me_nnz, zeros = oracle.rescheck(gold, res, rp)
    assert cnt == 1 and me_nnz == 100 and zeros == 0
    assert abs(max_err - 100 * eps) < 2 * eps
    res[1, 0] = np.nan
    assert oracle.rescheck(gold, res, rp)[0] == 2


def test_rcm_pubmed_matches_reference_bandwidth(pubmed, golden):
    # the reference's RCM takes pubmed's bandwidth 19482 -> 6241
    rank = oracle.order_rcm(pubmed.rowPtr, pubmed.col)
    assert sorted(rank.tolist()) == list(range(pubmed.n))
    vo, rp2, c2, v2 = oracle.perm_csr(pubmed.rowPtr, pubmed.col, pubmed.vals, rank)
    assert _bandwidth(pubmed.rowPtr, pubmed.col) == 19482
    assert _bandwidth(rp2, c2) == 6241
    assert np.array_equal(vo, golden["pubmed_rcm_vo_mp"])
    assert np.array_equal(rp2[:65], golden["pubmed_rcm_rowPtr_head"])
    assert np.array_equal(c2[:256], golden["pubmed_rcm_col_head"])
    # perm_apply's self-test (DataLoader.cu:294-320): per-column checksums survive the permutation
    inc_old = np.repeat(np.arange(pubmed.n) & 0xF, np.diff(pubmed.rowPtr.astype(np.int64)))
    inc_new = np.repeat(vo & 0xF, np.diff(rp2.astype(np.int64)))
    chk_old = np.bincount(pubmed.col, weights=inc_old, minlength=pubmed.n)
    chk_new = np.bincount(c2, weights=inc_new, minlength=pubmed.n)
    assert np.array_equal(chk_old, chk_new[rank.astype(np.int64)])
    # columns sorted ascending per row (DataLoader.cu:767)
    for r in (0, 5, 1000, pubmed.n - 1):
        assert np.all(np.diff(c2[rp2[r]:rp2[r + 1]].astype(np.int64)) > 0)


def test_rcm_permuted_spmm_equals_original(pubmed):
    # RCM-permuted SpMM un-permuted equals the original to ~3.6e-7
    k = 8
    B = oracle.gen_B(pubmed.n, k)
    C0 = oracle.spmm(pubmed.rowPtr, pubmed.col, pubmed.vals, B)
    rank = oracle.order_rcm(pubmed.rowPtr, pubmed.col)
    vo, rp2, c2, v2 = oracle.perm_csr(pubmed.rowPtr, pubmed.col, pubmed.vals, rank)
    C1 = oracle.spmm(rp2, c2, v2, B[vo])          # B' = B[vo_mp] (permuteX, flex.cu:276-289)
    C1_unperm = np.empty_like(C1)
    C1_unperm[vo] = C1                            # C[vo_mp[r']] = C'[r']
    cnt, max_err, _, _ = oracle.rescheck(C0, C1_unperm, pubmed.rowPtr)
    assert cnt == 0 and max_err < 1e-6


def test_csv_errors(tmp_path):
    bad = tmp_path / "bad.csv"
    bad.write_text("0,2\n0,1\n1.0\n")  # 2 cols, 1 val -> assert(col.size()==vals.size())
    with pytest.raises(ValueError):
        oracle.csv_load(str(bad))
    with pytest.raises(ValueError):
        oracle.csv_load(str(tmp_path / "missing.csv"))
    dup = tmp_path / "dup.csv"
    dup.write_text("0,2,2\n1,1\n1.0,2.0\n")  # duplicate edge -> assert(e_inv[dst].count(r)==0)
    with pytest.raises(ValueError):
        oracle.csv_load(str(dup))


def test_amazon_branch_uses_rand(tmp_path):
    # amazon.csv has no value line; vals = 2*rand()/RAND_MAX-1 BEFORE B is drawn (DataLoader.cu:36-46)
    f = tmp_path / "amazon.csv"
    f.write_text("0,1,2\n1,0\n")
    a = oracle.csv_load(str(f))
    assert np.allclose(a.vals, [0.680375, -0.211234], atol=5e-7) and a.c == 107
    B = oracle.gen_B(2, 1, reset_rand=False)      # continues the same rand() stream
    assert np.allclose(B.ravel(), [0.566198, 0.59688], atol=5e-7)


def test_gorder_is_a_permutation_and_improves_its_objective(pubmed, golden):
    # Gorder maximises, over a sliding window, the number of neighbour/sibling pairs placed close together
    r = oracle.order_gorder(pubmed.rowPtr, pubmed.col, 3)
    assert sorted(r.tolist()) == list(range(pubmed.n))
    assert np.array_equal(r.astype(np.int32), golden["pubmed_gorder_w3_rank"])
    rows = np.repeat(np.arange(pubmed.n), np.diff(pubmed.rowPtr.astype(np.int64)))

    def close_pairs(rank, w=3):
        d = np.abs(rank[rows].astype(np.int64) - rank[pubmed.col].astype(np.int64))
        return int(np.sum((d > 0) & (d <= w)))
    rcm = oracle.order_rcm(pubmed.rowPtr, pubmed.col)
    assert close_pairs(r) > 5 * close_pairs(rcm) > close_pairs(np.arange(pubmed.n))


def test_gorder_refuses_isolated_vertices():
    # the reference cannot complete on such a graph (unitheap.cu:35-38); the oracle reports it
    rp = np.array([0, 1, 1, 2], dtype=np.uint32)   # vertex 1 has no edge at all
    col = np.array([2, 0], dtype=np.uint32)
    with pytest.raises(RuntimeError):
        oracle.order_gorder(rp, col, 3)
