"""GPU parity tests of ROW BUNDLES (plan_build.cpp, form_tasks; spmm_kernels.hip, flush): tasks that hold up to S = 64 / G short rows
side by side, slot s of every step working on row s (≙ the reference's narrow kernel giving every thread its own row,
flex.cu:81-118).  Through the C ABI, against the CPU oracle with the reference's resCheck tolerance."""
import os

import numpy as np
import pytest

import flex_amd
import oracle
from conftest import GOLDEN
from flex_amd import FLEX_ORDER_NATURAL, Plan
from util import assert_matches_oracle, random_B, random_csr

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def run_plan(plan, B):
    C = plan(dev(B))
    torch.cuda.synchronize()
    return C.cpu().numpy()


@pytest.mark.parametrize("k,lanes", [(4, 4), (16, 4), (8, 8), (32, 8), (64, 16), (128, 16), (128, 8), (256, 16), (20, 0)])
def test_bundles_against_the_oracle(k, lanes):
    """Every tile width that has bundles (four or more slots per step); a low-degree matrix with a third of its rows empty, hubs that are cut
    into pieces, and rows just around the candidate length; natural and community order; twice (bitwise equal)."""
    a = random_csr(7000, 7000, 5, seed=61, long_rows={11: 6000, 4000: 800, 5: 33, 6: 32, 7: 31}, empty_frac=0.33)
    B = random_B(a.n, k, 62)
    knobs = {"bundle": 1}
    if lanes:
        knobs["lanes_per_nz"] = lanes
    for order in (FLEX_ORDER_NATURAL, flex_amd.FLEX_ORDER_CLUSTER):
        p = Plan(a, k, order=order, tuning=knobs)
        info = p.info()
        assert info["n_bundles"] > 100 and info["bundle_rows"] > 0.6 * a.m and info["n_split_rows"] >= 2, info
        p.self_check()
        C = run_plan(p, B)
        gold, _ = assert_matches_oracle(a, B, C)
        assert np.array_equal(C, run_plan(p, B))
        plain = Plan(a, k, order=order, tuning={**knobs, "bundle": 2})
        assert plain.info()["n_bundles"] == 0
        assert oracle.rescheck(run_plan(plain, B), C, a.rowPtr)[0] == 0
    assert not C[np.diff(a.rowPtr.astype(np.int64)) == 0].any()  # rows without nonzeros are written as zeros


@pytest.mark.parametrize("k", [16, 32, 64, 128, 7])
def test_bundles_keep_non_finite_values_where_they_belong(k):
    """A slot sums ITS row only, its padding shares the row's last value, a row without nonzeros stores zeros whatever its slot
    gathered: inf stays inf (same sign), NaN is NaN, and no other row sees either -- also when row 0 of B, which the slots without a
    row gather, is itself non-finite."""
    a = random_csr(900, 900, 4, seed=71, empty_frac=0.25, long_rows={40: 700})
    B = random_B(900, k, 72)
    B[0, :] = np.inf
    B[17, :] = -np.inf
    B[18, min(5, k - 1)] = np.nan
    B[500, 0] = np.inf
    gold = oracle.spmm(a.rowPtr, a.col, a.vals, B)
    p = Plan(a, k, tuning={"bundle": 1, "lanes_per_nz": 16} if k > 64 else {"bundle": 1})
    assert p.info()["n_bundles"] > 20
    C = run_plan(p, B)
    assert np.array_equal(np.isfinite(C), np.isfinite(gold))
    fin = np.isfinite(gold)
    assert np.allclose(C[fin], gold[fin], rtol=1e-5, atol=1e-5)
    assert np.array_equal(C[~fin], gold[~fin], equal_nan=True)
    assert np.any(np.isinf(gold)) and np.any(np.isnan(gold))


def test_bundles_on_mapped_plans_shards_strides_and_unaligned_operands():
    a = flex_amd.synth_graph(n=9000, nnz=9000 + 2 * 27000, community=64, p_in=0.6, p_near=0.2, seed=81)
    k = 32
    B = random_B(a.n, k, 82)
    knobs = {"bundle": 1}
    gold, _ = assert_matches_oracle(a, B, run_plan(Plan(a, k, tuning=knobs), B))
    # the reference's flow: a reordered loader, vo_mp folded back in
    rank = flex_amd.order_cluster(a)
    vo, a2 = flex_amd.perm_csr(a, rank)
    pm = Plan(a2, k, vo_mp=vo, tuning=knobs)
    assert pm.info()["n_bundles"] > 0
    assert oracle.rescheck(gold, run_plan(pm, B), a.rowPtr)[0] == 0
    # row shards of the reordered matrix over padded storage
    bounds = flex_amd.shard_rows(a2, k, 3)
    ldb, ldc = k + 8, k + 4
    Bp = torch.zeros((a.n, ldb), device="cuda")
    Bp[:, :k] = dev(B)
    parts = []
    for i in range(3):
        ps = Plan(a2, k, rows=(bounds[i], bounds[i + 1]), col_map=vo, ldb=ldb, ldc=ldc, tuning=knobs)
        assert ps.info()["n_bundles"] > 0
        ps.self_check()
        out = torch.full((bounds[i + 1] - bounds[i], ldc), 7.0, device="cuda")
        ps.spmm(Bp.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert torch.all(out[:, k:] == 7.0)  # the padding of C is not touched
        parts.append(out[:, :k].cpu().numpy())
    shard_gold = oracle.spmm(a2.rowPtr, a2.col, a2.vals, B[vo])
    assert oracle.rescheck(shard_gold, np.concatenate(parts), a2.rowPtr)[0] == 0
    # operands that are not 16-byte aligned, and a k that is not a multiple of 4: the generic kernel walks a bundle row by row
    p = Plan(a, k, tuning=knobs)
    buf = torch.zeros(a.n * k + 1, device="cuda")
    buf[1:] = dev(B).ravel()
    out = torch.zeros(a.m * k + 1, device="cuda")
    p.spmm(buf[1:].data_ptr(), out[1:].data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert oracle.rescheck(gold, out[1:].reshape(a.m, k).cpu().numpy(), a.rowPtr)[0] == 0
    for k_odd in (7, 30):
        Bo = random_B(a.n, k_odd, 83)
        po = Plan(a, k_odd, tuning=knobs)
        assert po.info()["n_bundles"] > 0
        assert_matches_oracle(a, Bo, run_plan(po, Bo))
    # the wide tiles (one or two slots per step) have none
    pw = Plan(a, 128, tuning={**knobs, "lanes_per_nz": 32})
    assert pw.info()["n_bundles"] == 0 and pw.tuning()["bundle"] == 2


def test_bundles_on_pubmed_and_in_a_captured_graph():
    """The reference's own file (BASELINE configs[0]: pubmed.csv, k = 32) through bundles, launched from a hipGraph, 50 replays."""
    a = flex_amd.csv_load(os.path.join(GOLDEN, "pubmed.csv"))
    for k in (32, 128):
        Bn = random_B(a.n, k, 91)
        p = Plan(a, k, tuning={"bundle": 1, "lanes_per_nz": 8 if k == 32 else 16})
        assert p.info()["n_bundles"] > 0
        p.self_check()
        B = dev(Bn)
        C = torch.zeros((a.m, k), device="cuda")
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            p(B, out=C)  # warm-up outside the capture
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            p(B, out=C)
        first = None
        for _ in range(50):
            C.fill_(float("nan"))
            g.replay()
            torch.cuda.synchronize()
            got = C.cpu().numpy()
            first = got if first is None else first
            assert np.array_equal(first, got)
        assert_matches_oracle(a, Bn, first)
