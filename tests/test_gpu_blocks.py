"""GPU parity tests of the hot-block path (the matrix split: nonzeros with reuse inside a block of rows are multiplied out of
LDS-staged B panels by spmm_hot_kernel after the flat kernel has done the rest; block_kernels.hip, block_plan.cpp) against the CPU
oracle, through the C ABI.  Same tolerance as everywhere: the reference's resCheck (flex.cu:4154-4213), zero mismatches."""
import numpy as np
import pytest

import flex_amd
import oracle
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


BLOCKS = {"blocks": 1}


@pytest.mark.parametrize("k", [128, 64, 100, 68, 256, 192, 32, 36, 4])
@pytest.mark.parametrize("order", [FLEX_ORDER_NATURAL, flex_amd.FLEX_ORDER_CLUSTER])
def test_hot_blocks_match_the_oracle(k, order):
    """A community graph (most nonzeros hot), default knobs: resCheck against the oracle, same bits on a second launch and
    from a second plan, and agreement with the flat plan of the same matrix within the tolerance.  k below one 64-column tile:
    the route is not taken (a flat plan, silently) and the result is as right."""
    g = flex_amd.synth_graph(n=20000, nnz=20000 + 2 * 500000, community=400, p_in=0.6, p_near=0.25, seed=21)
    B = random_B(g.n, k, 3)
    p = Plan(g, k, order=order, tuning=BLOCKS)
    i = p.info()
    if k >= 64:
        assert i["n_blocks"] >= 40 and i["block_rows"] > 0.9 * g.m and i["n_records"] < g.nnz  # the hot nonzeros left the flat stream
        if order == flex_amd.FLEX_ORDER_CLUSTER:
            assert i["block_hot_nnz"] > 0.4 * g.nnz
    else:
        assert i["n_blocks"] == 0
    p.self_check()
    C1 = run_plan(p, B)
    assert_matches_oracle(g, B, C1)
    assert np.array_equal(C1, run_plan(p, B))
    assert np.array_equal(C1, run_plan(Plan(g, k, order=order, tuning=BLOCKS), B))
    assert oracle.rescheck(run_plan(Plan(g, k, order=order, tuning={"blocks": 2}), B), C1, g.rowPtr)[0] == 0


@pytest.mark.parametrize("rounds,panel_rows,thr,cap", [(2, 64, 2, 40), (4, 128, 3, 24), (8, 196, 4, 0), (4, 8, 2, 16), (4, 200, 1, 0), (8, 200, 1000000, 0)])
def test_hot_blocks_over_the_knobs(rounds, panel_rows, thr, cap):
    """Every shape of the block image: 2..8 rows per slot, tiny panels (many panels and barriers, the panel budget overflowing into
    the flat plan), thr = 1 (every column staged: runs overflow their 16 steps into the flat plan), a threshold nothing reaches (no
    block image at all), small caps (long rows hold no slot: hubs and their pieces stay with the flat kernel)."""
    a = random_csr(9000, 9000, 20, seed=31, long_rows={5: 8000, 77: 1200, 4000: 300, 8999: 150, 100: 90, 101: 41}, empty_frac=0.05)
    g = flex_amd.synth_graph(n=12000, nnz=12000 + 2 * 240000, community=300, p_in=0.6, p_near=0.25, seed=12)
    knobs = dict(BLOCKS, block_rounds=rounds, block_panel_rows=panel_rows, block_thr=thr, block_cap=cap)
    for mat, order in ((a, FLEX_ORDER_NATURAL), (g, flex_amd.FLEX_ORDER_CLUSTER)):
        for k in (128, 64, 72):
            B = random_B(mat.n, k, 5)
            p = Plan(mat, k, order=order, tuning=knobs)
            i = p.info()
            if thr == 1000000:
                assert i["n_blocks"] == 0 and i["block_hot_nnz"] == 0 and i["n_records"] >= mat.nnz
            elif mat is g:
                assert i["n_blocks"] > 0 and i["block_hot_nnz"] > 0
            p.self_check()
            C = run_plan(p, B)
            assert_matches_oracle(mat, B, C)
            assert np.array_equal(C, run_plan(p, B))


def test_hot_blocks_mapped_shards_strides_and_non_finite_values():
    g = flex_amd.synth_graph(n=16000, nnz=16000 + 2 * 300000, community=256, p_in=0.6, p_near=0.25, seed=6)
    k = 128
    B = random_B(g.n, k, 5)
    gold = oracle.spmm(g.rowPtr, g.col, g.vals, B, nthreads=8)
    # the reference's flow (permuted loader + vo_mp) and three row shards of it
    vo, gp = flex_amd.perm_csr(g, flex_amd.order_cluster(g))
    pm = Plan(gp, k, vo_mp=vo, tuning=BLOCKS)
    assert pm.info()["n_blocks"] > 0
    assert oracle.rescheck(gold, run_plan(pm, B), g.rowPtr)[0] == 0
    bounds = flex_amd.shard_rows(gp, k, 3)
    got = np.zeros_like(gold)
    for s in range(3):
        ps = Plan(gp, k, rows=(bounds[s], bounds[s + 1]), col_map=vo, tuning=BLOCKS)
        assert ps.info()["n_blocks"] > 0
        ps.self_check()
        got[vo[bounds[s]:bounds[s + 1]]] = run_plan(ps, B)
    assert oracle.rescheck(gold, got, g.rowPtr)[0] == 0
    # padded storage: the tail of every C row is left alone
    ldb, ldc, kk = 96, 68, 64
    Bs = random_B(g.n, ldb, 7)
    Cd = torch.full((g.m, ldc), 2.5, device="cuda")
    pl = Plan(g, kk, order=flex_amd.FLEX_ORDER_CLUSTER, ldb=ldb, ldc=ldc, tuning=BLOCKS)
    assert pl.info()["n_blocks"] > 0
    pl.spmm(dev(Bs).data_ptr(), Cd.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    Cs = Cd.cpu().numpy()
    assert np.all(Cs[:, kk:] == 2.5)
    assert_matches_oracle(g, np.ascontiguousarray(Bs[:, :kk]), np.ascontiguousarray(Cs[:, :kk]))
    # inf / NaN in B reach exactly the rows that reference them (padding records of the block image point at a row of zeros in LDS;
    # the flat plan's point at a column the row uses anyway)
    Bn = B.copy()
    bad = [3, 4000, 15999]
    Bn[bad[0], 5] = np.inf
    Bn[bad[1], :] = np.nan
    Bn[bad[2], 77] = -np.inf
    Cn = run_plan(Plan(g, k, order=flex_amd.FLEX_ORDER_CLUSTER, tuning=BLOCKS), Bn)
    rows = np.repeat(np.arange(g.m), np.diff(g.rowPtr.astype(np.int64)))
    touched = np.zeros(g.m, bool)
    touched[rows[np.isin(g.col, bad)]] = True
    assert np.all(np.isfinite(Cn[~touched])) and not np.all(np.isfinite(Cn[touched]))
    gold_n = oracle.spmm(g.rowPtr, g.col, g.vals, Bn, nthreads=8)
    bad_e = ~np.isfinite(gold_n)
    assert np.array_equal(~np.isfinite(Cn), bad_e) and np.array_equal(Cn[bad_e], gold_n[bad_e], equal_nan=True)  # +-inf stays +-inf, as in the oracle
    clean = ~touched
    assert oracle.rescheck(gold[clean], Cn[clean], np.concatenate([[0], np.cumsum(np.diff(g.rowPtr.astype(np.int64))[clean])]).astype(np.uint32))[0] == 0
    # unaligned operands (C and B one float off a 16-byte boundary): the fast kernels need the alignment; the generic pair -- the
    # flat part's and the hot blocks' -- takes over and the result is as right (round 3 refused such a launch)
    pu = Plan(g, k, order=flex_amd.FLEX_ORDER_CLUSTER, tuning=BLOCKS)
    assert pu.info()["n_blocks"] > 0
    Cu = torch.empty(g.m * k + 1, device="cuda")[1:]
    Bu = torch.empty(g.n * k + 1, device="cuda")[1:]
    Bu.copy_(dev(B).reshape(-1))
    for b_ptr, c_t in ((dev(B).data_ptr(), Cu), (Bu.data_ptr(), torch.empty(g.m * k, device="cuda")), (Bu.data_ptr(), Cu)):
        c_t.fill_(float("nan"))
        pu.spmm(b_ptr, c_t.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert oracle.rescheck(gold, c_t.cpu().numpy().reshape(g.m, k), g.rowPtr)[0] == 0


def test_hot_blocks_are_stable_under_repetition():
    """Barriers and LDS hand-offs between the loader wave and the consumers: 200 launches under uneven load, same bits."""
    g = flex_amd.synth_graph(n=60000, nnz=60000 + 2 * 2400000, community=1024, p_in=0.6, p_near=0.25, seed=8)
    k = 128
    Bn = random_B(g.n, k, 92)
    B = dev(Bn)
    p = Plan(g, k, order=flex_amd.FLEX_ORDER_CLUSTER, tuning=BLOCKS)
    assert p.info()["block_panels"] > 2 * p.info()["n_blocks"]
    ref = p(B).clone()
    torch.cuda.synchronize()
    assert_matches_oracle(g, Bn, ref.cpu().numpy(), nthreads=8)
    C = torch.empty_like(ref)
    filler = torch.empty(64 << 20, device="cuda")
    for it in range(200):
        C.fill_(float("nan"))
        if it % 3 == 0:
            filler.add_(1.0)
        p(B, out=C)
        if it % 25 == 24 or it < 4:
            torch.cuda.synchronize()
            assert torch.equal(C, ref), f"launch {it}: result changed"
    torch.cuda.synchronize()
    assert torch.equal(C, ref)


def test_hot_blocks_in_a_hip_graph_and_on_two_streams():
    """flex_spmm with a block plan is two kernel launches (the flat kernel, then the hot blocks; + the fix-up when the flat part has split rows),
    no allocation, no host sync: it can be captured in a hipGraph; and two block plans on two streams do not disturb each other."""
    g = flex_amd.synth_graph(n=20000, nnz=20000 + 2 * 500000, community=400, p_in=0.6, p_near=0.25, seed=22)
    a = random_csr(9000, 9000, 20, seed=33, long_rows={5: 8000, 77: 1200}, empty_frac=0.05)
    k = 128
    Bg, Ba = random_B(g.n, k, 1), random_B(a.n, k, 2)
    pg = Plan(g, k, order=flex_amd.FLEX_ORDER_CLUSTER, tuning=BLOCKS)
    pa = Plan(a, k, tuning=dict(BLOCKS, block_cap=30))  # rows beyond 30 nonzeros are spread over several slots
    assert pg.info()["n_blocks"] > 0 and pa.info()["n_blocks"] > 0 and pa.info()["n_split_rows"] > 0
    dg, da = dev(Bg), dev(Ba)
    Cg = torch.zeros((g.m, k), device="cuda")
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        pg(dg, out=Cg)  # warm-up outside capture
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        pg(dg, out=Cg)
    Cg.zero_()
    graph.replay()
    torch.cuda.synchronize()
    ref_g = Cg.cpu().numpy()
    assert_matches_oracle(g, Bg, ref_g)
    ref_a = run_plan(pa, Ba)
    assert_matches_oracle(a, Ba, ref_a)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    Ca = torch.empty((a.m, k), device="cuda")
    torch.cuda.synchronize()
    for _ in range(30):
        pg.spmm(dg.data_ptr(), Cg.data_ptr(), s1.cuda_stream)
        pa.spmm(da.data_ptr(), Ca.data_ptr(), s2.cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(Cg.cpu().numpy(), ref_g) and np.array_equal(Ca.cpu().numpy(), ref_a)
