"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle.

Tolerance (north_star: "within a stated fp32 tolerance") is the reference's own resCheck
(flex.cu:4154-4213): per element err = |g|<1 ? |g-r| : |1-r/g| must not exceed
4 * FLT_EPSILON * nnz(row); zero elements may exceed it.
"""
import os

import numpy as np
import pytest

import flex_amd
import oracle
from conftest import GOLDEN
from flex_amd import FLEX_ORDER_NATURAL, FLEX_ORDER_RCM, Plan
from util import assert_matches_oracle, random_B, random_csr

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def run_plan(plan, B):
    C = plan(dev(B))
    torch.cuda.synchronize()
    return C.cpu().numpy()


def test_library_is_the_hip_one():
    assert torch.cuda.is_available()
    assert os.path.exists(flex_amd.lib_path())
    assert "gfx950" in torch.cuda.get_device_properties(0).gcnArchName


def test_a_mat_golden(golden):
    a = flex_amd.csv_load(os.path.join(GOLDEN, "a_mat.csv"))
    B = golden["a_mat_k8_B"]
    C = run_plan(Plan(a, 8), B)
    cnt, max_err, _, _ = oracle.rescheck(golden["a_mat_k8_C"], C, a.rowPtr)
    assert cnt == 0 and max_err < 1e-5


@pytest.mark.parametrize("k", [32, 128])
@pytest.mark.parametrize("order", [FLEX_ORDER_NATURAL, FLEX_ORDER_RCM, flex_amd.FLEX_ORDER_CLUSTER, flex_amd.FLEX_ORDER_GORDER])
def test_pubmed_vs_oracle_and_golden(golden, k, order):
    a = flex_amd.csv_load(os.path.join(GOLDEN, "pubmed.csv"))
    B = oracle.gen_B(a.n, k)  # the reference's rand() B (DataLoader.cu:198-209)
    C = run_plan(Plan(a, k, order=order), B)
    gold, max_err = assert_matches_oracle(a, B, C)
    # committed golden vectors: sampled elements and the fp64 checksum of the fp32 C
    idx = golden[f"pubmed_k{k}_idx"]
    deg = np.diff(a.rowPtr.astype(np.int64))[idx // k]
    g = golden[f"pubmed_k{k}_C_at_idx"]
    assert np.all(np.abs(C.ravel()[idx] - g) <= 4 * np.finfo(np.float32).eps * deg * np.maximum(1, np.abs(g)))
    assert abs(C.astype(np.float64).sum() - float(golden[f"pubmed_k{k}_C_sum"])) < 1e-2
    if k == 32:  # the number SURVEY.md 8(c) recorded from its probe of the reference's host half
        assert abs(C.astype(np.float64).sum() - 666.878358) < 1e-2


@pytest.mark.parametrize("k", [1, 3, 4, 8, 12, 16, 20, 32, 64, 96, 100, 128, 256, 260, 384, 512])
def test_k_sweep_ragged(k):
    a = random_csr(700, 900, 9, seed=k, long_rows={5: 800, 699: 600, 123: 513}, empty_frac=0.15)
    B = random_B(a.n, k, seed=1000 + k)
    p = Plan(a, k)
    info = p.info()
    assert info["n_split_rows"] >= 2 and info["n_partials"] >= 4
    assert_matches_oracle(a, B, run_plan(p, B))


def test_empty_and_degenerate():
    # all rows empty -> C must be overwritten with zeros (beta = 0)
    a = flex_amd.HostCsr(np.zeros(6, np.uint32), np.zeros(0, np.uint32), np.zeros(0, np.float32), n=7)
    p = Plan(a, 32)
    C = torch.full((5, 32), 7.0, device="cuda")
    p.spmm(dev(random_B(7, 32, 0)).data_ptr(), C.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.count_nonzero(C).item() == 0
    # zero rows
    z = flex_amd.HostCsr(np.zeros(1, np.uint32), np.zeros(0, np.uint32), np.zeros(0, np.float32), n=3)
    Plan(z, 16).spmm(0, 0, 0)
    # single dense row, single column matrix
    one = flex_amd.HostCsr(np.array([0, 1], np.uint32), np.array([0], np.uint32), np.array([2.0], np.float32), n=1)
    assert np.array_equal(run_plan(Plan(one, 4), np.array([[1, 2, 3, 4]], np.float32)), [[2, 4, 6, 8]])


def test_unsorted_columns_and_duplicates_are_summed():
    # mtx2csr.cc:171-195 emits unsorted columns; duplicates simply add up in CSR SpMM
    a = flex_amd.HostCsr(np.array([0, 4, 5], np.uint32), np.array([3, 0, 3, 1, 2], np.uint32),
                         np.array([1, 2, 3, 4, 5], np.float32), n=4)
    B = random_B(4, 32, 3)
    assert_matches_oracle(a, B, run_plan(Plan(a, 32), B))


@pytest.mark.parametrize("k", [128, 32, 16, 7])
def test_inf_nan_only_reach_rows_that_reference_them(k):
    """Non-finite B values reach exactly the rows that reference them, AS the oracle has them: +-inf stays +-inf (round 4: the
    padding records of a row share its last value instead of carrying 0, so no 0 x inf = NaN is added to a row's inf), NaN is NaN.
    Every tile width incl. the narrow one (k = 16) and the generic kernel (k = 7); rows cut into pieces included."""
    a = random_csr(300, 300, 7, seed=5, empty_frac=0.0, long_rows={40: 290})
    B = random_B(300, k, 6)
    B[17, :] = np.inf
    B[18, min(5, k - 1)] = np.nan
    B[200, 0] = -np.inf
    gold = oracle.spmm(a.rowPtr, a.col, a.vals, B)
    for tuning in (None, {"long_row": 64, "piece_records": 64}):
        C = run_plan(Plan(a, k, tuning=tuning), B)
        assert np.array_equal(np.isfinite(C), np.isfinite(gold))
        fin = np.isfinite(gold)
        assert np.allclose(C[fin], gold[fin], rtol=1e-5, atol=1e-5)
        assert np.array_equal(C[~fin], gold[~fin], equal_nan=True)  # inf where the oracle has inf (same sign), NaN where it has NaN
    assert np.any(np.isinf(gold)) and np.any(np.isnan(gold))


def test_rcm_schedule_and_mapped_plan_agree_with_natural():
    a = flex_amd.synth_graph(n=6000, nnz=6000 + 2 * 40000, community=64, p_in=0.6, p_near=0.2, seed=11)
    k = 64
    B = random_B(a.n, k, 12)
    gold, _ = assert_matches_oracle(a, B, run_plan(Plan(a, k), B))
    # (1) RCM as a schedule inside the plan: same B, C in original order
    assert_matches_oracle(a, B, run_plan(Plan(a, k, order=FLEX_ORDER_RCM), B))
    # (2) the reference's flow: DataLoaderRcm permutes the CSR, plan folds vo_mp back in
    rank = flex_amd.order_rcm(a)
    vo, a2 = flex_amd.perm_csr(a, rank)
    C2 = run_plan(Plan(a2, k, vo_mp=vo), B)
    cnt, _, _, _ = oracle.rescheck(gold, C2, a.rowPtr)
    assert cnt == 0
    # (3) the reference's literal flow with permuteX: B' = B[vo_mp], C' in RCM order
    Bd = dev(B)
    Bp = torch.empty_like(Bd)
    flex_amd.gather_rows(Bp.data_ptr(), Bd.data_ptr(), dev(vo).data_ptr(), a.n, k,
                         torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(Bp.cpu().numpy(), B[vo])
    C3 = run_plan(Plan(a2, k), B[vo])
    unperm = np.empty_like(C3)
    unperm[vo] = C3
    assert oracle.rescheck(gold, unperm, a.rowPtr)[0] == 0


def test_interleaved_xcd_dealing_for_a_reordered_loader():
    """FLEX_PLAN_XCD_INTERLEAVE (what the C++ host mirror passes for an RCM / Gorder loader): same result as the sliced plan."""
    a = flex_amd.synth_graph(n=30000, nnz=30000 + 2 * 400000, community=256, p_in=0.6, p_near=0.25, seed=12)
    B = random_B(a.n, 128, 3)
    vo, ap = flex_amd.perm_csr(a, flex_amd.order_rcm(a))
    sliced, dealt = Plan(ap, 128, vo_mp=vo), Plan(ap, 128, vo_mp=vo, order=flex_amd.FLEX_PLAN_XCD_INTERLEAVE)
    assert dealt.info()["n_slots"] == dealt.info()["n_chunks"] < sliced.info()["n_slots"]
    c1, c2 = run_plan(sliced, B), run_plan(dealt, B)
    assert np.array_equal(c1.view(np.uint32), c2.view(np.uint32))  # the same chunks, dealt differently
    assert_matches_oracle(a, B, c2)


def test_row_shards_concatenate_to_full():
    a = random_csr(5000, 5000, 20, seed=21, long_rows={7: 3000})
    k = 128
    B = random_B(a.n, k, 22)
    full = run_plan(Plan(a, k), B)
    bounds = flex_amd.shard_rows(a, k, 4)
    assert bounds[0] == 0 and bounds[-1] == a.m and np.all(np.diff(bounds) > 0)
    parts = [run_plan(Plan(a, k, rows=(bounds[i], bounds[i + 1])), B) for i in range(4)]
    # shards pick their own chunk budget, so a split row may be summed in a different order
    assert oracle.rescheck(full, np.concatenate(parts), a.rowPtr)[0] == 0
    assert_matches_oracle(a, B, full)
    assert_matches_oracle(a, B, np.concatenate(parts))


def test_unaligned_buffers_take_the_generic_kernel():
    a = random_csr(400, 500, 11, seed=31)
    k = 32
    B = random_B(a.n, k, 32)
    buf = torch.zeros(a.n * k + 1, device="cuda")
    buf[1:] = dev(B).ravel()
    out = torch.zeros(a.m * k + 1, device="cuda")
    Plan(a, k).spmm(buf[1:].data_ptr(), out[1:].data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert_matches_oracle(a, B, out[1:].reshape(a.m, k).cpu().numpy())


def test_deterministic_and_overwrites_output():
    a = random_csr(3000, 3000, 30, seed=41, long_rows={0: 2500, 1500: 2000})
    k = 128
    B = dev(random_B(a.n, k, 42))
    p = Plan(a, k)
    C1 = p(B)
    C2 = torch.full_like(C1, float("nan"))
    p(B, out=C2)
    torch.cuda.synchronize()
    assert torch.equal(C1, C2)


def test_hip_graph_capture():
    # flex_spmm allocates nothing and never syncs, so it can be captured (cdna guide G9)
    a = random_csr(2000, 2000, 15, seed=51, long_rows={3: 1200})
    k = 128
    Bn = random_B(a.n, k, 52)
    B = dev(Bn)
    C = torch.zeros((a.m, k), device="cuda")
    p = Plan(a, k)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        p(B, out=C)  # warm-up outside capture
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        p(B, out=C)
    C.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert_matches_oracle(a, Bn, C.cpu().numpy())


def test_flickr_shape_full_size_vs_oracle():
    # BASELINE config[1]: Flickr shape (89250^2, 989006 nnz), k=128 -- small enough for the oracle
    a = flex_amd.synth_graph("flickr")
    assert (a.m, a.nnz) == (89250, 989006)
    B = random_B(a.n, 128, 61)
    for order in (FLEX_ORDER_NATURAL, FLEX_ORDER_RCM):
        assert_matches_oracle(a, B, run_plan(Plan(a, 128, order=order), B), nthreads=8)


def test_large_properties_checksum_and_linearity():
    # size-independent checks at a size the scalar oracle would be slow on
    a = flex_amd.synth_graph(n=200000, nnz=200000 + 2 * 4000000, community=1024, p_in=0.6, p_near=0.2,
                             alpha=2.1, seed=71)
    k = 128
    p = Plan(a, k, order=FLEX_ORDER_RCM)
    # B = ones -> every column of C is the row sum of A (checksum of checksums)
    ones = torch.ones((a.n, k), device="cuda")
    C = p(ones).cpu().numpy()
    rows = np.repeat(np.arange(a.m), np.diff(a.rowPtr.astype(np.int64)))
    rowsum = np.bincount(rows, weights=a.vals.astype(np.float64), minlength=a.m)
    deg = np.diff(a.rowPtr.astype(np.int64))
    tol = 4 * np.finfo(np.float32).eps * np.maximum(deg, 1) * np.maximum(1.0, np.abs(rowsum))
    assert np.all(np.abs(C - rowsum[:, None]) <= tol[:, None])
    # linearity: A(B1 + B2) == A B1 + A B2 within the same tolerance scale
    B1, B2 = dev(random_B(a.n, k, 72)), dev(random_B(a.n, k, 73))
    lhs = p(B1 + B2).cpu().numpy().astype(np.float64)
    rhs = p(B1).cpu().numpy().astype(np.float64) + p(B2).cpu().numpy()
    assert np.all(np.abs(lhs - rhs) <= 8 * np.finfo(np.float32).eps * np.maximum(deg, 1)[:, None] * 2.0)
    # sampled rows against the oracle arithmetic
    samp = np.random.default_rng(7).choice(a.m, 2000, replace=False)
    B1h = B1.cpu().numpy()
    C1 = p(B1).cpu().numpy()
    for r in samp[:200]:
        cols = a.col[a.rowPtr[r]:a.rowPtr[r + 1]]
        v = a.vals[a.rowPtr[r]:a.rowPtr[r + 1]]
        ref = (v[:, None].astype(np.float64) * B1h[cols]).sum(axis=0)
        assert np.all(np.abs(C1[r] - ref) <= 4 * np.finfo(np.float32).eps * len(cols) * np.maximum(1, np.abs(ref)))


def test_cxx_host_mirror_cli_pubmed_and_amat(tmp_path):
    """The C++ DataLoader/Mat/run mirror (flex_amd/lib/flex): hipSPARSE gold + resCheck, every
    ordering must report zero mismatches (≙ assert(!count), flex.cu:4205)."""
    import json
    import subprocess
    exe = os.path.join(os.path.dirname(flex_amd.lib_path()), "flex")
    assert os.path.exists(exe)
    for path, k in ((os.path.join(GOLDEN, "pubmed.csv"), "32"), (os.path.join(GOLDEN, "a_mat.csv"), "8")):
        csv, log = str(tmp_path / "flex-tile-nperf.csv"), str(tmp_path / "flex-tile-stats2.log")
        out = subprocess.run([exe, path, k, "--json", "--stats", "--csv", csv, "--stats-log", log, "--perm-cache", str(tmp_path)],
                             capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stdout + out.stderr
        rows = [json.loads(line) for line in out.stdout.splitlines() if line.startswith("{")]
        assert {(r["ord"], r["schedule"]) for r in rows} >= {("OVO", "natural"), ("OVO", "cluster"), ("RCM", "natural"), ("RCM", "cluster"),
                                                             ("RBT", "natural"), ("DEG", "natural"), ("GOR", "natural"), ("DFS", "natural")}
        assert all(r["errs"] == 0 for r in rows)
        assert all(r["b_re1"] >= 1.0 and r["b_re2"] >= r["b_re1"] for r in rows)
    assert "hipSPARSE setup" in out.stdout and "B reuse: wave" in out.stdout
    # the AXW block of main.cu:22-77 (compiled out in the reference): both orders, compared with each other
    axw = subprocess.run([exe, os.path.join(GOLDEN, "pubmed.csv"), "32", "--axw", "--json"], capture_output=True, text=True, timeout=300)
    assert axw.returncode == 0, axw.stdout + axw.stderr
    assert axw.stdout.count("The results are correct..") == 10 and "A(XW):" in axw.stdout and "(AX)W:" in axw.stdout
    j = [json.loads(ln) for ln in axw.stdout.splitlines() if ln.startswith("{")][-1]
    assert j["c"] == 3 and j["dim"] == 32 and j["a_xw_ms"] > 0 and j["ax_w_ms"] > 0
    # opt_debug of the reference (A = 1, X[i][*] = i): still zero mismatches against the hipSPARSE gold
    dbg = subprocess.run([exe, os.path.join(GOLDEN, "a_mat.csv"), "8", "--json", "--debug-values"], capture_output=True, text=True, timeout=300)
    assert dbg.returncode == 0 and all(json.loads(ln)["errs"] == 0 for ln in dbg.stdout.splitlines() if ln.startswith("{"))
    # second run of the same graph: every ordering comes from the permutation cache and still checks out
    again = subprocess.run([exe, path, k, "--json", "--perm-cache", str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert again.returncode == 0 and "order: cached" not in out.stdout
    assert all(f"{o} order: cached" in again.stdout for o in ("RCM", "RBT", "DFS", "GOR", "DEG"))
    assert all(json.loads(ln)["errs"] == 0 for ln in again.stdout.splitlines() if ln.startswith("{"))
    # the report files of the reference's run(): appended CSV (graph name, header, one line per configuration) ...
    lines = open(csv).read().splitlines()
    assert lines.count("pubmed") == 1 and lines.count("a_mat") == 1
    body = [ln.split(",") for ln in lines if ln[:3] in ("OVO", "RCM", "RBT", "DFS", "GOR", "DEG")]
    assert len(body) == 2 * 9 and all(row[-1] == "0" for row in body)
    # ... and the per-plan statistics log (rewritten by each run)
    assert open(log).read().count("B reuse: wave") == 9


def test_vendor_baseline_matches_oracle():
    """libflex_vendor.so (hipSPARSE CSR_ALG3, row-major) is the reference's gold (cuSpmm,
    flex.cu:5717-5804): check it against the CPU oracle too."""
    import ctypes as C
    a = flex_amd.csv_load(os.path.join(GOLDEN, "pubmed.csv"))
    k = 32
    B = oracle.gen_B(a.n, k)
    V = C.CDLL(os.path.join(os.path.dirname(flex_amd.lib_path()), "libflex_vendor.so"))
    V.flex_vendor_spmm_create.argtypes = [C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_int64, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    V.flex_vendor_spmm_run.argtypes = [C.c_void_p, C.c_void_p]
    V.flex_vendor_spmm_destroy.argtypes = [C.c_void_p]
    rp, col, val = dev(a.rowPtr.astype(np.int32)), dev(a.col.astype(np.int32)), dev(a.vals)
    Bd = dev(B)
    Cd = torch.zeros((a.m, k), device="cuda")
    h = C.c_void_p()
    assert V.flex_vendor_spmm_create(C.byref(h), a.m, a.n, a.nnz, rp.data_ptr(), col.data_ptr(), val.data_ptr(), k,
                                     Bd.data_ptr(), Cd.data_ptr()) == 0
    assert V.flex_vendor_spmm_run(h, torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    V.flex_vendor_spmm_destroy(h)
    assert_matches_oracle(a, B, Cd.cpu().numpy())


@pytest.mark.parametrize("k", [4, 16, 28])
def test_in_launch_sum_with_partial_slots_that_share_a_cache_line(knobs, k):
    """k < 32: the k-wide partial sums of DIFFERENT rows share one 128-byte line, so a reducer's sc1 loads can pull a line into
    its XCD's L2 while a neighbouring row's pieces are still being written -- the case most sensitive to how the opt-in in-launch
    hand-off (split_rows = 1) is served.  100 launches under uneven load: same bits, right answer; and the default two-launch
    form gives the same bits."""
    knobs.set(split_rows=1)
    a = random_csr(5000, 5000, 6, seed=191, long_rows={r: 500 + 11 * (r % 83) for r in range(0, 5000, 2)}, empty_frac=0.05)
    Bn = random_B(a.n, k, 93)
    B = dev(Bn)
    p = Plan(a, k)
    assert p.info()["n_split_rows"] >= 2400 and p.tuning()["split_rows"] == 1
    ref = p(B).clone()
    torch.cuda.synchronize()
    assert_matches_oracle(a, Bn, ref.cpu().numpy(), nthreads=8)
    C = torch.empty_like(ref)
    filler = torch.empty(64 << 20, device="cuda")
    for it in range(100):
        C.fill_(float("nan"))
        if it % 3 == 0:
            filler.add_(1.0)
        p(B, out=C)
        if it % 20 == 19 or it < 3:
            torch.cuda.synchronize()
            assert torch.equal(C, ref), f"launch {it}: result changed"
    torch.cuda.synchronize()
    assert torch.equal(C, ref)
    knobs.set(split_rows=2)
    assert torch.equal(Plan(a, k)(B), ref)


@pytest.mark.parametrize("split_rows", [1, 2])
def test_split_row_reduction_is_stable_under_repetition(knobs, split_rows):
    """split_rows = 1: split rows are summed inside the launch by the last piece to arrive (write-through partial sums,
    agent-scope arrival counter, sc1 re-reads; opt-in since ABI 3).  A visibility bug would show as a rare wrong row:
    hammer a plan that is mostly split rows and require every launch to be bit-identical and correct.  split_rows = 2
    (the default: spmm_fixup_kernel after the launch) goes through the same hammering."""
    knobs.set(split_rows=split_rows)
    a = random_csr(6000, 6000, 8, seed=91, long_rows={r: 700 + 13 * (r % 97) for r in range(0, 6000, 3)}, empty_frac=0.05)
    k = 128
    Bn = random_B(a.n, k, 92)
    B = dev(Bn)
    p = Plan(a, k)
    info = p.info()
    assert info["n_split_rows"] >= 1900 and info["n_partials"] > 5000 and p.tuning()["split_rows"] == split_rows
    ref = p(B).clone()
    torch.cuda.synchronize()
    assert_matches_oracle(a, Bn, ref.cpu().numpy(), nthreads=8)
    C = torch.empty_like(ref)
    filler = torch.empty(64 << 20, device="cuda")  # unrelated traffic between launches (uneven load)
    for it in range(300):
        C.fill_(float("nan"))
        if it % 3 == 0:
            filler.add_(1.0)
        p(B, out=C)
        if it % 25 == 24 or it < 4:
            torch.cuda.synchronize()
            assert torch.equal(C, ref), f"launch {it}: result changed"
    torch.cuda.synchronize()
    assert torch.equal(C, ref)
    # several plans alive at once, launched back to back on one stream
    p2 = Plan(a, k, order=flex_amd.FLEX_ORDER_CLUSTER)
    r2 = p2(B).clone()
    for _ in range(50):
        p(B, out=C)
        c2 = p2(B)
    torch.cuda.synchronize()
    assert torch.equal(C, ref) and torch.equal(c2, r2)
    assert oracle.rescheck(ref.cpu().numpy(), r2.cpu().numpy(), a.rowPtr)[0] == 0


def test_bench_json_contract():
    """bench.py prints exactly one JSON line with the contract's keys, a roofline object and a CPU baseline."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "20", "--warmup", "3", "--workload", "pubmed",
                          "--k", "32", "--check"], capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in j, key
    assert j["n_gpus"] == 1 and j["steps"] == 20 and j["dtype"] == "f32" and j["vs_baseline"] is None
    assert j["check"]["mismatches"] == 0
    assert set(j["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert abs(j["roofline"]["frac"] - j["roofline"]["achieved"] / j["roofline"]["peak"]) < 1e-3
    assert j["cpu_baseline"]["kind"] == "port" and j["cpu_baseline"]["cores"] >= 1
    assert abs(j["value"] - 2 * j["config"]["nnz"] * j["config"]["k"] / (j["ms_per_step"] * 1e-3) / 1e9) < 0.01 * j["value"]
    assert j["hipsparse"]["value"] > 0
    # the memory side is read from the card's counters inside the same run (N=1): source named, bytes split, both levels of B reuse
    r = j["roofline"]
    assert r["traffic_source"].startswith("in-run") and r["traffic"] == r["traffic_read_bytes"] + r["traffic_write_bytes"] > 0, r
    assert 0 < r["l2_hit_rate"] <= 1 and r["l1_l2_bytes_measured"] > 0 and 0.8 < r["u_l1_measured"] < 2 and r["waves_per_launch"] > 0, r
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "20", "--warmup", "3", "--workload", "flickr", "--no-cpu-baseline",
                          "--no-vendor", "--no-copy-probe"], capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stdout + out.stderr
    r = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])["roofline"]
    # a graph that leaves the L2s: traffic above the algorithmic bytes, u at the L2 level defined and above the L1-level one
    assert r["traffic"] > r["algorithmic_bytes_per_launch"] and r["u_l2_measured"] > r["u_l1_measured"] > 0.8, r
    off = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "5", "--warmup", "1", "--workload", "pubmed", "--k", "32", "--no-cpu-baseline",
                          "--no-vendor", "--no-copy-probe", "--no-live-counters"], capture_output=True, text=True, timeout=600, cwd=root)
    assert off.returncode == 0, off.stdout + off.stderr
    r = json.loads([ln for ln in off.stdout.splitlines() if ln.startswith("{")][0])["roofline"]
    assert "u_l1_measured" not in r and (r["traffic_source"] is None or r["traffic_source"].startswith("profiles/")), r


@pytest.mark.parametrize("dense_block", [False, True])
def test_b_larger_than_4GiB_uses_64bit_row_addressing(knobs, dense_block):
    """n*k*4 > 2^32: records carry column ids instead of 32-bit byte offsets (OFF32 = false kernels); with a dense block of A
    at the far end of the column range the MFMA tile kernel's 64-bit row addressing is exercised too."""
    m, n, k = 1500, 2_200_000, 512
    free, _ = torch.cuda.mem_get_info()
    if free < 12 * (1 << 30):
        pytest.skip("needs ~5 GiB of HBM for B")
    rng = np.random.default_rng(123)
    deg = rng.integers(0, 40, size=m)
    deg[7] = 3000  # one split row
    if dense_block:
        deg[64:128] = 60  # rows 64..127 x the last 64 columns: four 32x32 tiles of fill ~0.9
    rp = np.zeros(m + 1, dtype=np.int64)
    np.cumsum(deg, out=rp[1:])
    col = rng.integers(0, n, size=rp[-1]).astype(np.uint32)
    col[:64] = n - 1 - np.arange(64)  # make sure the top of the address range is touched
    if dense_block:
        knobs.set(mfma=1)
        for r in range(64, 128):
            col[rp[r]:rp[r + 1]] = np.sort(rng.choice(np.arange(n - 64, n), size=60, replace=False)).astype(np.uint32)
    a = flex_amd.HostCsr(rp.astype(np.uint32), col, rng.uniform(-1, 1, rp[-1]).astype(np.float32), n=n)
    i = torch.arange(n, device="cuda", dtype=torch.float32).unsqueeze(1)
    j = torch.arange(k, device="cuda", dtype=torch.float32).unsqueeze(0)
    B = torch.sin(i * 0.37 + j * 0.11)  # cheap, non-trivial, reproducible on the host from the device copy
    assert B.numel() * 4 > (1 << 32)
    p = Plan(a, k)
    assert (p.info()["n_tiles"] >= 4) == dense_block
    C = p(B)
    torch.cuda.synchronize()
    used = np.unique(col)
    Bh_rows = B[torch.from_numpy(used.astype(np.int64)).cuda()].cpu().numpy()
    remap = np.searchsorted(used, col).astype(np.uint32)  # oracle on the compacted B (same arithmetic)
    gold = oracle.spmm(a.rowPtr, remap, a.vals, Bh_rows, nthreads=8)
    cnt, max_err, _, _ = oracle.rescheck(gold, C.cpu().numpy(), a.rowPtr)
    assert cnt == 0, (cnt, max_err)


def test_cxx_multi_gpu_driver_on_one_device():
    """libflex_mg.so (single-process row sharding + RCCL broadcast) with one device: plumbing and parity."""
    import json
    import subprocess
    exe = os.path.join(os.path.dirname(flex_amd.lib_path()), "flex")
    out = subprocess.run([exe, os.path.join(GOLDEN, "pubmed.csv"), "32", "--json", "--gpus", "1"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    rows = [json.loads(line) for line in out.stdout.splitlines() if line.startswith("{")]
    mg = [r for r in rows if r["ord"] == "MG"]
    assert len(mg) == 1 and mg[0]["errs"] == 0 and mg[0]["gpus"] == 1


def test_plan_self_check_on_every_kind_of_plan():
    """flex_plan_self_check (≙ csr2_DiagTiling's round-trip self-test, mat.cu:905-940) on full, mapped and row-shard
    plans of a graph with split rows, for one- and multi-tile widths."""
    a = random_csr(3000, 3000, 20, seed=61, long_rows={4: 2800, 77: 500})
    rank = flex_amd.order_rcm(a)
    vo, ap = flex_amd.perm_csr(a, rank)
    for k in (32, 128, 256):
        for plan in (Plan(a, k), Plan(a, k, order=flex_amd.FLEX_ORDER_CLUSTER), Plan(a, k, order=flex_amd.FLEX_ORDER_GORDER),
                     Plan(ap, k, vo_mp=vo), Plan(ap, k, rows=(0, 1200), col_map=vo), Plan(ap, k, rows=(1200, 3000), col_map=vo),
                     Plan(a, k, ldb=k + 32, ldc=k + 4)):
            plan.self_check()
            plan.destroy()
    ki = Plan(a, 128).kernel_info()  # ≙ Kernel_Info (flex.cu:4933-4941)
    assert 32 <= ki["vgprs"] <= 128 and ki["scratch_bytes"] == 0 and ki["lds_bytes"] == 8192 and ki["threads_per_block"] == 256
    assert ki["waves_per_cu"] >= 16
    empty = flex_amd.HostCsr(np.zeros(6, np.uint32), np.zeros(0, np.uint32), np.zeros(0, np.float32), n=5)
    Plan(empty, 32).self_check()  # five empty rows: five tasks, no records


def test_plan_stats_reuse_and_imbalance_report():
    """≙ alpha_stats_collect / B-Re1 / B-Re2 (mat.cu:944-1065, flex.cu:5217-5223): invariants of the report."""
    from flex_amd import FLEX_PLAN_STATS, FlexError
    n = 4096
    # (1) diagonal: no B row is ever reused
    rp = np.arange(n + 1, dtype=np.uint32)
    a = flex_amd.HostCsr(rp, np.arange(n, dtype=np.uint32), np.ones(n, np.float32), n=n)
    st = Plan(a, 32, order=FLEX_ORDER_NATURAL | FLEX_PLAN_STATS, tuning={"bundle": 2}).stats()
    assert st["cols_wave"] == st["cols_wg"] == st["cols_xcd"] == n
    assert st["reuse_wave"] == st["reuse_xcd"] == 1.0 and st["split_nnz_pct"] == 0.0
    assert st["records"] == n * 8 and abs(st["pad_pct"] - 700.0) < 1e-9  # k=32: 8 records per gather step, a row padded to one
    st = Plan(a, 32, order=FLEX_ORDER_NATURAL | FLEX_PLAN_STATS).stats()  # the rule: short rows side by side in bundles, nothing to pad
    assert st["records"] == n and st["pad_pct"] == 0.0 and st["cols_xcd"] == n
    # (2) every row reads the same 16 B rows: one fetch per chunk / workgroup / XCD slice
    deg = 16
    rp = (np.arange(n + 1) * deg).astype(np.uint32)
    col = np.tile(np.arange(deg, dtype=np.uint32) * 7, n)
    a = flex_amd.HostCsr(rp, col, np.ones(n * deg, np.float32), n=n)
    p = Plan(a, 128, order=FLEX_ORDER_NATURAL | FLEX_PLAN_STATS)
    st, info = p.stats(), p.info()
    assert st["cols_wave"] == deg * info["n_chunks"]
    assert st["cols_wg"] == deg * -(-info["n_chunks"] // 4)
    assert st["cols_xcd"] == deg * 8 and st["n_workgroups"] % 8 == 0
    assert st["records"] == info["nnz"] and st["pad_pct"] == 0.0
    assert st["gather_bytes"] == 4.0 * (n + 1) + 8.0 * n * deg + 4.0 * n * deg * 128 + 4.0 * n * 128
    assert st["chunk_imb_pct"] >= 0.0 and st["xcd_imb_pct"] >= 0.0
    # (3) ragged graph: ordering of the counts, split share, and the report leaves the result alone
    a = random_csr(3000, 3000, 12, seed=5, long_rows={7: 2500, 11: 900})
    B = random_B(3000, 64, 1)
    p = Plan(a, 64, order=flex_amd.FLEX_ORDER_CLUSTER | FLEX_PLAN_STATS)
    st = p.stats()
    distinct = len(np.unique(a.col))
    assert st["cols_wave"] >= st["cols_wg"] >= st["cols_xcd"] >= distinct
    assert st["records"] >= a.nnz and 0.0 < st["split_nnz_pct"] < 100.0
    assert st["split_nnz_pct"] >= 100.0 * (2500 + 900) / a.nnz - 1e-6  # the two planted long rows are cut
    assert_matches_oracle(a, B, run_plan(p, B))
    # (4) not collected unless asked for; unknown flag bits are refused
    with pytest.raises(FlexError):
        Plan(a, 64).stats()
    with pytest.raises(FlexError):
        Plan(a, 64, order=0x400)


def test_hbm_probe_reports_a_plausible_bandwidth():
    """flex_hbm_probe: the measured denominator beside the 8 TB/s spec figure."""
    pr = flex_amd.hbm_probe(0, mib=1024, reps=5)
    assert 1000.0 < pr["read_GBps"] < 8000.0 and 1000.0 < pr["copy_GBps"] < 8000.0
    with pytest.raises(flex_amd.FlexError):
        flex_amd.hbm_probe(0, mib=0)


@pytest.mark.parametrize("lanes,k", [(64, 256), (32, 256), (16, 128), (8, 128), (8, 64)])
def test_every_column_tile_width_gives_the_same_answer(knobs, lanes, k):
    """The planner picks the lanes-per-record G (column tile = 4G) from k and the average degree; every
    width it can pick -- including G=64, which the default rule never selects -- must pass resCheck,
    with split rows reduced in-launch across several k-tiles (and by the default two-launch form: same bits)."""
    a = random_csr(2500, 2500, 30, seed=21, long_rows={3: 2400, 9: 700})
    B = random_B(2500, k, 4)
    knobs.set(lanes_per_nz=lanes, split_rows=1)
    p = Plan(a, k, order=flex_amd.FLEX_ORDER_CLUSTER)
    assert p.info()["lanes_per_nz"] == lanes
    C1 = run_plan(p, B)
    assert_matches_oracle(a, B, C1)
    assert np.array_equal(C1, run_plan(p, B))  # deterministic, counters re-armed
    knobs.clear("split_rows")
    assert np.array_equal(C1, run_plan(Plan(a, k, order=flex_amd.FLEX_ORDER_CLUSTER), B))  # pieces are added in piece order either way
    knobs.clear("lanes_per_nz")
    p2 = Plan(a, k, order=flex_amd.FLEX_ORDER_CLUSTER)
    assert p2.info()["lanes_per_nz"] == 16  # average degree >= 24: narrow tiles by default
    assert_matches_oracle(a, B, run_plan(p2, B))


@pytest.mark.parametrize("dim,c", [(32, 7), (128, 41), (16, 64), (128, 100), (64, 130), (20, 33), (12, 5), (130, 40), (256, 16), (200, 70), (256, 100), (260, 8)])
def test_axw_both_orders_match_a_float64_reference(dim, c):
    """libflex_axw.so (≙ run1 / run2 of cusp.cu): A(XW) and (AX)W agree with each other (the reference's own
    check, DataLoader.cu:859-869) and with A @ X @ W in float64; padded columns are zero; AUTO takes the
    order with the narrower SpMM.  dim <= 128 and dim % 4 == 0 run the hand-written MFMA GEMM (one to several
    column passes, odd and even k-pair counts), the rest rocBLAS."""
    import scipy.sparse as sp
    import torch
    from flex_amd.axw import FLEX_AXW_A_XW, FLEX_AXW_AX_W, Axw
    a = random_csr(3000, 3000, 9, seed=31, long_rows={5: 1500})
    rng = np.random.default_rng(3)
    X = rng.uniform(-1, 1, size=(a.n, dim)).astype(np.float32)
    W = rng.uniform(0, 1, size=(dim, c)).astype(np.float32)  # DataLoader.cu:172: W in [0,1)
    A = sp.csr_matrix((a.vals.astype(np.float64), a.col.astype(np.int64), a.rowPtr.astype(np.int64)), shape=(a.m, a.n))
    gold = A @ (X.astype(np.float64) @ W.astype(np.float64))
    scale = np.abs(A) @ (np.abs(X).astype(np.float64) @ np.abs(W).astype(np.float64))  # magnitude of the terms summed
    h = Axw(a, dim, c)
    assert h.ld == -(-c // 32) * 32  # whole 128-byte lines per row
    Xd, Wd = torch.from_numpy(X).cuda(), torch.from_numpy(W).cuda()
    outs = {}
    for order in (FLEX_AXW_A_XW, FLEX_AXW_AX_W):
        out, (gemm_ms, spmm_ms) = h.run(Xd, Wd, order, timed=True)
        o = out.cpu().numpy()
        assert gemm_ms > 0 and spmm_ms > 0
        assert np.all(o[:, c:] == 0)
        assert np.all(np.abs(o[:, :c] - gold) <= 1e-5 * scale + 1e-6), np.abs(o[:, :c] - gold).max()
        outs[order] = o
    assert np.allclose(outs[FLEX_AXW_A_XW], outs[FLEX_AXW_AX_W], rtol=1e-3, atol=1e-3)
    auto = h.run(Xd, Wd).cpu().numpy()
    assert np.array_equal(auto, outs[FLEX_AXW_A_XW if h.ld <= dim else FLEX_AXW_AX_W])
    h.destroy()


@pytest.mark.parametrize("k,ldb,ldc", [(100, 128, 128), (100, 100, 128), (12, 20, 16), (7, 9, 11), (128, 160, 128), (260, 288, 264)])
def test_strided_dense_operands(k, ldb, ldc):
    """flex_plan_create_ld: rows of B / C start every ldb / ldc floats; columns >= k of B are never read into
    the result and the tail of every C row is left untouched (vector and generic kernels, split rows included)."""
    import torch
    a = random_csr(2000, 2000, 14, seed=41, long_rows={2: 1900, 17: 600})
    rng = np.random.default_rng(8)
    Bs = rng.uniform(-1, 1, size=(a.n, ldb)).astype(np.float32)
    Bs[:, k:] = np.nan  # poison: anything read from the padding would surface
    p = Plan(a, k, order=flex_amd.FLEX_ORDER_CLUSTER, ldb=ldb, ldc=ldc)
    Bd = torch.from_numpy(Bs).cuda()
    Cd = torch.full((a.m, ldc), -7.0, dtype=torch.float32, device="cuda")
    for _ in range(2):
        p.spmm(Bd.data_ptr(), Cd.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    Cs = Cd.cpu().numpy()
    assert np.all(Cs[:, k:] == -7.0)
    assert_matches_oracle(a, np.ascontiguousarray(Bs[:, :k]), np.ascontiguousarray(Cs[:, :k]))
    with pytest.raises(flex_amd.FlexError):
        Plan(a, k, ldb=k - 1)


def test_seeded_fuzz_over_shapes_degrees_widths_schedules():
    """250 seeded random cases through the planner's every branch at small sizes: rectangular and square
    shapes, empty rows / empty matrices, degrees on both sides of the column-tile thresholds, rows long
    enough to split (in-launch reduction across k-tiles), every k class (vector, generic, multi-tile),
    every schedule, strided operands -- each checked with the reference's resCheck against the oracle."""
    import torch
    rng = np.random.default_rng(int(os.environ.get("FLEX_FUZZ_SEED", "20251004")))  # other seeds: soak runs
    orders = [FLEX_ORDER_NATURAL, FLEX_ORDER_RCM, flex_amd.FLEX_ORDER_CLUSTER]
    for case in range(250):
        m = int(rng.choice([1, 2, 7, 63, 64, 65, 300, 1500, 4000, 20000]))
        square = bool(rng.integers(0, 2)) or m < 3
        n = m if square else int(rng.choice([1, 5, 97, 1000, 5000]))
        avg = float(rng.choice([0.0, 0.5, 3, 12, 30, 140]))
        long_rows = {}
        if m >= 300 and rng.integers(0, 2):
            long_rows = {int(rng.integers(0, m)): int(min(n, rng.integers(200, 3000))) for _ in range(int(rng.integers(1, 4)))}
        a = random_csr(m, n, min(avg, n), seed=1000 + case, long_rows=long_rows, empty_frac=float(rng.choice([0.0, 0.1, 0.6])),
                       sorted_cols=bool(rng.integers(0, 2)))
        k = int(rng.choice([1, 4, 5, 8, 32, 36, 64, 100, 128, 132, 256, 300]))
        order = int(rng.choice(orders)) if square else FLEX_ORDER_NATURAL
        strided = bool(rng.integers(0, 3) == 0)
        ldb = k + int(rng.choice([0, 4, 28])) if strided else k
        ldc = k + int(rng.choice([0, 4, 28])) if strided else k
        B = rng.uniform(-1, 1, size=(n, ldb)).astype(np.float32)
        how = {"split_rows": 1 + case % 2}  # both forms of the split-row sum
        p = Plan(a, k, order=order | flex_amd.FLEX_PLAN_STATS, ldb=ldb, ldc=ldc, tuning=how) if strided else Plan(a, k, order=order | flex_amd.FLEX_PLAN_STATS, tuning=how)
        Cd = torch.full((m, ldc), 3.5, dtype=torch.float32, device="cuda")
        p.spmm(torch.from_numpy(B).cuda().data_ptr(), Cd.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        C = Cd.cpu().numpy()
        tag = f"case {case}: m={m} n={n} avg={avg} k={k} order={order} ld=({ldb},{ldc}) long={long_rows}"
        assert np.all(C[:, k:] == 3.5), tag
        gold = oracle.spmm(a.rowPtr, a.col, a.vals, np.ascontiguousarray(B[:, :k]), nthreads=4)
        cnt, max_err, me_nnz, _ = oracle.rescheck(gold, np.ascontiguousarray(C[:, :k]), a.rowPtr)
        assert cnt == 0, f"{tag}: {cnt} mismatches, max err {max_err:g} on a row of {me_nnz} nnz"
        st, info = p.stats(), p.info()
        assert st["records"] + info["tile_nnz"] >= a.nnz and info["n_slots"] >= info["n_chunks"] and info["nnz"] == a.nnz, tag
        p.self_check()  # the device image of the plan is a partition of the work (≙ the tiler round-trip, mat.cu:905-940)
        p.destroy()


@pytest.mark.parametrize("k", [32, 64, 128])
def test_two_launch_form_of_split_rows(knobs, k):
    """split_rows = 2 (the rule above 4e8 multiply-adds per launch): the vector kernel leaves the partial sums of split rows to
    spmm_fixup_kernel (the form the generic kernel always uses); split_rows = 1 (the rule for small launches, where the second
    launch's kernel boundary is the larger cost) sums them inside the launch.  Same bits."""
    a = random_csr(2500, 2500, 10, seed=51, long_rows={1: 2400, 8: 1100, 900: 300})
    B = random_B(2500, k, 6)
    knobs.set(split_rows=2)
    p = Plan(a, k, order=flex_amd.FLEX_ORDER_CLUSTER)
    assert p.info()["n_split_rows"] >= 3 and p.tuning()["split_rows"] == 2
    C1 = run_plan(p, B)
    assert_matches_oracle(a, B, C1)
    knobs.clear("split_rows")
    p_in = Plan(a, k, order=flex_amd.FLEX_ORDER_CLUSTER)
    assert p_in.tuning()["split_rows"] == 1  # by the size rule
    C2 = run_plan(p_in, B)
    assert np.array_equal(C1, C2)  # both forms add the pieces in piece order: bit-identical


def test_two_plans_on_two_streams_concurrently():
    """The contract: one plan must not run on two streams at once, but DIFFERENT plans may; each
    owns its partial-sum workspace and arrival counters.  Interleaved launches of two plans with split rows on two
    streams give the same bits as running each alone."""
    import torch
    a1 = random_csr(4000, 4000, 12, seed=71, long_rows={3: 3500, 50: 900})
    a2 = random_csr(3000, 3000, 40, seed=72, long_rows={7: 2900})
    B1, B2 = random_B(4000, 128, 1), random_B(3000, 64, 2)
    p1, p2 = Plan(a1, 128, order=flex_amd.FLEX_ORDER_CLUSTER), Plan(a2, 64, order=flex_amd.FLEX_ORDER_CLUSTER)
    ref1, ref2 = run_plan(p1, B1), run_plan(p2, B2)
    assert_matches_oracle(a1, B1, ref1)
    assert_matches_oracle(a2, B2, ref2)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    d1, d2 = torch.from_numpy(B1).cuda(), torch.from_numpy(B2).cuda()
    c1 = torch.empty((4000, 128), device="cuda")
    c2 = torch.empty((3000, 64), device="cuda")
    torch.cuda.synchronize()
    for _ in range(50):
        p1.spmm(d1.data_ptr(), c1.data_ptr(), s1.cuda_stream)
        p2.spmm(d2.data_ptr(), c2.data_ptr(), s2.cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(c1.cpu().numpy(), ref1) and np.array_equal(c2.cpu().numpy(), ref2)


@pytest.mark.parametrize("k", [4, 8, 12, 16])
@pytest.mark.parametrize("split_rows", [1, 2])
def test_narrow_tile_for_k_up_to_16(knobs, k, split_rows):
    """k <= 16 (≙ the reference's narrow kernel, flex.cu:81-118): four lanes per record, sixteen records per step.  By rule from
    an average degree of 8; forced on a low-degree graph; against the oracle, against the 8-lane tile of the same plan shape, with
    rows cut into pieces under both forms of the split-row sum."""
    knobs.set(split_rows=split_rows)
    a = random_csr(6000, 6000, 14, seed=101, long_rows={7: 5000, 300: 900, 4000: 70})
    B = random_B(a.n, k, 9)
    for order in (FLEX_ORDER_NATURAL, flex_amd.FLEX_ORDER_CLUSTER):
        p = Plan(a, k, order=order)
        assert p.info()["lanes_per_nz"] == 4 and p.info()["n_split_rows"] >= 2
        p.self_check()
        C4 = run_plan(p, B)
        assert_matches_oracle(a, B, C4)
        assert np.array_equal(C4, run_plan(p, B))
        knobs.set(lanes_per_nz=8)
        p8 = Plan(a, k, order=order)
        assert p8.info()["lanes_per_nz"] == 8
        assert oracle.rescheck(run_plan(p8, B), C4, a.rowPtr)[0] == 0
        knobs.clear("lanes_per_nz")
    # average degree 3: without bundles the rule keeps the 8-lane tile (rows are padded to whole steps) and the knob forces the narrow
    # one; with bundles (the rule for rows this short) nothing is padded and the narrow tile is the rule
    low = random_csr(5000, 5000, 3, seed=102)
    Bl = random_B(low.n, k, 10)
    knobs.set(bundle=2)
    assert Plan(low, k).info()["lanes_per_nz"] == 8
    knobs.set(lanes_per_nz=4)
    pl = Plan(low, k)
    assert pl.info()["lanes_per_nz"] == 4 and pl.info()["n_bundles"] == 0
    pl.self_check()
    assert_matches_oracle(low, Bl, run_plan(pl, Bl))
    knobs.clear("lanes_per_nz")
    knobs.clear("bundle")
    pb = Plan(low, k)
    assert pb.info()["lanes_per_nz"] == 4 and pb.info()["n_bundles"] > 0
    assert_matches_oracle(low, Bl, run_plan(pb, Bl))
    knobs.set(lanes_per_nz=4)
    # wider k ignores the request
    assert Plan(low, 32).info()["lanes_per_nz"] == 8


def test_far_records_first_is_a_reordering_inside_tasks_only(knobs):
    """tuning.far_first: inside every task the records whose column lies far from the row in the schedule come first (measured: no
    effect on launch time, profiles/r04_far_first_probe.txt; the knob stays for experiments).  Same plan shape, same result within
    the tolerance, for a whole plan, a mapped plan and pieces of long rows."""
    g = flex_amd.synth_graph(n=20000, nnz=20000 + 2 * 400000, community=300, p_in=0.6, p_near=0.25, seed=31)
    B = random_B(g.n, 128, 4)
    base = Plan(g, 128, order=flex_amd.FLEX_ORDER_CLUSTER)
    C0 = run_plan(base, B)
    gold, _ = assert_matches_oracle(g, B, C0)
    knobs.set(far_first=600)
    p = Plan(g, 128, order=flex_amd.FLEX_ORDER_CLUSTER)
    assert p.tuning()["far_first"] == 600 and p.info()["n_records"] == base.info()["n_records"] and p.info()["n_chunks"] == base.info()["n_chunks"]
    p.self_check()
    C1 = run_plan(p, B)
    assert oracle.rescheck(gold, C1, g.rowPtr)[0] == 0 and not np.array_equal(C0, C1)  # the sum order of a row did change
    vo, gp = flex_amd.perm_csr(g, flex_amd.order_cluster(g))
    knobs.set(long_row=64, piece_records=64)
    assert oracle.rescheck(gold, run_plan(Plan(gp, 128, vo_mp=vo), B), g.rowPtr)[0] == 0


def test_one_plan_on_two_streams_at_once_is_refused_not_corrupted():
    """include/flex_spmm.h: a plan with split rows owns their workspace, so a launch on a DIFFERENT stream while its latest launch
    is still in flight returns FLEX_ERR_INVALID (and enqueues nothing) instead of silently corrupting those rows; the same stream,
    or another stream once the first has finished, is fine -- and a plan WITHOUT split rows may overlap freely."""
    import torch
    a = random_csr(4000, 4000, 12, seed=71, long_rows={3: 3500, 50: 900})
    B = random_B(4000, 128, 1)
    p = Plan(a, 128, order=flex_amd.FLEX_ORDER_CLUSTER)
    assert p.info()["n_partials"] > 0
    ref = run_plan(p, B)
    assert_matches_oracle(a, B, ref)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    d = torch.from_numpy(B).cuda()
    c1, c2 = torch.empty((4000, 128), device="cuda"), torch.empty((4000, 128), device="cuda")
    torch.cuda.synchronize()
    with torch.cuda.stream(s1):
        torch.cuda._sleep(400_000_000)  # ~0.2 s of device time in front of the launch: it IS in flight when the next call comes
    p.spmm(d.data_ptr(), c1.data_ptr(), s1.cuda_stream)
    p.spmm(d.data_ptr(), c1.data_ptr(), s1.cuda_stream)  # same stream: ordered by the stream, accepted
    with pytest.raises(flex_amd.FlexError, match="invalid argument"):
        p.spmm(d.data_ptr(), c2.data_ptr(), s2.cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(c1.cpu().numpy(), ref)
    p.spmm(d.data_ptr(), c2.data_ptr(), s2.cuda_stream)  # the first stream has drained: another stream is fine now
    torch.cuda.synchronize()
    assert np.array_equal(c2.cpu().numpy(), ref)
    # no split rows -> no workspace -> nothing to guard
    a0 = random_csr(3000, 3000, 8, seed=73)
    p0 = Plan(a0, 64, order=flex_amd.FLEX_ORDER_CLUSTER, tuning={"long_row": 1 << 20})
    assert p0.info()["n_partials"] == 0
    d0 = torch.from_numpy(random_B(3000, 64, 2)).cuda()
    e1, e2 = torch.empty((3000, 64), device="cuda"), torch.empty((3000, 64), device="cuda")
    with torch.cuda.stream(s1):
        torch.cuda._sleep(100_000_000)
    p0.spmm(d0.data_ptr(), e1.data_ptr(), s1.cuda_stream)
    p0.spmm(d0.data_ptr(), e2.data_ptr(), s2.cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(e1.cpu().numpy(), e2.cpu().numpy())


def test_autotuned_plan_is_correct_and_no_slower_choice_is_kept():
    """FLEX_PLAN_AUTOTUNE plans the neighbouring column-tile widths as well and keeps the fastest: whatever it keeps
    must pass resCheck and the self-check, statistics must describe the kept plan, and odd k (generic kernel) is a no-op."""
    from flex_amd import FLEX_PLAN_AUTOTUNE, FLEX_PLAN_STATS
    for deg, k in ((8, 128), (40, 128), (40, 32), (8, 100), (30, 7)):
        a = random_csr(6000, 6000, deg, seed=81, long_rows={9: 5000})
        B = random_B(6000, k, 5)
        p = Plan(a, k, order=flex_amd.FLEX_ORDER_CLUSTER | FLEX_PLAN_AUTOTUNE | FLEX_PLAN_STATS)
        info, st = p.info(), p.stats()
        assert info["lanes_per_nz"] in (8, 16, 32) and st["records"] >= a.nnz
        p.self_check()
        assert_matches_oracle(a, B, run_plan(p, B))
    # round 4: on the tiles that have row bundles the OTHER bundle setting is planned and timed as well (the rule is one size
    # threshold); whichever is kept is a valid plan that says what it is, and a caller's own choice is not overridden
    a = flex_amd.synth_graph(n=60000, nnz=60000 + 2 * 240000, community=128, p_in=0.6, p_near=0.2, seed=82)
    B = random_B(a.n, 32, 6)
    p = Plan(a, 32, order=flex_amd.FLEX_ORDER_CLUSTER | FLEX_PLAN_AUTOTUNE)
    p.self_check()
    assert p.tuning()["bundle"] == (1 if p.info()["n_bundles"] else 2)
    assert_matches_oracle(a, B, run_plan(p, B))
    forced = Plan(a, 32, order=flex_amd.FLEX_ORDER_CLUSTER | FLEX_PLAN_AUTOTUNE, tuning={"bundle": 2})
    assert forced.info()["n_bundles"] == 0


def test_general_entry_point_covers_combinations():
    """flex_plan_create_ex: a reordered (mapped) matrix and a row shard of it, both over padded storage -- the
    combinations the named entry points do not offer."""
    import torch
    a = random_csr(2400, 2400, 15, seed=91, long_rows={6: 2000})
    rank = flex_amd.order_rcm(a)
    vo, ap = flex_amd.perm_csr(a, rank)
    k, ldb, ldc = 100, 128, 104
    rng = np.random.default_rng(3)
    Bs = rng.uniform(-1, 1, size=(a.n, ldb)).astype(np.float32)
    Bd = torch.from_numpy(Bs).cuda()
    gold = oracle.spmm(a.rowPtr, a.col, a.vals, np.ascontiguousarray(Bs[:, :k]), nthreads=4)
    # mapped + strided: B and C in ORIGINAL order although the plan is built from the permuted CSR
    p = Plan(ap, k, vo_mp=vo, ldb=ldb, ldc=ldc)
    p.self_check()
    Cd = torch.full((a.m, ldc), 9.0, dtype=torch.float32, device="cuda")
    p.spmm(Bd.data_ptr(), Cd.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    C = Cd.cpu().numpy()
    assert np.all(C[:, k:] == 9.0) and oracle.rescheck(gold, np.ascontiguousarray(C[:, :k]), a.rowPtr)[0] == 0
    # two row shards of the permuted matrix, strided: slice-local C rows, original B
    got = np.zeros_like(gold)
    for r0, r1 in ((0, 1000), (1000, 2400)):
        ps = Plan(ap, k, rows=(r0, r1), col_map=vo, ldb=ldb, ldc=ldc)
        ps.self_check()
        Cs = torch.full((r1 - r0, ldc), 9.0, dtype=torch.float32, device="cuda")
        ps.spmm(Bd.data_ptr(), Cs.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        Csh = Cs.cpu().numpy()
        assert np.all(Csh[:, k:] == 9.0)
        got[vo[r0:r1]] = Csh[:, :k]
    assert oracle.rescheck(gold, got, a.rowPtr)[0] == 0
    # argument checks of the descriptor form
    import ctypes as C
    from flex_amd import binding
    h = C.c_void_p()
    v = a.view()
    bad = binding._PlanDesc(8, C.pointer(v), k, 0, 0, 0, 0, 0, 0, None, None)  # wrong struct_size
    assert binding.lib().flex_plan_create_ex(C.byref(h), C.byref(bad)) == -1
    shard_ordered = binding._PlanDesc(C.sizeof(binding._PlanDesc), C.pointer(v), k, 0, 0, 0, flex_amd.FLEX_ORDER_RCM | flex_amd.FLEX_PLAN_ROW_RANGE, 0, 100, None, None)
    assert binding.lib().flex_plan_create_ex(C.byref(h), C.byref(shard_ordered)) == -1  # reorder first, then shard
    # an EMPTY shard [0,0) through the general entry point is a plan over zero rows (it used to mean "all rows",
    # and flex_spmm then wrote m rows into a zero-row C): nothing is launched, nothing is written
    pe = Plan(ap, k, rows=(0, 0), col_map=vo, ldb=ldb, ldc=ldc)
    assert pe.info()["m"] == 0 and pe.info()["nnz"] == 0
    guard = torch.full((4, ldc), 5.0, dtype=torch.float32, device="cuda")
    pe.spmm(Bd.data_ptr(), guard.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.all(guard == 5.0)
    # zero-initialised descriptor (no FLEX_PLAN_ROW_RANGE) = all rows
    allrows = binding._PlanDesc(C.sizeof(binding._PlanDesc), C.pointer(v), k, 0, 0, 0, 0, 0, 0, None, None)
    assert binding.lib().flex_plan_create_ex(C.byref(h), C.byref(allrows)) == 0
    info = binding._PlanInfo()
    binding.lib().flex_plan_get_info(h, C.byref(info))
    assert info.m == a.m
    binding.lib().flex_plan_destroy(h)


@pytest.mark.parametrize("k,lanes", [(32, 0), (64, 0), (128, 0), (128, 8), (128, 32), (256, 0), (100, 0), (36, 0), (7, 0)])
@pytest.mark.parametrize("order", [FLEX_ORDER_NATURAL, flex_amd.FLEX_ORDER_CLUSTER])
def test_two_d_column_panel_schedule(knobs, k, lanes, order):
    """FLEX_2D=1: every XCD slice of the rows is walked column panel by column panel; a row with records in several
    panels is summed from several pieces (partial slots + arrival counters, many pieces per chunk).  Tiny panels force
    many phases on a small graph: resCheck against the oracle, the plan self-check, bit-identical repeat launches,
    and the same bits from the two-launch form (spmm_fixup_kernel adds the pieces in the same order)."""
    a = flex_amd.synth_graph(n=6000, nnz=6000 + 2 * 90000, community=200, p_in=0.55, p_near=0.3, seed=5)
    B = random_B(a.n, k, 9)
    knobs.set(two_d=1, panel_kb=32, seg_min=2, split_rows=1)
    if lanes:
        knobs.set(lanes_per_nz=lanes)
    p = Plan(a, k, order=order)
    info = p.info()
    assert info["two_d"] == 1 and info["panel_rows"] >= 32 and info["n_split_rows"] > a.m // 2 and info["n_partials"] > a.m
    p.self_check()
    C1 = run_plan(p, B)
    assert_matches_oracle(a, B, C1)
    assert np.array_equal(C1, run_plan(p, B))  # counters re-armed, same piece order
    knobs.set(split_rows=2)
    p2 = Plan(a, k, order=order)
    assert np.array_equal(C1, run_plan(p2, B))
    knobs.clear("split_rows")
    # against the 1-D schedule of the same matrix: same tolerance, not the same bits
    knobs.clear("two_d")
    p1 = Plan(a, k, order=order)
    assert p1.info()["two_d"] == 0
    assert oracle.rescheck(run_plan(p1, B), C1, a.rowPtr)[0] == 0


def test_two_d_with_hubs_empty_rows_shards_and_strides(knobs):
    """2-D plans of awkward inputs: hub rows (runs longer than a budget inside a panel), empty rows, a rectangular
    matrix, row shards with a column map, padded storage."""
    knobs.set(two_d=1)
    knobs.set(panel_kb=64)
    a = random_csr(5000, 5000, 25, seed=77, long_rows={3: 4500, 2500: 1800, 4999: 700}, empty_frac=0.1)
    for k, ldb, ldc in ((128, None, None), (64, 96, 68)):
        B = random_B(a.n, ldb or k, 3)
        p = Plan(a, k, ldb=ldb, ldc=ldc) if ldb else Plan(a, k)
        assert p.info()["two_d"] == 1
        p.self_check()
        if ldb:
            Cd = torch.full((a.m, ldc), 2.5, device="cuda")
            p.spmm(dev(B).data_ptr(), Cd.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            C = Cd.cpu().numpy()
            assert np.all(C[:, k:] == 2.5)
            assert_matches_oracle(a, np.ascontiguousarray(B[:, :k]), np.ascontiguousarray(C[:, :k]))
        else:
            assert_matches_oracle(a, B, run_plan(p, B))
    rect = random_csr(3000, 9000, 40, seed=78, long_rows={5: 6000})
    Br = random_B(rect.n, 128, 4)
    pr = Plan(rect, 128)
    pr.self_check()
    assert_matches_oracle(rect, Br, run_plan(pr, Br))
    # row shards of a reordered matrix (what one GPU of a row-sharded run plans)
    g = flex_amd.synth_graph(n=8000, nnz=8000 + 2 * 150000, community=256, p_in=0.6, p_near=0.25, seed=6)
    Bg = random_B(g.n, 128, 5)
    gold = oracle.spmm(g.rowPtr, g.col, g.vals, Bg, nthreads=8)
    got = np.zeros_like(gold)
    for r in range(3):
        sh = flex_amd.make_shard(g, 128, r, 3, order="cluster")
        ps = sh.plan(128, 0)
        assert ps.info()["two_d"] == 1
        ps.self_check()
        got[sh.original_rows()] = run_plan(ps, Bg)
    assert oracle.rescheck(gold, got, g.rowPtr)[0] == 0


def test_two_d_schedule_on_top_of_the_dense_tile_route(knobs):
    """Both plan features at once, executed: dense diagonal blocks go to the MFMA kernel, the rest is walked panel by panel."""
    a = block_dense_graph(6400, 64, 0.85, 12, seed=11)
    B = random_B(a.n, 128, 2)
    knobs.set(two_d=1)
    knobs.set(panel_kb=64)
    knobs.set(seg_min=2)
    p = Plan(a, 128)
    info = p.info()
    assert info["two_d"] == 1 and info["n_tiles"] > 300 and info["n_partials"] > 0
    p.self_check()
    C1 = run_plan(p, B)
    assert_matches_oracle(a, B, C1)
    assert np.array_equal(C1, run_plan(p, B))


def test_two_d_reduction_is_stable_under_repetition(knobs):
    """The in-launch combination now runs for pieces scattered over many chunks (several arrivals per chunk, up to S
    rows completed per round): 200 launches under uneven load must give the same bits and the right answer."""
    knobs.set(two_d=1, split_rows=1)
    knobs.set(panel_kb=64)
    knobs.set(seg_min=2)
    a = flex_amd.synth_graph(n=40000, nnz=40000 + 2 * 1500000, community=512, p_in=0.5, p_near=0.3, seed=8)
    k = 128
    Bn = random_B(a.n, k, 92)
    B = dev(Bn)
    p = Plan(a, k, order=flex_amd.FLEX_ORDER_CLUSTER)
    info = p.info()
    assert info["two_d"] == 1 and info["n_split_rows"] > 30000
    ref = p(B).clone()
    torch.cuda.synchronize()
    assert_matches_oracle(a, Bn, ref.cpu().numpy(), nthreads=8)
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


def block_dense_graph(n, block, fill, noise_deg, seed):
    """Block-diagonal-plus-noise matrix: dense diagonal blocks of `block` rows at `fill`, plus `noise_deg` random entries
    per row -- the kind of input north_star's MFMA clause is about (what a good ordering makes of a clustered graph)."""
    rng = np.random.default_rng(seed)
    rows, cols = [], []
    for b0 in range(0, n, block):
        b1 = min(n, b0 + block)
        mask = rng.random((b1 - b0, b1 - b0)) < fill
        r, c = np.nonzero(mask)
        rows.append(r + b0)
        cols.append(c + b0)
    rows.append(np.repeat(np.arange(n), noise_deg))
    cols.append(rng.integers(0, n, size=n * noise_deg))
    r, c = np.concatenate(rows), np.concatenate(cols)
    key = np.unique(r.astype(np.int64) * n + c)
    r, c = key // n, key % n
    rp = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(r, minlength=n), out=rp[1:])
    vals = rng.uniform(-1, 1, size=len(c)).astype(np.float32)
    return flex_amd.HostCsr(rp.astype(np.uint32), c.astype(np.uint32), vals, n=n)


@pytest.mark.parametrize("k", [128, 32, 100, 256, 7])
def test_mfma_dense_tile_route_matches_oracle(knobs, k):
    """north_star: "MFMA only where ... reordering yields dense block-sparse tiles".  On a block-dense input the planner's
    detector routes the dense 32x32 tiles to the v_mfma_f32_32x32x2_f32 kernel and the rest to the vector kernel; the sum
    must pass resCheck against the oracle, be reproducible, and agree with the vector-only plan of the same matrix."""
    a = block_dense_graph(12000, 64, 0.85, 6, seed=3)  # well above the default routing threshold (fill 0.6)
    B = random_B(a.n, k, 17)
    p = Plan(a, k, order=FLEX_ORDER_NATURAL | flex_amd.FLEX_PLAN_STATS)
    info, st = p.info(), p.stats()
    assert info["n_tiles"] > 300 and info["tile_nnz"] > 0.7 * a.nnz, info      # the route is taken
    assert st["tile_nnz_pct_50"] > 70.0 and st["mfma_tiles"] == info["n_tiles"] and st["mfma_nnz_pct"] > 70.0
    p.self_check()
    C1 = run_plan(p, B)
    assert_matches_oracle(a, B, C1)
    assert np.array_equal(C1, run_plan(p, B))
    knobs.set(mfma=2)
    pv = Plan(a, k, order=FLEX_ORDER_NATURAL | flex_amd.FLEX_PLAN_STATS)
    assert pv.info()["n_tiles"] == 0 and pv.stats()["tile_nnz_pct_50"] > 70.0   # the detector still reports, nothing is routed
    assert oracle.rescheck(run_plan(pv, B), C1, a.rowPtr)[0] == 0


def test_mfma_route_after_reordering_shards_strides_and_duplicates(knobs):
    """The detector works in SCHEDULE coordinates: a shuffled block-dense graph has no dense tile in natural order and
    plenty after the community ordering; mapped plans, row shards, padded storage and duplicate entries go through."""
    a0 = block_dense_graph(8000, 32, 0.8, 4, seed=5)
    perm = np.random.default_rng(1).permutation(a0.m)  # relabel vertices at random
    inv = np.argsort(perm)
    rows = np.repeat(np.arange(a0.m), np.diff(a0.rowPtr.astype(np.int64)))
    r2, c2 = perm[rows], perm[a0.col]
    o = np.lexsort((c2, r2))
    rp = np.zeros(a0.m + 1, dtype=np.int64)
    np.cumsum(np.bincount(r2, minlength=a0.m), out=rp[1:])
    a = flex_amd.HostCsr(rp.astype(np.uint32), c2[o].astype(np.uint32), a0.vals[o], n=a0.n)
    del inv
    k = 128
    B = random_B(a.n, k, 4)
    knobs.set(mfma=1)
    knobs.set(mfma_fill_pct=25)  # communities do not start on tile boundaries: a block straddles tiles
    nat = Plan(a, k, order=FLEX_ORDER_NATURAL)
    clu = Plan(a, k, order=flex_amd.FLEX_ORDER_CLUSTER)
    assert nat.info()["tile_nnz"] < 0.05 * a.nnz < 0.3 * a.nnz < clu.info()["tile_nnz"]
    gold, _ = assert_matches_oracle(a, B, run_plan(clu, B))
    assert_matches_oracle(a, B, run_plan(nat, B))
    # the reference's flow: permuted loader + vo_mp, then 3 row shards of it over padded storage
    vo, ap = flex_amd.perm_csr(a, flex_amd.order_cluster(a))
    pm = Plan(ap, k, vo_mp=vo)
    assert pm.info()["n_tiles"] > 0
    pm.self_check()
    assert oracle.rescheck(gold, run_plan(pm, B), a.rowPtr)[0] == 0
    ldb, ldc = k + 32, k + 4
    Bs = np.full((a.n, ldb), np.nan, dtype=np.float32)
    Bs[:, :k] = B
    Bd = dev(Bs)
    got = np.zeros_like(gold)
    bounds = flex_amd.shard_rows(ap, k, 3)
    for i in range(3):
        ps = Plan(ap, k, rows=(bounds[i], bounds[i + 1]), col_map=vo, ldb=ldb, ldc=ldc)
        assert ps.info()["n_tiles"] > 0
        ps.self_check()
        Cs = torch.full((bounds[i + 1] - bounds[i], ldc), 4.0, device="cuda")
        ps.spmm(Bd.data_ptr(), Cs.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        Ch = Cs.cpu().numpy()
        assert np.all(Ch[:, k:] == 4.0)
        got[vo[bounds[i]:bounds[i + 1]]] = Ch[:, :k]
    assert oracle.rescheck(gold, got, a.rowPtr)[0] == 0
    # duplicates: the same (row, col) twice -- one copy rides in the tile, the other stays with the vector kernel
    dup = flex_amd.HostCsr(np.arange(0, 64 * 65, 64, dtype=np.uint32)[:65], np.tile(np.repeat(np.arange(32, dtype=np.uint32), 2), 64),
                           np.random.default_rng(2).uniform(-1, 1, 64 * 64).astype(np.float32), n=64)
    Bdup = random_B(64, 32, 6)
    pd = Plan(dup, 32)
    assert pd.info()["n_tiles"] == 2 and pd.info()["tile_nnz"] == 2 * 1024
    assert_matches_oracle(dup, Bdup, run_plan(pd, Bdup))


def test_measured_per_cu_imbalance_report():
    """flex_plan_measure_imbalance (≙ the reference's per-SM "Imb" column, flex.cu:27-79, 5087-5126): one launch of the
    stamped twin of the kernel; the C it leaves is the ordinary result, every CU of the chip shows up, the numbers are
    sane, and a schedule with all the heavy rows at one end of one XCD's slice reads as more imbalanced than a balanced one."""
    a = flex_amd.synth_graph("flickr")
    k = 128
    B = random_B(a.n, k, 7)
    Bd = dev(B)
    p = Plan(a, k, order=flex_amd.FLEX_ORDER_CLUSTER)
    C = torch.empty((a.m, k), device="cuda")
    im = p.measure_imbalance(Bd.data_ptr(), C.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert_matches_oracle(a, B, C.cpu().numpy(), nthreads=8)
    assert im["cus_seen"] >= 200 and im["xcds_seen"] == 8 and im["waves"] >= p.info()["n_chunks"]
    assert 0.0 <= im["cu_busy_imb_pct"] < 300.0 and 0.0 <= im["cu_end_spread_pct"] <= 100.0
    assert 5.0 < im["span_us"] < 2000.0 and 0 < im["wave_us_mean"] <= im["wave_us_max"]
    # odd k has no stamped twin: refused, not faked
    with pytest.raises(flex_amd.FlexError):
        Plan(a, 7).measure_imbalance(dev(random_B(a.n, 7, 1)).data_ptr(), torch.empty((a.m, 7), device="cuda").data_ptr(), 0)


@pytest.mark.parametrize("k", [128, 100, 32])
def test_mfma_route_keeps_non_finite_values_where_they_belong(k):
    """A routed tile is stored dense, so its absent cells are zeros of the MFMA's A operand: 0 x inf must NOT turn into a NaN
    in a row that does not reference that B row (the vector kernel, the oracle and the reference never touch it).  Block-dense
    input of fill 0.7 (30 % of every routed tile absent), n not a multiple of 32 (the last column tile hangs over the edge and
    is padded with a clamped column), inf / NaN / -inf planted in B rows that sit inside dense blocks, in the overhanging
    tile and among the noise columns; an explicit zero VALUE of A next to an inf must still give NaN, as in the oracle."""
    n = 3210
    a = block_dense_graph(n, 64, 0.7, 5, seed=13)
    a.vals[a.rowPtr[70] + 3] = 0.0  # a stored zero: it references its column like any other entry
    zero_col = int(a.col[a.rowPtr[70] + 3])
    B = random_B(n, k, 6)
    bad_rows = [5, 64 + 17, 1000, n - 1, n - 2, zero_col]
    B[bad_rows[0], :] = np.inf
    B[bad_rows[1], min(5, k - 1)] = np.nan
    B[bad_rows[2], :: 7] = -np.inf
    B[bad_rows[3], :] = np.inf
    B[bad_rows[4], 0] = np.nan
    B[zero_col, :] = np.inf
    p = Plan(a, k, tuning={"mfma": 1, "mfma_fill_pct": 50})
    info = p.info()
    assert info["n_tiles"] > 150 and info["tile_nnz"] > 0.6 * a.nnz, info  # the route is taken
    C = run_plan(p, B)
    gold = oracle.spmm(a.rowPtr, a.col, a.vals, B)
    assert np.array_equal(np.isfinite(C), np.isfinite(gold))
    assert np.array_equal(np.isnan(gold[70]), np.isnan(C[70])) and np.isnan(gold[70]).any()  # the stored zero times inf
    fin = np.isfinite(gold)
    assert np.allclose(C[fin], gold[fin], rtol=1e-5, atol=1e-5)
    # and the finite rows are bit-identical to what the same plan gives on a finite B (the masked pass only adds non-finite terms)
    Bf = B.copy()
    Bf[~np.isfinite(Bf)] = 0.25
    rows_touched = np.zeros(n, bool)
    rows = np.repeat(np.arange(n), np.diff(a.rowPtr.astype(np.int64)))
    rows_touched[rows[np.isin(a.col, bad_rows)]] = True
    Cf = run_plan(p, Bf)
    assert np.array_equal(C[~rows_touched], Cf[~rows_touched])
