#!/usr/bin/env python3
"""Time-bounded randomized soak of the HIP path against the CPU oracle (a checker, like tests/): on top of what the
seeded 250-case fuzz in tests/ varies (shape, degree, k, ordering, strides) it draws the PLAN-TIME KNOBS (column-tile
width, chunk budget, piece size, 2-D column panels, dense-tile routing and its threshold, two-launch split rows,
non-temporal records, gathers in flight, XCD remap) and the FORM of the plan (whole / mapped / row shards with a
column map), plants dense blocks and hub rows, launches every plan twice and compares the two results bit for bit.

usage: tools/soak_gpu.py [seconds=300] [seed]        -> one line per 25 cases, a summary line, exit 1 on a mismatch
"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import flex_amd, oracle
import tools._knobs  # noqa: E402,F401  (FLEX_* environment knobs -> plan descriptor)
from flex_amd import HostCsr, Plan
from util import random_csr

budget_s = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
rng = np.random.default_rng(seed)
KNOBS = {
    "FLEX_LANES": [None, None, "4", "8", "16", "32", "64"],
    # row bundles (short rows side by side, one per record slot): on for half of the cases, every candidate length
    "FLEX_BUNDLE": [None, "1", "1", "2"],
    "FLEX_BUNDLE_LEN": [None, None, "1", "9", "100"],
    "FLEX_WAVE_NNZ": [None, None, "32", "100", "512", "4000"],
    "FLEX_LONG_ROW": [None, None, "40", "300"],
    "FLEX_PIECE": [None, None, "16", "200"],
    "FLEX_2D": [None, None, "1"],
    "FLEX_PANEL_KB": [None, "1", "16", "512"],
    "FLEX_SEG_MIN": [None, "1", "8"],
    "FLEX_MFMA": [None, None, "1", "2"],
    "FLEX_MFMA_FILL": [None, "5", "30", "90"],
    "FLEX_FUSED_FIXUP": [None, None, "1", "2"],  # 1 = in-launch (opt-in since ABI 3), 2 = two launches (default)
    "FLEX_REC_NT": [None, "1", "2"],
    "FLEX_U": [None, None, "8"],
    "FLEX_XCD_REMAP": [None, "1", "2", "3"],
    "FLEX_XCD_STRETCH": [None, "1", "5", "64"],
    "FLEX_XCD_BALANCE": [None, None, "2"],
    "FLEX_LDS_EXTRA": [None, None, "16384"],
    "FLEX_HOST_THREADS": [None, "1", "3", "16"],
    # the row-block path (LDS-staged B panels): on for a third of the cases, every shape of its image
    "FLEX_BLOCKS": [None, None, "1"],
    "FLEX_BLOCK_ROUNDS": [None, "2", "4", "8"],
    "FLEX_BLOCK_PANEL_ROWS": [None, "8", "64", "128", "200"],
    "FLEX_BLOCK_THR": [None, "1", "2", "3", "6"],
    "FLEX_BLOCK_CAP": [None, "8", "30", "200"],
}


def with_blocks(a, rng):
    """Plant a few dense 32-aligned blocks (what the dense-tile detector looks for) into a random CSR; duplicates stay in."""
    m, n = a.m, a.n
    rows = [list(zip(a.col[a.rowPtr[r]:a.rowPtr[r + 1]].tolist(), a.vals[a.rowPtr[r]:a.rowPtr[r + 1]].tolist())) for r in range(m)]
    for _ in range(int(rng.integers(1, 6))):
        r0 = int(rng.integers(0, max(1, m // 32))) * 32
        c0 = int(rng.integers(0, max(1, n // 32))) * 32
        h, w = int(rng.choice([32, 32, 64, 96])), int(rng.choice([32, 64, 128]))
        fill = float(rng.choice([0.3, 0.7, 1.0]))
        for r in range(r0, min(m, r0 + h)):
            for c in range(c0, min(n, c0 + w)):
                if rng.random() < fill:
                    rows[r].append((c, float(rng.uniform(-1, 1))))
    rp = np.zeros(m + 1, dtype=np.int64)
    rp[1:] = np.cumsum([len(r) for r in rows])
    col = np.array([c for r in rows for c, _ in r], dtype=np.uint32)
    vals = np.array([v for r in rows for _, v in r], dtype=np.float32)
    return HostCsr(rp.astype(np.uint32), col, vals, n=n)


def launch(p, Bd, rows, ldc, k):
    Cd = torch.full((rows, ldc), 3.5, dtype=torch.float32, device="cuda")
    p.spmm(Bd.data_ptr(), Cd.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    C1 = Cd.cpu().numpy()
    Cd.fill_(-1.25)
    p.spmm(Bd.data_ptr(), Cd.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    C2 = Cd.cpu().numpy()
    assert np.all(C1[:, k:] == 3.5) and np.all(C2[:, k:] == -1.25), "padding columns of C were written"
    assert np.array_equal(C1[:, :k].view(np.uint32), C2[:, :k].view(np.uint32)), "two launches of one plan differ bitwise"
    return np.ascontiguousarray(C1[:, :k])


t_end = time.time() + budget_s
case = fails = 0
forms = {"whole": 0, "mapped": 0, "shards": 0}
print(f"soak seed {seed}, {budget_s:.0f} s", flush=True)
while time.time() < t_end:
    case += 1
    env = {}
    for name, choices in KNOBS.items():
        os.environ.pop(name, None)
        v = choices[int(rng.integers(0, len(choices)))]
        if v is not None and (rng.random() < 0.6 or name == "FLEX_BLOCKS"):
            env[name] = v
    os.environ.update(env)
    m = int(rng.choice([1, 31, 64, 65, 300, 1500, 4000, 20000, 60000]))
    form = str(rng.choice(["whole", "whole", "mapped", "shards"])) if m >= 64 else "whole"
    square = form != "whole" or bool(rng.integers(0, 2)) or m < 3
    n = m if square else int(rng.choice([1, 5, 97, 1000, 5000, 70000]))
    avg = float(rng.choice([0.0, 0.5, 3, 12, 30, 140]))
    if m >= 20000 and avg > 30: avg = 30.0
    long_rows = {}
    if m >= 300 and rng.integers(0, 2):
        long_rows = {int(rng.integers(0, m)): int(min(n, rng.integers(200, 6000))) for _ in range(int(rng.integers(1, 5)))}
    a = random_csr(m, n, min(avg, n), seed=int(rng.integers(1 << 30)), long_rows=long_rows, empty_frac=float(rng.choice([0.0, 0.1, 0.6])),
                   sorted_cols=bool(rng.integers(0, 2)))
    if 64 <= m <= 4000 and n >= 64 and rng.integers(0, 2):
        a = with_blocks(a, rng)
    k = int(rng.choice([1, 4, 5, 8, 16, 32, 36, 64, 100, 128, 132, 256, 300]))
    strided = bool(rng.integers(0, 3) == 0)
    ldb = k + int(rng.choice([0, 4, 28])) if strided else k
    ldc = k + int(rng.choice([0, 4, 28])) if strided else k
    order = int(rng.choice([flex_amd.FLEX_ORDER_NATURAL, flex_amd.FLEX_ORDER_RCM, flex_amd.FLEX_ORDER_CLUSTER])) if square and form == "whole" else 0
    B = rng.uniform(-1, 1, size=(n, ldb)).astype(np.float32)
    Bd = torch.from_numpy(B).cuda()
    tag = f"case {case} seed {seed}: form={form} m={m} n={n} nnz={a.nnz} avg={avg} k={k} order={order} ld=({ldb},{ldc}) long={long_rows} env={env}"
    ld = dict(ldb=ldb, ldc=ldc) if strided else {}
    try:
        gold = oracle.spmm(a.rowPtr, a.col, a.vals, np.ascontiguousarray(B[:, :k]), nthreads=8)
        if form == "whole":
            p = Plan(a, k, order=order | flex_amd.FLEX_PLAN_STATS, **ld)
            p.self_check()
            got = launch(p, Bd, m, ldc, k)
            p.destroy()
        else:
            rank = flex_amd.order_cluster(a) if rng.integers(0, 2) else flex_amd.order_rcm(a)
            vo, ap = flex_amd.perm_csr(a, rank)
            if form == "mapped":
                p = Plan(ap, k, vo_mp=vo, **ld)
                p.self_check()
                got = launch(p, Bd, m, ldc, k)
                p.destroy()
            else:
                cuts = sorted({0, m, *(int(x) for x in rng.integers(0, m + 1, size=int(rng.integers(1, 4))))})
                got = np.zeros_like(gold)
                for r0, r1 in zip(cuts[:-1], cuts[1:]):
                    p = Plan(ap, k, rows=(r0, r1), col_map=vo, **ld)
                    p.self_check()
                    got[vo[r0:r1]] = launch(p, Bd, r1 - r0, ldc, k)
                    p.destroy()
        cnt, max_err, me_nnz, _ = oracle.rescheck(gold, got, a.rowPtr)
        assert cnt == 0, f"{cnt} mismatches, max err {max_err:g} on a row of {me_nnz} nnz"
        forms[form] += 1
    except Exception as e:  # keep going: one line per failure, the exit code says whether there was one
        fails += 1
        print(f"FAIL {tag}: {type(e).__name__}: {e}", flush=True)
        if fails >= 10: break
    if case % 25 == 0:
        print(f"{case} cases, {fails} failures, {t_end - time.time():.0f} s left", flush=True)
print(f"soak done: seed {seed}, {case} cases ({forms}), {fails} failures")
sys.exit(1 if fails else 0)
