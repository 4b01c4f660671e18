"""N>1 path on CPU: world_size-2 `gloo` run of the sharding / broadcast / assembly logic.

The per-rank kernel needs a GPU, so the CPU ranks compute their slice with the oracle (test
infrastructure) -- what is under test here is everything AROUND the kernel: every rank derives the
same re-ordered matrix and the same row boundaries, the slice handed to the plan uses the original
column numbering, B arrives by one broadcast, and the scattered C slices assemble to the full result.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import flex_amd
import oracle
from conftest import ROOT

WORKER = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, %(root)r)
import flex_amd, oracle

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
order, k, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
a = flex_amd.synth_graph(n=4000, nnz=4000 + 2 * 30000, community=80, p_in=0.6, p_near=0.2, seed=77)
shard = flex_amd.make_shard(a, k, rank, world, order=order)
# B exists on rank 0 only and reaches the others by ONE broadcast
B = torch.from_numpy(np.random.default_rng(5).uniform(-1, 1, (a.n, k)).astype(np.float32)) if rank == 0 \
    else torch.zeros((a.n, k), dtype=torch.float32)
flex_amd.broadcast_dense(B, src=0)
rp, cols, vals = shard.local_csr()
C_local = oracle.spmm(rp, cols, vals, B.numpy())          # stand-in for plan(B) on a GPU rank
# assemble on rank 0: gather (original row ids, C slice)
rows = torch.from_numpy(shard.original_rows().astype(np.int64))
sizes = [None] * world
dist.all_gather_object(sizes, int(rows.numel()))
gathered = [None] * world
dist.all_gather_object(gathered, (rows.numpy(), C_local, shard.bounds.tolist(), shard.nnz))
if rank == 0:
    full = np.zeros((a.m, k), dtype=np.float32)
    seen = np.zeros(a.m, dtype=np.int64)
    for r, c, _, _ in gathered:
        full[r] = c
        seen[r] += 1
    np.savez(out, full=full, seen=seen, B=B.numpy(), bounds=np.array(gathered[0][2]),
             same_bounds=all(g[2] == gathered[0][2] for g in gathered), nnz=np.array([g[3] for g in gathered]))
dist.barrier()
dist.destroy_process_group()
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("order", ["cluster", "rcm", "natural"])
def test_two_rank_row_sharding_assembles_full_result(tmp_path, order):
    pytest.importorskip("torch")
    k, world = 16, 2
    out = str(tmp_path / f"res_{order}.npz")
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script), order, str(k), out], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    for p in procs:
        log, _ = p.communicate(timeout=300)
        assert p.returncode == 0, log
    res = np.load(out)
    a = flex_amd.synth_graph(n=4000, nnz=4000 + 2 * 30000, community=80, p_in=0.6, p_near=0.2, seed=77)
    assert bool(res["same_bounds"]) and np.all(res["seen"] == 1)          # every row owned by exactly one rank
    assert res["bounds"][0] == 0 and res["bounds"][-1] == a.m
    assert res["nnz"].sum() == a.nnz and res["nnz"].min() > 0.35 * a.nnz  # cost-balanced shards
    gold = oracle.spmm(a.rowPtr, a.col, a.vals, res["B"])
    cnt, max_err, _, _ = oracle.rescheck(gold, res["full"], a.rowPtr)
    assert cnt == 0, (cnt, max_err)


def test_shard_local_csr_uses_original_columns():
    a = flex_amd.synth_graph(n=1500, nnz=1500 + 2 * 6000, community=50, p_in=0.6, p_near=0.2, seed=3)
    B = np.random.default_rng(1).uniform(-1, 1, (a.n, 8)).astype(np.float32)
    gold = oracle.spmm(a.rowPtr, a.col, a.vals, B)
    for world in (1, 3, 8):
        got = np.zeros_like(gold)
        for r in range(world):
            sh = flex_amd.make_shard(a, 8, r, world, order="cluster")
            rp, cols, vals = sh.local_csr()
            got[sh.original_rows()] = oracle.spmm(rp, cols, vals, B)
        assert oracle.rescheck(gold, got, a.rowPtr)[0] == 0


BCAST_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
import flex_amd
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
ok = True
for shape in ((1001, 5), (999, 8), (3, 4), (1, 2), (0, 4)):   # rows that do and do not divide by 3, fewer rows than ranks, none
    for method in ("scatter_allgather", "broadcast"):
        for src in (0, 2):
            ref = torch.from_numpy(np.random.default_rng(hash((shape, src)) %% 1000).uniform(-1, 1, shape).astype(np.float32))
            B = ref.clone() if rank == src else torch.full(shape, 7.0)
            flex_amd.broadcast_dense(B, src=src, method=method)
            ok = ok and bool(torch.equal(B, ref))
flags = [None] * world
dist.all_gather_object(flags, ok)
if rank == 0:
    open(sys.argv[1], "w").write("ok" if all(flags) else "bad")
dist.barrier()
dist.destroy_process_group()
'''


def test_three_rank_broadcast_of_the_dense_operand(tmp_path):
    """broadcast_dense: scatter + all-gather (the xGMI-friendly form) and the plain broadcast deliver the
    same bytes for every shape, from any source."""
    pytest.importorskip("torch")
    world = 3
    out = str(tmp_path / "bcast.txt")
    script = tmp_path / "bcast_worker.py"
    script.write_text(BCAST_WORKER % {"root": ROOT})
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script), out], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    for p in procs:
        log, _ = p.communicate(timeout=300)
        assert p.returncode == 0, log
    assert open(out).read() == "ok"


def _launch_bench(world, extra, timeout=600, env_extra=None, ranks=None):
    port = _free_port()
    procs = []
    for r in (range(world) if ranks is None else ranks):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1", **(env_extra or {}))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world)] + extra, env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT))
    outs = [p.communicate(timeout=timeout) for p in procs]
    return procs, outs


def test_eight_rank_dry_run_of_the_bench_on_a_scaled_down_amazon():
    """bench.py --dry-run: the N = 8 strong-scaling path the driver launches (same preset scaled down, same re-ordering,
    same flex_shard_rows, one broadcast of B) with 8 `gloo` ranks and no GPU: every rank must derive the same shards,
    receive the same B, and rank 0's JSON line must carry the per-rank fields."""
    import json
    pytest.importorskip("torch")
    procs, outs = _launch_bench(8, ["--dry-run", "--workload", "amazon", "--shrink", "512", "--scaling", "strong", "--bcast", "scatter_allgather",
                                    "--host-threads", "2"])
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]
    lines = [ln for so, _ in outs for ln in so.splitlines() if ln.startswith("{")]
    assert len(lines) == 1  # rank 0 only
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["scaling"] == "strong" and j["dry_run"] and j["shards_consistent"]
    c = j["config"]
    assert len(c["per_rank_ms"]) == 8 and len(c["per_rank_nnz"]) == 8 and sum(c["per_rank_nnz"]) == c["nnz"]
    assert sum(c["per_rank_rows"]) == c["n"] and min(c["per_rank_rows"]) > 0
    assert c["shard_nnz_imbalance_pct"] < 25.0 and "amazon/512" in c["workload"]
    # multi-GPU readiness (verdict r02, item 4): the ordering runs on rank 0 ONLY and is broadcast; every rank reports its planning
    # time and its share of the host's threads; the line says how many ranks the collective library itself counted
    assert len(c["per_rank_order_s"]) == 8 and c["per_rank_order_s"][0] > 0 and all(t == 0 for t in c["per_rank_order_s"][1:]), c["per_rank_order_s"]
    assert len(c["per_rank_plan_s"]) == 8 and min(c["per_rank_plan_s"]) > 0
    assert c["per_rank_host_threads"] == [2] * 8
    assert c["rccl_world"] == 0  # gloo rehearsal: present, and says that RCCL was not involved


def test_default_host_threads_are_the_ranks_share_of_the_cores_and_the_perm_cache_round_trips(tmp_path):
    """Without --host-threads a rank takes cores / ranks-on-this-node threads (8 ranks x 32 planner threads on one host otherwise);
    --perm-cache: the first run computes the ordering on rank 0 and writes it, the second loads it and nobody orders."""
    import json
    pytest.importorskip("torch")
    cache = str(tmp_path / "amazon.perm")
    want = max(1, min(32, (os.cpu_count() or 1) // 4))
    for state in ("written", "loaded"):
        procs, outs = _launch_bench(4, ["--dry-run", "--workload", "amazon", "--shrink", "1024", "--perm-cache", cache])
        for p, (so, se) in zip(procs, outs):
            assert p.returncode == 0, se[-2000:]
        j = json.loads([ln for so, _ in outs for ln in so.splitlines() if ln.startswith("{")][0])
        c = j["config"]
        assert j["shards_consistent"] and c["plan"]["perm_cache"] == state and c["per_rank_host_threads"] == [want] * 4
        if state == "written":
            assert c["per_rank_order_s"][0] > 0 and all(t == 0 for t in c["per_rank_order_s"][1:]) and os.path.getsize(cache) > 0
        else:
            assert all(t == 0 for t in c["per_rank_order_s"])


def test_a_rank_that_cannot_reach_the_others_exits_nonzero_instead_of_hanging():
    """Start-up failure path: world size 2 but only rank 1 is launched; with --init-timeout the rendezvous gives up, the rank
    prints which layer failed and exits with a non-zero status (never a hang, never a silent fallback)."""
    pytest.importorskip("torch")
    import time
    t0 = time.time()
    procs, outs = _launch_bench(2, ["--dry-run", "--workload", "pubmed", "--init-timeout", "5"], timeout=120, ranks=[1])
    assert procs[0].returncode not in (0, None), outs[0]
    assert "initialisation failed" in outs[0][1] or "imed out" in outs[0][1] or "onnect" in outs[0][1], outs[0][1][-1500:]
    assert time.time() - t0 < 100


FAILING_WORKER = r'''
import os, sys, time
import numpy as np
import torch.distributed as dist
sys.path.insert(0, %(root)r)
import flex_amd
from flex_amd.multigpu import OrderingFailed, shared_ordering

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
mode = sys.argv[1]
n = 64
if mode == "gorder-isolated":  # a ring with vertex 5 cut out of it: Gorder (order_gorder.cu: every vertex needs a neighbour) refuses
    keep = [v for v in range(n) if v != 5]
    rows = {v: [] for v in range(n)}
    for i, v in enumerate(keep):
        w = keep[(i + 1) %% len(keep)]
        rows[v].append(w); rows[w].append(v)
    rp = np.cumsum([0] + [len(rows[v]) for v in range(n)]).astype(np.uint32)
    col = np.array([c for v in range(n) for c in sorted(rows[v])], dtype=np.uint32)
    a = flex_amd.HostCsr(rp, col, np.ones(len(col), np.float32), n=n)
    order, cache = "gorder", None
else:                          # the permutation cache cannot be written: the ordering succeeds, perm_save raises
    a = flex_amd.synth_graph(n=2000, nnz=2000 + 2 * 8000, community=50, p_in=0.6, p_near=0.2, seed=3)
    order, cache = "cluster", "/nonexistent-dir-for-this-test/x.perm"
t0 = time.time()
try:
    shared_ordering(a, order, {}, cache=cache)
except OrderingFailed as e:
    print(f"rank {rank}: OrderingFailed after {time.time() - t0:.1f} s: {e}", flush=True)
    os._exit(7)  # a fresh exit: no destructor waits on the group
print(f"rank {rank}: no failure", flush=True)
os._exit(0)
'''


@pytest.mark.parametrize("mode", ["gorder-isolated", "cache-unwritable"])
def test_rank0_failing_to_order_fails_every_rank_instead_of_stranding_them(tmp_path, mode):
    """The N>1 failure path of shared_ordering: whatever rank 0 raises while ordering (Gorder given an isolated vertex) or while
    saving the permutation cache becomes a state word in the SAME broadcast, and every rank raises together -- no rank is left in
    the collective until the process-group timeout (the sibling of the cache-file race fixed in round 3)."""
    pytest.importorskip("torch")
    import time
    world, port = 3, _free_port()
    script = tmp_path / "w.py"
    script.write_text(FAILING_WORKER % {"root": ROOT})
    t0 = time.time()
    procs = [subprocess.Popen([sys.executable, str(script), mode],
                              env=dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1"),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = [p.communicate(timeout=240) for p in procs]
    assert time.time() - t0 < 200
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 7, (r, p.returncode, so[-800:], se[-800:])
        assert "OrderingFailed" in so and "rank 0 could not produce" in so, so
    assert ("FlexError" in outs[0][0]) or ("Error" in outs[0][0])  # rank 0 says what it was
