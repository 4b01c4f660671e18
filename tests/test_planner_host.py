"""The planner's host logic WITHOUT a GPU: every kind of plan is created through the C ABI of a host-simulated build
(tests/hostsim: same sources, "device memory" = malloc, no kernels) and its image is verified by flex_plan_self_check --
chunks tile the tasks, tasks tile the records, every C row has exactly one writer, pieces name their row, the dense-tile
directory is consistent -- for the 1-D schedule, the 2-D column-panel schedule, the MFMA dense-tile route, mapped plans,
row shards, padded storage, every column-tile width.  flex_spmm must refuse (no CPU compute path exists)."""
import ctypes as C
import os

import numpy as np
import pytest

import flex_amd
from flex_amd import binding
from util import random_csr

hostsim = pytest.importorskip("hostsim")


@pytest.fixture(scope="module")
def sim():
    """binding.py pointed at the host-simulated library for the duration of this module."""
    so = os.environ.get("FLEX_HOSTSIM_LIB") or hostsim.build()
    old_so, old_lib = binding._SO, binding._lib
    binding._SO, binding._lib = so, None
    yield binding.lib()
    binding._SO, binding._lib = old_so, old_lib


def block_dense(n, block, fill, noise, seed):
    rng = np.random.default_rng(seed)
    nb = n // block
    b, r, c = np.nonzero(rng.random((nb, block, block)) < fill)
    rows = np.concatenate([b * block + r, np.repeat(np.arange(nb * block), noise)])
    cols = np.concatenate([b * block + c, rng.integers(0, nb * block, size=nb * block * noise)])
    key = np.unique(rows.astype(np.int64) * (nb * block) + cols)
    r, c = key // (nb * block), key % (nb * block)
    rp = np.zeros(nb * block + 1, dtype=np.int64)
    np.cumsum(np.bincount(r, minlength=nb * block), out=rp[1:])
    return flex_amd.HostCsr(rp.astype(np.uint32), c.astype(np.uint32), rng.uniform(-1, 1, len(c)).astype(np.float32), n=nb * block)


def test_hostsim_cannot_compute(sim):
    a = random_csr(200, 200, 5, seed=1)
    p = flex_amd.Plan(a, 32)
    p.self_check()
    with pytest.raises(flex_amd.FlexError, match="not supported"):
        p.spmm(0x1000, 0x2000, 0)  # no kernels in this build: the launch is refused, nothing is computed
    assert not hasattr(sim, "flex_spmm_host")


@pytest.mark.parametrize("k", [1, 7, 32, 100, 128, 256, 512])
@pytest.mark.parametrize("order", [flex_amd.FLEX_ORDER_NATURAL, flex_amd.FLEX_ORDER_RCM, flex_amd.FLEX_ORDER_CLUSTER, flex_amd.FLEX_ORDER_GORDER])
def test_one_d_plans_are_partitions(sim, k, order):
    a = flex_amd.synth_graph(n=5000, nnz=5000 + 2 * 60000, community=100, p_in=0.55, p_near=0.25, seed=3)
    p = flex_amd.Plan(a, k, order=order | flex_amd.FLEX_PLAN_STATS)
    p.self_check()
    info, st = p.info(), p.stats()
    assert info["nnz"] == a.nnz and info["n_chunks"] <= info["n_slots"] and st["records"] + info["tile_nnz"] >= a.nnz
    assert info["n_records"] == st["records"]
    # the LDS-level reuse report (flex_plan_stats.lds_*): hot share falls and u rises with the threshold; a community order finds more
    assert 0 <= st["lds_hot_pct_4"] <= st["lds_hot_pct_2"] <= 100 and (st["lds_hot_pct_4"] == 0 or st["lds_u_4"] >= max(4.0, st["lds_u_2"]))
    if order == flex_amd.FLEX_ORDER_CLUSTER:
        assert st["lds_hot_pct_2"] > 40 and st["lds_u_2"] >= 2


def test_ragged_shapes_shards_maps_and_strides(sim):
    a = random_csr(4000, 4000, 25, seed=7, long_rows={3: 3900, 2000: 1500, 3999: 700}, empty_frac=0.2)
    for k in (32, 128):
        flex_amd.Plan(a, k).self_check()
        flex_amd.Plan(a, k, ldb=k + 32, ldc=k + 4).self_check()
    rank = flex_amd.order_cluster(a)
    vo, ap = flex_amd.perm_csr(a, rank)
    flex_amd.Plan(ap, 128, vo_mp=vo).self_check()
    bounds = flex_amd.shard_rows(ap, 128, 5)
    total = 0
    for i in range(5):
        p = flex_amd.Plan(ap, 128, rows=(bounds[i], bounds[i + 1]), col_map=vo)
        p.self_check()
        total += p.info()["nnz"]
    assert total == a.nnz
    pe = flex_amd.Plan(ap, 64, rows=(0, 0), col_map=vo, ldb=96, ldc=64)  # an empty shard through the descriptor form
    assert pe.info()["m"] == 0
    rect = random_csr(900, 5000, 40, seed=8, long_rows={5: 4000})
    flex_amd.Plan(rect, 128).self_check()
    empty = flex_amd.HostCsr(np.zeros(6, np.uint32), np.zeros(0, np.uint32), np.zeros(0, np.float32), n=5)
    flex_amd.Plan(empty, 32).self_check()


@pytest.mark.parametrize("lanes,k", [(8, 32), (8, 128), (16, 128), (32, 128), (64, 256)])
def test_two_d_plans_are_partitions(sim, lanes, k):
    a = flex_amd.synth_graph(n=6000, nnz=6000 + 2 * 90000, community=200, p_in=0.55, p_near=0.3, seed=5)
    knobs = {"two_d": 1, "panel_kb": 32, "seg_min": 2, "lanes_per_nz": lanes}
    for order in (flex_amd.FLEX_ORDER_NATURAL, flex_amd.FLEX_ORDER_CLUSTER):
        p = flex_amd.Plan(a, k, order=order, tuning=knobs)
        t = p.tuning()
        assert all(t[key] == v for key, v in knobs.items()), t
        info = p.info()
        assert info["two_d"] == 1 and info["lanes_per_nz"] == lanes and info["n_split_rows"] > a.m // 2
        p.self_check()


def test_dense_tile_route_is_consistent(sim):
    a = block_dense(6400, 64, 0.85, 5, seed=2)
    p = flex_amd.Plan(a, 128, order=flex_amd.FLEX_ORDER_NATURAL | flex_amd.FLEX_PLAN_STATS)
    info, st = p.info(), p.stats()
    assert info["n_tiles"] == 2 * 2 * (6400 // 64) and info["tile_nnz"] > 0.8 * a.nnz  # every 64-block = four 32x32 tiles
    assert abs(st["mfma_nnz_pct"] - 100.0 * info["tile_nnz"] / a.nnz) < 1e-9 and st["tile_nnz_pct_50"] >= st["mfma_nnz_pct"] - 1e-9
    assert st["records"] + info["tile_nnz"] >= a.nnz
    p.self_check()
    p2 = flex_amd.Plan(a, 128, order=flex_amd.FLEX_PLAN_STATS, tuning={"mfma": 2})
    assert p2.info()["n_tiles"] == 0 and p2.stats()["tile_nnz_pct_50"] == st["tile_nnz_pct_50"]  # the report does not depend on the route
    route = {"mfma": 1, "mfma_fill_pct": 25}
    # a tile row that hangs over the end of the matrix, a shard, and a 2-D plan on top of the route
    odd = block_dense(1000, 40, 0.9, 3, seed=4)
    flex_amd.Plan(odd, 64, tuning=route).self_check()
    flex_amd.Plan(odd, 64, rows=(100, 777), tuning=route).self_check()
    p3 = flex_amd.Plan(a, 128, tuning=dict(route, two_d=1, panel_kb=16))
    assert p3.info()["two_d"] == 1 and p3.info()["n_tiles"] > 0
    p3.self_check()


def test_xcd_slices_are_a_rule_per_ordering_and_a_flag_for_reordered_loaders(sim):
    """Contiguous XCD slices (8 equal, padded eighths of the chunk table) for the community and natural orders; a flat
    table, dealt round-robin by the hardware, for RCM / Gorder and for a mapped plan that says FLEX_PLAN_XCD_INTERLEAVE."""
    a = flex_amd.synth_graph(n=20000, nnz=20000 + 2 * 300000, community=256, p_in=0.6, p_near=0.25, seed=4)
    def shape(**kw):
        p = flex_amd.Plan(a if "vo_mp" not in kw else ap, 128, **kw)
        p.self_check()
        i = p.info()
        return i["n_chunks"], i["n_slots"]
    vo, ap = flex_amd.perm_csr(a, flex_amd.order_rcm(a))
    for order in (flex_amd.FLEX_ORDER_CLUSTER, flex_amd.FLEX_ORDER_NATURAL):
        c, s = shape(order=order)
        assert s % 8 == 0 and s >= c
    for order in (flex_amd.FLEX_ORDER_RCM, flex_amd.FLEX_ORDER_GORDER):
        cr, s = shape(order=order)
        assert s == cr
    cv, s = shape(vo_mp=vo)
    assert s % 8 == 0 and s > cv  # a reordered loader planned as given: slices, unless it asks otherwise
    c2, s2 = shape(vo_mp=vo, order=flex_amd.FLEX_PLAN_XCD_INTERLEAVE)
    assert s2 == c2 == cv
    # stretches of the schedule dealt to the XCDs in turn (tuning.xcd_slices = 3): still eight padded slices, every chunk once
    c, _ = shape(order=flex_amd.FLEX_ORDER_CLUSTER)
    for stretch in (1, 7, 64, 100000):
        p = flex_amd.Plan(a, 128, order=flex_amd.FLEX_ORDER_CLUSTER, tuning={"xcd_slices": 3, "xcd_stretch": stretch})
        p.self_check()
        i, t = p.info(), p.tuning()
        assert i["n_chunks"] == c and i["n_slots"] % 8 == 0 and i["n_slots"] >= c and t["xcd_slices"] == 3 and t["xcd_stretch"] == stretch
    with pytest.raises(flex_amd.FlexError):
        flex_amd.Plan(a, 128, order=0x4000)  # unknown flag bits are refused


def test_planner_result_does_not_depend_on_the_thread_count(sim):
    """The whole device image -- every byte the planner uploads, fingerprinted by the shim -- for 1, 3 and 8 host threads:
    1-D with the community ordering, the 2-D schedule, and the dense-tile route."""
    sim.hostsim_upload_hash.restype = C.c_uint64
    sim.hostsim_upload_hash.argtypes = [C.c_int]
    a = flex_amd.synth_graph(n=20000, nnz=20000 + 2 * 400000, community=256, p_in=0.6, p_near=0.25, seed=9)
    bd = block_dense(4096, 64, 0.8, 6, seed=2)
    cases = [(a, {}), (a, {"two_d": 1}), (bd, {"mfma": 1, "mfma_fill_pct": 50}), (bd, {"two_d": 1, "mfma": 1})]
    for g, env in cases:
        shapes = []
        for threads in (1, 3, 8):
            sim.hostsim_upload_hash(1)
            p = flex_amd.Plan(g, 128, order=flex_amd.FLEX_ORDER_CLUSTER | flex_amd.FLEX_PLAN_STATS, tuning=dict(env, host_threads=threads))
            assert p.tuning()["host_threads"] == threads
            image = sim.hostsim_upload_hash(1)
            p.self_check()
            i, s = p.info(), p.stats()
            shapes.append((image, i["n_tasks"], i["n_chunks"], i["n_slots"], i["n_partials"], i["n_records"], i["n_tiles"], s["cols_wave"], s["cols_xcd"], s["chunk_rec_max"]))
        assert shapes[0] == shapes[1] == shapes[2], env


def test_knobs_travel_in_the_descriptor_not_in_the_environment(sim, monkeypatch):
    """ABI 3: plan-time knobs are fields of flex_plan_tuning.  The environment variables of rounds 1-2 are ignored; two
    threads creating plans with DIFFERENT knobs at the same time each get their own (the getenv form was process-global)."""
    import threading
    a = flex_amd.synth_graph(n=6000, nnz=6000 + 2 * 90000, community=200, p_in=0.55, p_near=0.3, seed=5)
    monkeypatch.setenv("FLEX_LANES", "8")
    monkeypatch.setenv("FLEX_2D", "1")
    monkeypatch.setenv("FLEX_FUSED_FIXUP", "1")
    p = flex_amd.Plan(a, 128)
    t = p.tuning()
    assert p.info()["two_d"] == 0 and t["lanes_per_nz"] == 16 and t["split_rows"] == 1  # rules, whatever the environment says
    # the form of the split-row sum goes by size (in-launch up to 4e8 multiply-adds per launch, two launches above); either can be forced
    assert flex_amd.Plan(a, 128, tuning={"split_rows": 2}).tuning()["split_rows"] == 2
    big = flex_amd.synth_graph(n=60000, nnz=60000 + 2 * 2000000, community=500, p_in=0.55, p_near=0.3, seed=6)
    assert flex_amd.Plan(big, 128).tuning()["split_rows"] == 2 and flex_amd.Plan(big, 128, tuning={"split_rows": 1}).tuning()["split_rows"] == 1
    got, errs = {}, []
    def make(name, knobs, rounds=6):
        try:
            for _ in range(rounds):
                q = flex_amd.Plan(a, 128, order=flex_amd.FLEX_ORDER_CLUSTER, tuning=knobs)
                q.self_check()
                i, tq = q.info(), q.tuning()
                got.setdefault(name, set()).add((i["lanes_per_nz"], i["two_d"], tq["chunk_records"], tq["host_threads"], i["n_chunks"]))
        except Exception as e:  # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=make, args=("x", {"lanes_per_nz": 8, "two_d": 1, "panel_kb": 32, "chunk_records": 96, "host_threads": 2})),
          threading.Thread(target=make, args=("y", {"lanes_per_nz": 32, "chunk_records": 192, "host_threads": 3}))]
    for t_ in th:
        t_.start()
    for t_ in th:
        t_.join()
    assert not errs, errs
    assert len(got["x"]) == 1 and len(got["y"]) == 1, got  # same plan every time, however the two threads interleaved
    (gx, twx, cx, hx, _), = got["x"]
    (gy, twy, cy, hy, _), = got["y"]
    assert (gx, twx, cx, hx) == (8, 1, 96, 2) and (gy, twy, cy, hy) == (32, 0, 192, 3)
    # nonsense is refused, not clamped
    with pytest.raises(flex_amd.FlexError):
        flex_amd.Plan(a, 128, tuning={"chunk_records": -5})
    with pytest.raises(flex_amd.FlexError, match="unknown tuning knob"):
        flex_amd.Plan(a, 128, tuning={"wave_nnz": 5})


def test_descriptor_without_the_range_flag_refuses_a_range(sim):
    """A shard range in a descriptor that lacks FLEX_PLAN_ROW_RANGE (a caller that forgot the flag, or was built before it
    existed) used to give a plan over ALL rows -- and flex_spmm would then write past the shard's C buffer."""
    a = random_csr(300, 300, 6, seed=2)
    v = a.view()
    L = binding.lib()
    for begin, end in ((10, 50), (0, 50), (7, 0)):
        d = binding._PlanDesc(C.sizeof(binding._PlanDesc), C.pointer(v), 32, 0, 0, 0, 0, begin, end, None, None, None)
        h = C.c_void_p()
        assert L.flex_plan_create_ex(C.byref(h), C.byref(d)) == -1
    # a caller built against ABI 2 passes the shorter struct (no tuning member): still accepted
    d = binding._PlanDesc(binding._PlanDesc.tuning.offset, C.pointer(v), 32, 0, 0, 0, 0, 0, 0, None, None, None)
    h = C.c_void_p()
    assert L.flex_plan_create_ex(C.byref(h), C.byref(d)) == 0
    L.flex_plan_destroy(h)


def test_far_first_reorders_records_inside_tasks(sim):
    """tuning.far_first only permutes the records of each task: same tasks, chunks and record count, a different image, still a partition."""
    g = flex_amd.synth_graph(n=12000, nnz=12000 + 2 * 240000, community=300, p_in=0.6, p_near=0.25, seed=12)
    sim.hostsim_upload_hash.restype = C.c_uint64
    sim.hostsim_upload_hash.argtypes = [C.c_int]
    shapes, images = [], []
    for far in (0, 500):
        sim.hostsim_upload_hash(1)
        p = flex_amd.Plan(g, 128, order=flex_amd.FLEX_ORDER_CLUSTER, tuning={"far_first": far})
        images.append(sim.hostsim_upload_hash(1))
        p.self_check()
        i = p.info()
        shapes.append((i["n_tasks"], i["n_chunks"], i["n_records"], i["n_partials"]))
        assert p.tuning()["far_first"] == far
    assert shapes[0] == shapes[1] and images[0] != images[1]


@pytest.mark.parametrize("rounds,panel_rows,thr,cap", [(8, 200, 2, 0), (2, 64, 2, 40), (4, 128, 3, 24), (8, 196, 4, 0), (4, 8, 2, 16)])
def test_hot_block_plans_are_partitions(sim, rounds, panel_rows, thr, cap):
    """The hot-block route (the matrix is split: nonzeros with reuse inside a block of rounds x 60 rows go to LDS-staged B panels,
    the rest stays with the flat planner; a row longer than `cap` is spread over several slots): the device image must be a partition of the work
    (flex_plan_self_check reads the block tables and every record stream back; flat records + hot records = every nonzero) for
    every shape of the knobs."""
    a = random_csr(9000, 9000, 20, seed=31, long_rows={5: 8000, 77: 1200, 4000: 300, 8999: 150}, empty_frac=0.05)
    g = flex_amd.synth_graph(n=12000, nnz=12000 + 2 * 240000, community=300, p_in=0.6, p_near=0.25, seed=12)
    knobs = {"blocks": 1, "block_rounds": rounds, "block_panel_rows": panel_rows, "block_thr": thr, "block_cap": cap}
    for mat, order, k in ((a, flex_amd.FLEX_ORDER_NATURAL, 128), (g, flex_amd.FLEX_ORDER_CLUSTER, 64), (g, flex_amd.FLEX_ORDER_RCM, 100)):
        p = flex_amd.Plan(mat, k, order=order, tuning=knobs)
        p.self_check()
        i, t = p.info(), p.tuning()
        assert t["blocks"] == 1 and t["block_rounds"] == rounds and t["block_panel_rows"] == panel_rows and t["block_thr"] == thr
        if mat is g:
            assert i["n_blocks"] > 0
        assert i["block_rows"] <= mat.m and i["block_nnz"] <= mat.nnz and i["block_records"] >= i["block_hot_nnz"]
        assert i["block_hot_nnz"] <= i["block_nnz"] and i["block_hot_cols"] <= i["block_panels"] * panel_rows
        assert i["n_records"] >= mat.nnz - i["block_hot_nnz"]  # the flat plan holds exactly the nonzeros the blocks do not (+ padding)
        # ... and still every row -- as a task of its own or inside a bundle: the flat kernel writes all of C, the hot kernel adds to it
        assert i["n_tasks"] - i["n_bundles"] + i["bundle_rows"] >= mat.m
        if mat is g and order == flex_amd.FLEX_ORDER_CLUSTER and rounds >= 4 and thr == 2 and panel_rows >= 64:
            assert i["block_hot_nnz"] > 0.3 * mat.nnz  # the planted communities are found as hot columns
    # a row longer than the cap is spread over several slots (parts chained through LDS); with a tiny cap every row is, and rows
    # longer than a whole block of slots hold none
    for tiny in (1, 3):
        pt = flex_amd.Plan(g, 128, order=flex_amd.FLEX_ORDER_CLUSTER, tuning=dict(knobs, block_cap=tiny))
        pt.self_check()
    # shards and mapped plans go through the route too
    vo, gp = flex_amd.perm_csr(g, flex_amd.order_cluster(g))
    flex_amd.Plan(gp, 128, vo_mp=vo, tuning=knobs).self_check()
    pb = flex_amd.Plan(gp, 64, rows=(1000, 7000), col_map=vo, tuning=knobs)
    pb.self_check()
    assert pb.info()["block_rows"] <= 6000
    # shapes the hot kernel does not take (k below one 64-column tile, k % 4 != 0) get a flat plan silently
    assert flex_amd.Plan(g, 7, tuning=knobs).info()["n_blocks"] == 0 and flex_amd.Plan(g, 32, tuning=knobs).info()["n_blocks"] == 0
    # nonsense is refused
    for bad in ({"block_rounds": 3}, {"block_panel_rows": 6}, {"block_panel_rows": 308}):
        with pytest.raises(flex_amd.FlexError):
            flex_amd.Plan(g, 128, tuning=dict(knobs, **bad))
    # the same image for any number of host threads
    sim.hostsim_upload_hash.restype = C.c_uint64
    sim.hostsim_upload_hash.argtypes = [C.c_int]
    images = []
    for threads in (1, 5):
        sim.hostsim_upload_hash(1)
        flex_amd.Plan(g, 128, order=flex_amd.FLEX_ORDER_CLUSTER, tuning=dict(knobs, host_threads=threads))
        images.append(sim.hostsim_upload_hash(1))
    assert images[0] == images[1]


@pytest.mark.parametrize("k,lanes", [(8, 4), (16, 4), (32, 8), (64, 16), (128, 16), (128, 8), (30, 0)])
def test_row_bundle_plans_are_partitions(sim, k, lanes):
    """tuning.bundle = 1: short rows share tasks, one row per record slot (plan_build.cpp, form_tasks).  Every C row still has one
    writer -- a task, a split row or ONE slot of one bundle -- a chunk's bundles name consecutive groups of its part of bd_rows, rows
    without nonzeros and slots without a row sum zeros only (flex_plan_self_check); fewer tasks and fewer records than the plain plan
    of the same matrix, the same nonzeros."""
    a = random_csr(6000, 6000, 6, seed=21, long_rows={17: 5000, 4000: 900}, empty_frac=0.3)
    knobs = {"bundle": 1}
    if lanes:
        knobs["lanes_per_nz"] = lanes
    for order in (flex_amd.FLEX_ORDER_NATURAL, flex_amd.FLEX_ORDER_CLUSTER):
        plain = flex_amd.Plan(a, k, order=order, tuning={**knobs, "bundle": 2})
        p = flex_amd.Plan(a, k, order=order | flex_amd.FLEX_PLAN_STATS, tuning=knobs)
        p.self_check()
        info, t = p.info(), p.tuning()
        slots = 64 // info["lanes_per_nz"]
        assert t["bundle"] == 1 and t["bundle_len"] == (12 if slots >= 8 else 16) and plain.tuning()["bundle"] == 2 and plain.info()["n_bundles"] == 0
        assert info["n_bundles"] > 0 and info["bundle_rows"] >= 2 * info["n_bundles"] and info["bundle_rows"] <= slots * info["n_bundles"]
        assert info["n_tasks"] < plain.info()["n_tasks"] and info["n_split_rows"] == plain.info()["n_split_rows"] > 0
        assert info["n_records"] < plain.info()["n_records"] and info["n_records"] >= a.nnz
    # mapped plans, row shards (an empty one too), padded storage; the candidate length is a knob
    rank = flex_amd.order_cluster(a)
    vo, ap = flex_amd.perm_csr(a, rank)
    pm = flex_amd.Plan(ap, k, vo_mp=vo, tuning=knobs)
    pm.self_check()
    assert pm.info()["n_bundles"] > 0
    if k % 4 == 0:
        bounds = flex_amd.shard_rows(ap, k, 3)
        for i in range(3):
            ps = flex_amd.Plan(ap, k, rows=(bounds[i], bounds[i + 1]), col_map=vo, ldb=k + 8, ldc=k + 4, tuning=knobs)
            ps.self_check()
            assert ps.info()["n_bundles"] > 0
    flex_amd.Plan(ap, k, rows=(5, 5), col_map=vo, tuning=knobs).self_check()
    short = flex_amd.Plan(a, k, tuning={**knobs, "bundle_len": 2})
    short.self_check()
    assert 0 < short.info()["bundle_rows"] < p.info()["bundle_rows"] and short.tuning()["bundle_len"] == 2


def test_row_bundles_of_rows_without_nonzeros_and_of_one_thread_or_five(sim):
    # nothing but empty rows: bundles of zero steps; and the image does not depend on the host thread count
    empty = flex_amd.HostCsr(np.zeros(1001, np.uint32), np.zeros(0, np.uint32), np.zeros(0, np.float32), n=1000)
    pe = flex_amd.Plan(empty, 32, tuning={"bundle": 1})
    pe.self_check()
    assert pe.info()["n_bundles"] == 125 and pe.info()["bundle_rows"] == 1000 and pe.info()["n_records"] == 0
    a = flex_amd.synth_graph(n=20000, nnz=20000 + 2 * 50000, community=100, p_in=0.55, p_near=0.25, seed=9)
    sim.hostsim_upload_hash.restype = C.c_uint64
    sim.hostsim_upload_hash.argtypes = [C.c_int]
    imgs = []
    for threads in (1, 5):
        sim.hostsim_upload_hash(1)
        p = flex_amd.Plan(a, 32, order=flex_amd.FLEX_ORDER_CLUSTER, tuning={"bundle": 1, "host_threads": threads})
        image = sim.hostsim_upload_hash(1)
        p.self_check()
        imgs.append((p.info()["n_tasks"], p.info()["n_records"], p.info()["n_bundles"], p.info()["bundle_rows"], image))
    assert imgs[0] == imgs[1]
    # the 2-D schedule's tasks are runs of a row, not rows: no bundles there; nor on the wide tiles (one or two slots per step)
    for knobs, k in (({"two_d": 1, "panel_kb": 32}, 32), ({"lanes_per_nz": 32}, 128), ({"lanes_per_nz": 64}, 256)):
        p2 = flex_amd.Plan(a, k, tuning={"bundle": 1, **knobs})
        p2.self_check()
        assert p2.info()["n_bundles"] == 0 and p2.tuning()["bundle"] == 2 and p2.tuning()["bundle_len"] == 0


def test_row_bundles_by_rule_where_the_plan_fills_the_chip_or_the_rows_are_short(sim):
    """No knob set: the flickr shape (89 250 rows, degree 11: ~13 000 chunks) gets bundles, on the 16-lane tile from k = 64 up (with
    bundles it beats the wide tile at low degree too) and on the narrow tile at k = 16 although its degree is below 8; so does
    pubmed.csv (19 717 rows of degree 5.5: too few chunks to fill the chip, but its rows are short); a small graph of degree 28
    (the ppi shape's class) keeps the plain plan (plan_build.cpp, bundle_rule)."""
    fl = flex_amd.synth_graph("flickr")
    pm = flex_amd.csv_load(os.path.join(os.path.dirname(__file__), "golden", "pubmed.csv"))
    for g in (fl, pm):
        for k, lanes in ((16, 4), (32, 8), (64, 16), (128, 16), (256, 16)):
            p = flex_amd.Plan(g, k, order=flex_amd.FLEX_ORDER_CLUSTER)
            p.self_check()
            i, t = p.info(), p.tuning()
            assert i["lanes_per_nz"] == lanes and i["n_bundles"] > 1000 and t["bundle"] == 1 and t["bundle_len"] == (12 if lanes <= 8 else 16), (k, i, t)
    small_dense = flex_amd.synth_graph(n=20000, nnz=20000 + 2 * 270000, community=200, p_in=0.6, p_near=0.25, seed=4)
    for k, lanes in ((16, 4), (32, 8), (64, 16), (128, 16)):
        p = flex_amd.Plan(small_dense, k)
        assert p.info()["n_bundles"] == 0 and p.tuning()["bundle"] == 2 and p.tuning()["bundle_len"] == 0 and p.info()["lanes_per_nz"] == lanes
    assert flex_amd.Plan(fl, 128, tuning={"bundle": 2}).info()["lanes_per_nz"] == 32  # without bundles the wide tile stays the rule
    off = flex_amd.Plan(fl, 32, tuning={"bundle": 2})
    assert off.info()["n_bundles"] == 0 and off.info()["n_tasks"] > 3 * flex_amd.Plan(fl, 32).info()["n_tasks"]
