#!/usr/bin/env python3
"""bench.py -- SpMM GFLOPS (2*nnz*k/t) of the MI355X-native engine, one process per GPU.

    python bench.py [--gpus N --steps K --warmup W] [--workload amazon] [--k 128]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one flex_spmm() over the whole (sharded) matrix with A's plan and B resident
in HBM.  Default workload, for every N: the configuration BASELINE.json's metric and target are quoted on --
the Amazon shape (1 569 960^2, 264 339 468 nnz, values U(-1,1)), k=128, fp32 -- a synthetic stand-in with
exactly that n and nnz (the reference ships only pubmed.csv); it fits one GPU (plan 2.3 GB + operands 1.6 GB).
N>1: STRONG scaling of that same matrix (north_star: "rows of A partitioned across the 8 GPUs of one node with B
broadcast once"): every rank derives the same community re-ordering, rows are sharded by flex_shard_rows, B is
broadcast once over RCCL (torch.distributed "nccl") before the timed region; no collective on the data path.
`--workload flickr|reddit|yelp|ppi|pubmed`, `--scaling weak` and `--graph file` remain as options.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and
`cpu_baseline` objects.  The oracle is used here only for the reported CPU baseline.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="amazon", help="synthetic preset: flickr|reddit|amazon|yelp|ppi|pubmed")
    ap.add_argument("--graph", default=None, help="a real graph instead of a preset: .csv (the reference's format), .mtx or .bin "
                                                  "(e.g. tests/golden/pubmed.csv, the one data file the reference ships); strong scaling only")
    ap.add_argument("--variant", default="preset", choices=["preset", "best", "worst"],
                    help="where on the stand-in generator's range the synthetic graph sits: the preset (15 %% uniformly random edges), `best` "
                         "(none: every edge inside a community or its ring) or `worst` (40 %%); same n, nnz and degree law.  The headline is the preset; "
                         "the other two bracket what the numbers owe to the generator (DESIGN.md 3.4).  --workload rmat20: no communities at all")
    ap.add_argument("--k", type=int, default=128)
    ap.add_argument("--order", default="cluster", choices=["cluster", "rcm", "natural"],
                    help="cluster / natural: the row schedule of the plan.  rcm (BASELINE configs[2], 'RCM-reordered rows'): the matrix is RCM-reordered "
                         "on the host as the reference's DataLoaderRcm does (DataLoader.cu:723-787: permuted CSR + vo_mp) and that loader is the plan's "
                         "INPUT; see --schedule for how its rows are then scheduled")
    ap.add_argument("--schedule", default="auto", choices=["auto", "as-given", "cluster"],
                    help="N=1, --order rcm only: `cluster` (= auto) lays the engine's community schedule over the reordered loader "
                         "(flex_plan_create_mapped(A', vo_mp, FLEX_ORDER_CLUSTER)); `as-given` walks the loader's rows in RCM order (rounds 1-3)")
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"])
    ap.add_argument("--shuffle", type=int, default=1, help="0: keep the generator's planted order (locality upper bound)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + FLEX_BENCH_DEVICE=0 rehearses the N>1 path on a one-GPU box (not a measurement)")
    ap.add_argument("--autotune", action="store_true", help="FLEX_PLAN_AUTOTUNE: measure the column-tile width instead of trusting the degree rule (N=1)")
    ap.add_argument("--bcast", default="broadcast", choices=["broadcast", "scatter_allgather"],
                    help="how B reaches the other ranks (untimed, reported as b_bcast_ms)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-copy-probe", action="store_true", help="skip the streaming read / copy probe that measures achievable HBM GB/s")
    ap.add_argument("--no-vendor", action="store_true", help="skip the hipSPARSE side-by-side (N=1 only)")
    ap.add_argument("--no-live-counters", action="store_true",
                    help="do not read the card's memory counters inside the run (N=1 reads them by default, after the timed region: "
                         "flex_amd/counters.py); roofline.traffic then falls back to the committed rocprofv3 figure")
    ap.add_argument("--check", action="store_true", help="verify rank 0's shard against the oracle (small workloads)")
    ap.add_argument("--shrink", type=int, default=1, help="rehearsals only: the preset with n and nnz divided by this (same generator and code path)")
    ap.add_argument("--host-threads", type=int, default=0, help="worker threads of the planner / orderings / generator in THIS process "
                                                               "(default: host cores / ranks on this node, at most 32)")
    ap.add_argument("--perm-cache", default=None, help="file that holds the row ordering (rank[old] = new, keyed by a fingerprint of the matrix): "
                                                       "loaded if it matches, else computed once (by rank 0) and written; saves the 2 s ordering of a re-run")
    ap.add_argument("--tuning", default="", help="plan-time knobs, fields of flex_plan_tuning: e.g. blocks=1,block_rounds=4 (experiments; default: the planner's rules)")
    ap.add_argument("--init-timeout", type=float, default=600.0, help="seconds to wait for the other ranks at start-up before failing (non-zero exit)")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU, nothing measured: every rank runs the host side of the N-GPU path (same graph, same re-ordering, "
                         "flex_shard_rows, the B broadcast over gloo) and rank 0 prints the JSON line with zeros for the times -- "
                         "what the world_size-8 CPU test drives")
    return ap.parse_args()


def _under_a_profiler():
    """rocprofv3 (or any other rocprofiler-sdk tool) already owns the counters of this process: two clients programming the same
    hardware counters is the combination to stay away from."""
    pre = os.environ.get("LD_PRELOAD", "")
    return bool(os.environ.get("ROCP_TOOL_LIBRARIES")) or "rocprofiler" in pre or "rocprofv3" in pre


def main():
    args = parse()
    # In-run memory counters (≙ the NPerf metrics of the reference's run(), flex.cu:4583-4656): the profiler has to be asked for
    # BEFORE the first HIP call of the process.  N=1 only (one process, one card), never under rocprofv3, never in a dry run.
    live = None
    if args.gpus == 1 and not args.dry_run and not args.no_live_counters and not _under_a_profiler():
        try:
            from flex_amd import counters as live
            live.init()
        except Exception as e:  # noqa: BLE001 -- the measurement must not depend on the profiler being usable
            print(f"bench.py: in-run counters unavailable ({type(e).__name__}: {e}); roofline.traffic falls back to profiles/pmc_traffic.json",
                  file=sys.stderr, flush=True)
            live = None
    import torch
    import torch.distributed as dist

    import flex_amd

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if args.dry_run:
        args.backend = "gloo"
        args.no_cpu_baseline = args.no_copy_probe = args.no_vendor = True
    elif not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: there is no CPU fallback for the measured path")
    if "FLEX_BENCH_DEVICE" in os.environ:  # rehearsal only: every rank on one card
        local_rank = int(os.environ["FLEX_BENCH_DEVICE"])
    if not args.dry_run:
        torch.cuda.set_device(local_rank)
    # N ranks on one host plan at the same time: each gets its share of the cores (the planner would otherwise start up to 32
    # threads per rank, 256 on an 8-GPU node)
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    host_threads = args.host_threads or max(1, min(32, (os.cpu_count() or 1) // max(local_world, 1)))
    flex_amd.set_host_threads(host_threads)
    rccl_world = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        tmo = datetime.timedelta(seconds=args.init_timeout)
        try:
            if args.backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank), timeout=tmo)
                probe = torch.ones(1, device=torch.device("cuda", local_rank))
                dist.all_reduce(probe)  # the communicator is created lazily: force RCCL up before any planning work
                torch.cuda.synchronize()
                if int(probe.item()) != world:
                    raise RuntimeError(f"all_reduce over {world} ranks returned {probe.item()}")
                rccl_world = int(probe.item())  # what RCCL itself counted: goes into the JSON line
            else:
                dist.init_process_group("gloo", rank=rank, world_size=world, timeout=tmo)
                probe = torch.ones(1)
                dist.all_reduce(probe)
                if int(probe.item()) != world:
                    raise RuntimeError(f"all_reduce over {world} ranks returned {probe.item()}")
                rccl_world = 0  # a gloo rehearsal: RCCL was not involved
        except Exception as e:  # noqa: BLE001 -- a rank that cannot reach the others must fail the job, not hang it
            print(f"bench.py: rank {rank}: {args.backend} initialisation failed: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
            os._exit(3)  # no destructor may block on a half-made communicator; the launcher tears the other ranks down

    k = args.k
    scale = world if args.scaling == "weak" else 1
    gen_desc = None  # where on the stand-in generator's range a synthetic workload sits (goes into config.workload)
    # ---- workload: synthetic graph with the README's shape (x N vertices and nonzeros when weak-scaling)
    t_gen = time.perf_counter()
    if args.graph:
        if world > 1 and args.scaling != "strong":
            raise SystemExit("--graph is one fixed matrix: use --scaling strong with more than one GPU")
        ext = os.path.splitext(args.graph)[1].lower()
        a = (flex_amd.mtx_load(args.graph, sort_columns=True) if ext == ".mtx" else
             flex_amd.csr_load_bin(args.graph) if ext == ".bin" else flex_amd.csv_load(args.graph))
        args.workload = os.path.splitext(os.path.basename(args.graph))[0] + " (file)"
    else:
        if args.shrink > 1:  # same generator, same structure parameters, n and nnz divided (rehearsals)
            sp = flex_amd.synth_preset(args.workload, scale)
            n_s = max(64, sp.n // args.shrink)
            nnz_s = max(n_s, sp.nnz // args.shrink)
            nnz_s -= (nnz_s - n_s) & 1 if not sp.directed else 0
            a = flex_amd.synth_graph(n=n_s, nnz=nnz_s, alpha=sp.alpha, community=sp.community, p_in=sp.p_in, p_near=sp.p_near,
                                     near_window=sp.near_window, shuffle=bool(args.shuffle), gcn_norm=bool(sp.gcn_norm),
                                     directed=bool(sp.directed), seed=sp.seed)
            args.workload += f"/{args.shrink}"
        elif args.workload.startswith("rmat"):
            a = rmat_graph(int(args.workload[4:] or 20))
            args.workload = f"R-MAT scale {int(args.workload[4:] or 20)} (a/b/c = 0.57/0.19/0.19, symmetrised, relabelled)"
        elif args.variant != "preset":
            sp = flex_amd.synth_preset(args.workload, scale)
            p_in = sp.p_in + (1.0 - sp.p_in - sp.p_near) if args.variant == "best" else max(0.0, 0.60 - sp.p_near)  # random share 0 / 0.40
            a = flex_amd.synth_graph(n=sp.n, nnz=sp.nnz, alpha=sp.alpha, community=sp.community, p_in=p_in, p_near=sp.p_near,
                                     near_window=sp.near_window, shuffle=bool(args.shuffle), gcn_norm=bool(sp.gcn_norm),
                                     directed=bool(sp.directed), seed=sp.seed)
            gen_desc = (f"variant {args.variant}: {100 * p_in:.0f} % of the edges inside a community of ~{sp.community}, {100 * sp.p_near:.0f} % within "
                        f"+-{sp.near_window} communities, {100 * (1 - p_in - sp.p_near):.0f} % uniformly random; power-law degrees alpha={sp.alpha:.2f}")
            args.workload += f" [{args.variant}: {100 * (1 - p_in - sp.p_near):.0f} % random edges]"
        else:
            a = flex_amd.synth_graph(args.workload, scale=scale, shuffle=bool(args.shuffle))
            sp = flex_amd.synth_preset(args.workload, scale)
            if sp.community:  # the headline is a property of the generator point as much as of the kernel: name it
                gen_desc = (f"preset: {100 * sp.p_in:.0f} % of the edges inside a community of ~{sp.community}, {100 * sp.p_near:.0f} % within +-{sp.near_window} "
                            f"communities, {100 * (1 - sp.p_in - sp.p_near):.0f} % uniformly random; power-law degrees alpha={sp.alpha:.2f}"
                            + ("; relabelled at random" if args.shuffle else "; PLANTED order kept"))
    t_gen = time.perf_counter() - t_gen

    # ---- plan: RCM is a schedule (N=1) or an explicit permutation followed by row sharding (N>1)
    t_plan = time.perf_counter()
    order = {"cluster": flex_amd.FLEX_ORDER_CLUSTER, "rcm": flex_amd.FLEX_ORDER_RCM,
             "natural": flex_amd.FLEX_ORDER_NATURAL}[args.order]
    want_stats = False
    schedule_name = f"{args.order} schedule"
    tuning = {kv.split("=")[0].strip(): int(kv.split("=")[1]) for kv in args.tuning.split(",") if kv.strip()} or None
    # the ordering: from the permutation cache when it matches this matrix; else computed ONCE (rank 0, then broadcast)
    timings = {"order_s": 0.0}
    rank_arr, cache_state = None, None
    def ordering_failed(e):
        # rank 0 could not order (or could not read / write the permutation cache): shared_ordering has told every rank through
        # the one broadcast, so all of them arrive here together.  A fresh exit, never a re-exec; no destructor waits on the group.
        print(f"bench.py: rank {rank}: {e}", file=sys.stderr, flush=True)
        sys.stderr.flush()
        os._exit(4)

    if args.perm_cache and args.order != "natural":
        # only rank 0 reads or writes the file; every rank joins the one broadcast (flex_amd/multigpu.py, shared_ordering)
        try:
            rank_arr = flex_amd.multigpu.shared_ordering(a, args.order, timings, cache=args.perm_cache)
        except flex_amd.multigpu.OrderingFailed as e:
            ordering_failed(e)
        cache_state = timings.get("perm_cache")
    if world == 1 and not args.dry_run:
        want_stats = a.nnz <= 50_000_000  # one extra pass over the records: skipped on amazon-size inputs
        flags = (flex_amd.FLEX_PLAN_STATS if want_stats else 0) | (flex_amd.FLEX_PLAN_AUTOTUNE if args.autotune else 0)
        if args.order == "rcm":
            # configs[2]: the RCM-reordered LOADER is the input (the reference's flow: permuted CSR + vo_mp, B and C stay in the
            # original order); the schedule on top of it is the engine's -- communities, unless --schedule as-given
            t_o = time.perf_counter()
            if rank_arr is None:
                rank_arr = flex_amd.order_rcm(a)
                timings["order_s"] = time.perf_counter() - t_o
            vo, ap = flex_amd.perm_csr(a, rank_arr)
            over = flex_amd.FLEX_PLAN_XCD_INTERLEAVE if args.schedule == "as-given" else flex_amd.FLEX_ORDER_CLUSTER
            plan = flex_amd.Plan(ap, k, device=local_rank, vo_mp=vo, tuning=tuning, order=flags | over)
            schedule_name = "RCM-reordered loader (vo_mp), rows walked " + ("as given" if args.schedule == "as-given" else "in the community schedule found on top of it")
            del ap
        elif rank_arr is None:
            plan = flex_amd.Plan(a, k, device=local_rank, order=order | flags, tuning=tuning)
        else:  # the reference's flow: a reordered loader + vo_mp, planned in the order given
            vo, ap = flex_amd.perm_csr(a, rank_arr)
            plan = flex_amd.Plan(ap, k, device=local_rank, vo_mp=vo, tuning=tuning, order=flags)
            del ap
        shard_nnz, shard_rows = a.nnz, a.m
        shard = None
    else:
        try:
            shard = flex_amd.make_shard(a, k, rank, world, order=args.order, rank_arr=rank_arr, timings=timings if rank_arr is None else None)
        except flex_amd.multigpu.OrderingFailed as e:
            ordering_failed(e)
        plan = None if args.dry_run else shard.plan(k, local_rank, tuning=tuning)
        shard_nnz, shard_rows = shard.nnz, shard.r1 - shard.r0
    t_plan = time.perf_counter() - t_plan
    info = plan.info() if plan is not None else {"n_chunks": 0, "n_tasks": 0, "n_split_rows": 0, "lanes_per_nz": 0, "two_d": 0, "n_tiles": 0}

    # ---- B: generated on rank 0, broadcast once over RCCL/xGMI (untimed, reported)
    if args.dry_run:
        return dry_run_tail(args, a, k, rank, world, shard, t_gen, t_plan, timings, host_threads, rccl_world, cache_state)
    dev = torch.device("cuda", local_rank)
    if rank == 0:
        g = torch.Generator(device=dev)
        g.manual_seed(1)
        B = torch.rand((a.n, k), generator=g, device=dev, dtype=torch.float32) * 2 - 1
    else:
        B = torch.empty((a.n, k), device=dev, dtype=torch.float32)
    bcast_ms = 0.0
    if world > 1:
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        if args.backend == "nccl":
            flex_amd.broadcast_dense(B, src=0, method=args.bcast)  # RCCL over xGMI
        else:
            Bh = B.cpu()
            flex_amd.broadcast_dense(Bh, src=0, method=args.bcast)
            B.copy_(Bh)
        torch.cuda.synchronize()
        bcast_ms = (time.perf_counter() - t0) * 1e3
    C = torch.empty((shard_rows, k), device=dev, dtype=torch.float32)
    stream = torch.cuda.current_stream().cuda_stream
    bp, cp = B.data_ptr(), C.data_ptr()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        plan.spmm(bp, cp, stream)
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        plan.spmm(bp, cp, stream)
    ev1.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
    dev_ms = ev0.elapsed_time(ev1)  # HIP events on the launch stream (torch's current stream)
    per_rank = None
    if world > 1:
        # every rank's own figures (rank 0 reports them: which shard was the slow one), then the MAX over ranks
        mine = torch.tensor([wall, dev_ms, float(shard_nnz), float(shard_rows), t_plan, timings["order_s"]],
                            device=dev if args.backend == "nccl" else "cpu", dtype=torch.float64)
        allr = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [[float(x) for x in t_.tolist()] for t_ in allr]
        wall, dev_ms = max(r[0] for r in per_rank), max(r[1] for r in per_rank)

    # ---- in-run counters, AFTER the timed region (N=1): the same launches again, one pass per counter set; the card is
    # otherwise idle, so the card-wide sums are this kernel's
    counted = None
    if live is not None and world == 1:
        try:
            n_cnt = max(1, min(args.steps, 10 if shard_nnz > 1e8 else 30))

            def launches():
                for _ in range(n_cnt):
                    plan.spmm(bp, cp, stream)

            # the counting service numbers the GPU agents as rocminfo does; HIP's ordinal is the same only while no *_VISIBLE_DEVICES
            # variable re-maps the process's view -- with one set, the counters of ANOTHER card would be reported as this launch's
            remapped = [v for v in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "GPU_DEVICE_ORDINAL") if os.environ.get(v)]
            if "FLEX_BENCH_DEVICE" in os.environ and local_rank != 0:
                raise RuntimeError(f"FLEX_BENCH_DEVICE={local_rank}: HIP device {local_rank} need not be counting-service agent {local_rank}")
            counted = live.traffic(launches, device=local_rank, sync=torch.cuda.synchronize, launches=n_cnt)
            # The check that does not depend on how the box numbers its cards: C is written exactly once per launch (+ the partial sums
            # of split rows).  The counters of another card -- idle or busy -- fail it, and then nothing measured is reported.
            c_bytes = 4.0 * shard_rows * k
            hi = 2.6 if (plan.info().get("n_blocks", 0) or plan.info().get("n_tiles", 0)) else 1.5  # hot blocks / dense tiles add to C: a second write of their rows
            if not (0.9 * c_bytes <= counted["write_bytes"] <= hi * c_bytes + (64 << 20)):
                raise RuntimeError(f"counted {counted['write_bytes']:.3g} B written per launch against {c_bytes:.3g} B of C: not this launch's counters"
                                   + (f" (device re-mapping in effect: {', '.join(remapped)})" if remapped else ""))
            counted["device_note"] = (f"{', '.join(remapped)} set; agent {local_rank} accepted because it wrote C's bytes" if remapped else None)
            l2 = live.count(launches, live.L2_PASS, device=local_rank, sync=torch.cuda.synchronize)
            counted["l2_hit_rate"] = l2["TCC_HIT_sum"] / max(1.0, l2["TCC_HIT_sum"] + l2["TCC_MISS_sum"])
            counted["l2_requests"] = l2["TCC_REQ_sum"] / n_cnt
            counted["l2_reads"] = l2["TCC_READ_sum"] / n_cnt
            counted["launches_per_pass"] = n_cnt
            try:  # the instruction mix is a bonus: a card that refuses the SQ pass still reports the memory side
                sq = live.count(launches, live.SQ_PASS, device=local_rank, sync=torch.cuda.synchronize)
                counted["sq"] = {name: v / n_cnt for name, v in sq.items()}
            except Exception as e:  # noqa: BLE001
                print(f"bench.py: SQ pass refused ({e})", file=sys.stderr, flush=True)
        except Exception as e:  # noqa: BLE001
            print(f"bench.py: in-run counters failed ({type(e).__name__}: {e}); roofline.traffic falls back to profiles/pmc_traffic.json",
                  file=sys.stderr, flush=True)
            counted = None

    ok = None
    if args.check and rank == 0:
        import oracle
        Bh = B.cpu().numpy()
        if shard is None:
            rp0, cols, vals = a.rowPtr, a.col, a.vals
        else:
            rp0, cols, vals = shard.local_csr()
        gold = oracle.spmm(rp0, cols, vals, Bh, nthreads=os.cpu_count() or 1)
        cnt, max_err, _, _ = oracle.rescheck(gold, C.cpu().numpy(), rp0)
        ok = {"mismatches": cnt, "max_err": max_err}
        if cnt:
            raise SystemExit(f"bench --check: {cnt} mismatches vs oracle")

    if rank == 0:
        ms_per_step = wall * 1e3 / args.steps
        flops = 2.0 * a.nnz * k  # all ranks together process every nonzero once per step
        gflops = flops / (wall / args.steps) / 1e9
        # roofline of the dominant kernel (spmm_flat_kernel): algorithmic bytes of ONE launch on this
        # rank = rowPtr + (col,val) + B read once + C written once  (flex.cu:4672)
        b_alg = 4.0 * (shard_rows + 1) + 8.0 * shard_nnz + 4.0 * a.n * k + 4.0 * shard_rows * k
        kern_ms = dev_ms / args.steps
        achieved = b_alg / (kern_ms * 1e-3) / 1e9
        g_lanes = max(info["lanes_per_nz"], 1)
        # records of the flat plan; a plan with hot blocks keeps the nonzeros that have reuse on chip in its block image (64-column
        # tiles: 256 bytes of LDS per record and tile, a staged B row per hot column and tile)
        n_blocks = info.get("n_blocks", 0)
        flat_records = float(info.get("n_records", shard_nnz))
        gather_demand = flat_records * 16.0 * g_lanes * -(-k // (4 * g_lanes))  # (the hot blocks' nonzeros read LDS, not the texture path)
        out = {
            "metric": "SpMM GFLOPS (2*nnz*k/t)", "value": round(gflops, 2), "unit": "GFLOPS",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 6), "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "file" if args.graph else "synthetic",
            "config": {
                "workload": (f"{args.workload} (n={a.n}, nnz={a.nnz}), k={k}, fp32, " if args.graph else
                             f"{args.workload}-shape synthetic graph x{scale} (n={a.n}, nnz={a.nnz}"
                             + (f"; generator {gen_desc}" if gen_desc else "") + f"), k={k}, fp32, ")
                            + schedule_name + (f", rows sharded over {world} GPUs, B broadcast once" if world > 1 else ""),
                "n": a.n, "nnz": a.nnz, "k": k, "order": args.order, "schedule": schedule_name, "parallelism": f"row-shard x{world}", "generator": gen_desc,
                "plan": {"chunks": info["n_chunks"], "tasks": info["n_tasks"], "split_rows": info["n_split_rows"],
                         "lanes_per_nz": info["lanes_per_nz"], "two_d": info["two_d"], "mfma_tiles": info["n_tiles"],
                         "blocks": info.get("n_blocks", 0), "block_hot_nnz": info.get("block_hot_nnz", 0), "block_hot_cols": info.get("block_hot_cols", 0),
                         "block_records": info.get("block_records", 0), "block_panels": info.get("block_panels", 0),
                         "bundles": info.get("n_bundles", 0), "bundle_rows": info.get("bundle_rows", 0), "records": info.get("n_records", 0), "plan_s": round(t_plan, 3), "order_s": round(timings["order_s"], 3),
                         "gen_s": round(t_gen, 3), "host_threads": host_threads, "perm_cache": cache_state},
                "b_bcast_ms": round(bcast_ms, 3),
                # ≙ the README's "tPre/tElap" column (README.md:34-42): preprocessing (ordering + planning + upload) over
                # one execution of the kernel
                "tpre_over_telap": round(t_plan * 1e3 / max(kern_ms, 1e-9), 1),
            },
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         # HBM-side bytes of ONE launch: read inside this run from the card's counters when possible (N=1), else
                         # the committed rocprofv3 figure while it applies to the running sources, else null
                         "traffic": int(counted["traffic_bytes"]) if counted else _pmc_traffic(args, world),
                         "traffic_source": (f"in-run: rocprofiler-sdk device counting service, {counted['launches_per_pass']} launches per pass after the "
                                            "timed region, 2*FETCH_SIZE + WRITE_SIZE (KiB)" if counted else
                                            "profiles/pmc_traffic.json (rocprofv3 --pmc passes of tools/pmc.sh)" if _pmc_traffic(args, world) is not None else None),
                         "kernel": "spmm_flat_kernel + spmm_hot_kernel" if info.get("n_blocks", 0) else "spmm_flat_kernel", "kernel_ms": round(kern_ms, 6),
                         "algorithmic_bytes_per_launch": int(b_alg),
                         # what the flat kernel asks of the texture path: every record pulls one B-row segment of 16*G
                         # bytes, once per column tile, L2 hit or not.  NOT a roof: with every gather an L2 hit this kernel moves those
                         # bytes at 22-23 TB/s (`all_l2_hit_ms_measured`, the folded-B probe), and the launch is then still slower
                         # than its L2-miss traffic at the fabric's rate would make it -- both terms are reported, neither is claimed
                         "gather_demand_bytes_per_launch": int(gather_demand),
                         "gather_rate_GBps": round(gather_demand / (kern_ms * 1e-3) / 1e9, 1),
                         "all_l2_hit_ms_measured": _all_hit_ms(args, world)},
        }
        # the fabric term of DESIGN.md 3.4: the launch's L2-miss traffic (committed PMC figure, when it applies to the running
        # sources) at the 6.3 TB/s the fabric delivers
        if counted:
            # ≙ the reference's per-row DRAM bytes, L2 figures and measured B reuse u (flex.cu:5237, 5513-5528: nD = 4/u + ...):
            # u = B bytes the nonzeros ask for / bytes the L2s fetched beyond A's own
            out["roofline"]["traffic_read_bytes"] = int(counted["read_bytes"])
            out["roofline"]["traffic_write_bytes"] = int(counted["write_bytes"])
            if counted.get("device_note"):
                out["roofline"]["traffic_device_note"] = counted["device_note"]
            out["roofline"]["traffic_over_algorithmic"] = round(counted["traffic_bytes"] / b_alg, 2)
            out["roofline"]["l2_hit_rate"] = round(counted["l2_hit_rate"], 4)
            # ≙ the reference's measured L1<->L2 bytes over its estimate ("L2/", flex.cu:5279-5330): requests the L2s received x 128 B,
            # next to what the schedule asks of them (B gathers + the record stream once per column tile + C)
            out["roofline"]["l1_l2_bytes_measured"] = int(counted["l2_requests"] * live.L2_REQUEST_BYTES)
            ktiles = -(-k // (4 * g_lanes))
            est = gather_demand + 8.0 * flat_records * ktiles + 4.0 * shard_rows * k
            if n_blocks:  # + the block image's record streams (once per 64-column tile), its staged panels and its second pass over C
                est += (8.0 * float(info["block_records"]) + 256.0 * float(info["block_hot_cols"])) * -(-k // 64) + 8.0 * float(info["block_rows"]) * k
            out["roofline"]["l1_l2_bytes_over_estimate"] = round(counted["l2_requests"] * live.L2_REQUEST_BYTES / max(est, 1.0), 3)
            # The reference's own `u` (flex.cu:5513-5528) is one level up: nD = bytes L1 reads from L2 per multiply-add = 4/u + (A's share),
            # i.e. how often a B element that reached a CU is used there.  Here A's share is the record stream, re-read once per column tile.
            n_madd = shard_nnz * float(k)
            nd = counted["l2_reads"] * live.L2_REQUEST_BYTES / max(n_madd, 1.0)
            out["roofline"]["nD_l1_from_l2_bytes_per_fma"] = round(nd, 3)
            out["roofline"]["u_l1_measured"] = round(4.0 / max(nd - 8.0 * ktiles / k, 1e-9), 3)
            if "sq" in counted:  # wave instructions per 64 multiply-adds (one wave-wide FMA's worth), ≙ "Per Mult / Num Insns" (flex.cu:5350-5420)
                per = shard_nnz * float(k) / 64.0
                out["roofline"]["wave_insns_per_64_fma"] = {n_[len("SQ_INSTS_"):].lower(): round(v / per, 3) for n_, v in counted["sq"].items() if n_ != "SQ_WAVES"}
                out["roofline"]["waves_per_launch"] = int(counted["sq"]["SQ_WAVES"])
            b_fetched = counted["read_bytes"] - 8.0 * shard_nnz - 4.0 * (shard_rows + 1)
            # null when the L2s served (nearly) everything: a graph whose A and B stay resident in 32 MiB of L2 fetches no B at all
            out["roofline"]["u_measured"] = round(4.0 * shard_nnz * k / b_fetched, 3) if b_fetched > 0.01 * 4.0 * a.n * k else None
            out["roofline"]["u_l2_measured"] = out["roofline"]["u_measured"]  # the same number under its level's name (L2 <- HBM side)
            out["roofline"]["traffic_rocprofv3_committed"] = _pmc_traffic(args, world)  # the offline figure for the same sources, or null
        if out["roofline"]["traffic"] is not None:
            out["roofline"]["model_fabric_ms"] = round(out["roofline"]["traffic"] / 6.3e12 * 1e3, 6)
            # the north_star's "rocprof-measured HBM GB/s against the chip's peak": PMC bytes of the launch (memory side of
            # the L2s: HBM + Infinity Cache) over the launch time measured here; `frac` above is the ALGORITHMIC bytes' share
            out["roofline"]["traffic_GBps"] = round(out["roofline"]["traffic"] / (kern_ms * 1e-3) / 1e9, 1)
            out["roofline"]["traffic_frac_of_peak"] = round(out["roofline"]["traffic"] / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        if world > 1:
            out["config"]["rccl_world"] = rccl_world  # ranks RCCL's own all-reduce counted at start-up (0: gloo rehearsal)
        if per_rank is not None:
            nnzs = [r[2] for r in per_rank]
            out["config"]["per_rank_plan_s"] = [round(r[4], 3) for r in per_rank]
            out["config"]["per_rank_order_s"] = [round(r[5], 3) for r in per_rank]  # the ordering runs on rank 0 only (then broadcast)
            # every rank's own roofline: its shard's algorithmic bytes (rowPtr + records + all of B + its C rows) over its launch time
            prf = []
            for r in per_rank:
                b_r = 4.0 * (r[3] + 1) + 8.0 * r[2] + 4.0 * a.n * k + 4.0 * r[3] * k
                gbps = b_r / (max(r[1], 1e-9) / args.steps * 1e-3) / 1e9
                prf.append({"algorithmic_bytes": int(b_r), "achieved": round(gbps, 1), "frac": round(gbps / HBM_PEAK_GBS, 4)})
            out["roofline"]["per_rank"] = prf
            out["config"]["per_rank_ms"] = [round(r[1] / args.steps, 6) for r in per_rank]        # HIP events, per step
            out["config"]["per_rank_wall_ms"] = [round(r[0] * 1e3 / args.steps, 6) for r in per_rank]
            out["config"]["per_rank_nnz"] = [int(x) for x in nnzs]
            out["config"]["per_rank_rows"] = [int(r[3]) for r in per_rank]
            out["config"]["shard_nnz_imbalance_pct"] = round(100.0 * max(nnzs) * world / max(sum(nnzs), 1.0) - 100.0, 2)
        if world == 1 and want_stats and not n_blocks:  # (with hot blocks the statistics would cover the flat plan's nonzeros only)  ≙ B-Re1 / B-Re2 and alpha_stats_collect (flex.cu:5217-5223, mat.cu:944-1065)
            st = plan.stats()
            out["config"]["plan"].update({
                "b_reuse_wave": round(st["reuse_wave"], 3), "b_reuse_xcd": round(st["reuse_xcd"], 3),
                "chunk_imb_pct": round(st["chunk_imb_pct"], 1), "xcd_imb_pct": round(st["xcd_imb_pct"], 2),
                "split_nnz_pct": round(st["split_nnz_pct"], 2), "pad_pct": round(st["pad_pct"], 2),
                # reuse a workgroup could have above the L2: share of nnz in columns a block of 480 rows uses >= 2 / 4 times, and u
                "lds_hot_pct": [round(st["lds_hot_pct_2"], 1), round(st["lds_hot_pct_4"], 1)], "lds_u": [round(st["lds_u_2"], 2), round(st["lds_u_4"], 2)]})
            # bytes the launch must move if every XCD (private L2) fetches each B row it needs exactly once:
            # the floor of a row-partitioned schedule on this chip, between `algorithmic_bytes_per_launch` and `traffic`
            out["roofline"]["private_l2_model_bytes"] = int(st["l2_bytes"])
        # The side reports below come AFTER the measurement; a failure in one of them is recorded, it does not
        # cost the line (the measured path itself has no fallback and has already succeeded or raised).
        def side(name, fn):
            try:
                return fn()
            except Exception as e:  # noqa: BLE001
                return {"error": f"{name}: {type(e).__name__}: {e}"}

        if world == 1:
            # ≙ the reference's per-SM "Imb" column (flex.cu:5087-5126): one stamped launch AFTER the timed region
            im = side("imbalance", lambda: plan.measure_imbalance(bp, cp, stream))
            out["config"]["plan"]["imbalance"] = im if "error" in im else {
                "cu_busy_imb_pct": round(im["cu_busy_imb_pct"], 2), "cu_end_spread_pct": round(im["cu_end_spread_pct"], 2),
                "xcd_busy_imb_pct": round(im["xcd_busy_imb_pct"], 2), "xcd_end_spread_pct": round(im["xcd_end_spread_pct"], 2),
                "cus_seen": im["cus_seen"], "waves": im["waves"], "wave_us_mean": round(im["wave_us_mean"], 2), "wave_us_max": round(im["wave_us_max"], 2)}
        if world == 1 and not args.no_copy_probe:
            # achievable HBM bandwidth on this box, measured by the library's own streaming kernels
            pr = side("hbm_probe", lambda: flex_amd.hbm_probe(local_rank, mib=2048, reps=10))
            if "error" in pr:
                out["roofline"]["hbm_probe_error"] = pr["error"]
            else:
                out["roofline"]["hbm_read_GBps_measured"] = round(pr["read_GBps"], 1)
                out["roofline"]["hbm_copy_GBps_measured"] = round(pr["copy_GBps"], 1)
        if ok is not None:
            out["check"] = ok
        if world == 1 and not args.no_vendor:
            out["hipsparse"] = side("hipsparse", lambda: vendor_baseline(a, k, B, C))
        if world == 1 and not args.no_cpu_baseline:  # a reported baseline, timed at N=1 only
            out["cpu_baseline"] = side("cpu_baseline", lambda: cpu_baseline(a, k, B))
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def rmat_graph(scale, edge_factor=16, seed=1):
    """A graph WITHOUT communities (what no schedule can help): R-MAT a/b/c = 0.57/0.19/0.19, symmetrised, self loops, vertices
    relabelled at random; values U(-1,1).  scale 20: 1 048 576 vertices, ~32 M nonzeros."""
    import flex_amd
    rng = np.random.default_rng(seed)
    n, m = 1 << scale, edge_factor << scale
    r = np.zeros(m, np.int64)
    c = np.zeros(m, np.int64)
    for lvl in range(scale):
        u = rng.random(m)
        r |= (u >= 0.76).astype(np.int64) << lvl
        c |= (((u >= 0.57) & (u < 0.76)) | (u >= 0.95)).astype(np.int64) << lvl
    perm = rng.permutation(n)
    key = np.unique(np.concatenate([perm[r] * n + perm[c], perm[c] * n + perm[r], np.arange(n) * (n + 1)]))
    rr, cc = key // n, key % n
    rp = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(rr, minlength=n), out=rp[1:])
    return flex_amd.HostCsr(rp.astype(np.uint32), cc.astype(np.uint32), rng.uniform(-1, 1, len(cc)).astype(np.float32), n=n)


def dry_run_tail(args, a, k, rank, world, shard, t_gen, t_plan, timings, host_threads, rccl_world, cache_state):
    """--dry-run: the host side of the N-rank path without a GPU -- B by one broadcast, every rank's shard reported to
    rank 0, which checks that the shards tile the rows and prints the JSON line (times zero: nothing was measured)."""
    import torch
    import torch.distributed as dist
    B = torch.from_numpy(np.random.default_rng(1).uniform(-1, 1, (a.n, k)).astype(np.float32)) if rank == 0 \
        else torch.zeros((a.n, k), dtype=torch.float32)
    t0 = time.perf_counter()
    if world > 1:
        import flex_amd
        flex_amd.broadcast_dense(B, src=0, method=args.bcast)
    bcast_ms = (time.perf_counter() - t0) * 1e3
    mine = (shard.r0, shard.r1, shard.nnz, float(B.double().sum()), [int(x) for x in shard.bounds], t_plan, timings["order_s"], host_threads)
    allr = [None] * world
    if world > 1:
        dist.all_gather_object(allr, mine)
    else:
        allr = [mine]
    if rank == 0:
        ok = (all(r[4] == allr[0][4] for r in allr) and all(abs(r[3] - allr[0][3]) < 1e-6 for r in allr)
              and [r[0] for r in allr] == allr[0][4][:-1] and [r[1] for r in allr] == allr[0][4][1:]
              and sum(r[2] for r in allr) == a.nnz)
        nnzs = [r[2] for r in allr]
        out = {"metric": "SpMM GFLOPS (2*nnz*k/t)", "value": 0.0, "unit": "GFLOPS", "n_gpus": world, "steps": 0, "warmup": 0,
               "ms_per_step": 0.0, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32",
               "data": "synthetic", "dry_run": True, "shards_consistent": bool(ok),
               "config": {"workload": f"{args.workload}-shape synthetic graph (n={a.n}, nnz={a.nnz}), k={k}, fp32, {args.order} schedule, "
                                      f"rows sharded over {world} ranks, B broadcast once (DRY RUN: no GPU, nothing measured)",
                          "n": a.n, "nnz": a.nnz, "k": k, "order": args.order, "parallelism": f"row-shard x{world}",
                          "plan": {"plan_s": round(t_plan, 3), "gen_s": round(t_gen, 3), "host_threads": host_threads, "perm_cache": cache_state},
                          "b_bcast_ms": round(bcast_ms, 3), "rccl_world": rccl_world,
                          "per_rank_ms": [0.0] * world, "per_rank_nnz": nnzs, "per_rank_rows": [r[1] - r[0] for r in allr],
                          "per_rank_plan_s": [round(r[5], 3) for r in allr], "per_rank_order_s": [round(r[6], 3) for r in allr],
                          "per_rank_host_threads": [r[7] for r in allr],
                          "shard_nnz_imbalance_pct": round(100.0 * max(nnzs) * world / max(sum(nnzs), 1) - 100.0, 2)}}
        print(json.dumps(out), flush=True)
        if not ok:
            raise SystemExit("bench --dry-run: the ranks disagree about the shards or about B")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


TRAFFIC_SOURCES = ("spmm_kernels.hip", "plan.h", "plan.cpp", "plan_build.cpp", "dense_tiles.cpp", "internal.h", "cluster.cpp", "synth.cpp",
                   "block_kernels.hip", "block_plan.cpp")


def traffic_source_hash():
    """Fingerprint of the sources that decide how many bytes a launch moves (kernel, planner, ordering, generator):
    their CODE -- `//` comments, blank lines and indentation do not count, so that rewording a comment does not
    void a measurement."""
    import hashlib
    h = hashlib.sha256()
    for f in TRAFFIC_SOURCES:
        for line in open(os.path.join(ROOT, "flex_amd", "csrc", f), encoding="utf-8"):
            code = line.split("//", 1)[0].strip()
            if code:
                h.update(code.encode() + b"\n")
    return h.hexdigest()[:16]


def _all_hit_ms(args, world):
    """The launch time of THIS kernel with every B gather an L2 hit -- measured (tools/probe_bound.py: same plan, the B row a
    record names folded onto 1024 rows), committed as profiles/r03_all_l2_hit.json; null for a workload that was not probed."""
    try:
        e = json.load(open(os.path.join(ROOT, "profiles", "r03_all_l2_hit.json"))).get(f"{args.workload}_k{args.k}")
        return None if e is None or world != 1 or args.order != "cluster" else round(e["us"] / 1e3, 6)
    except Exception:
        return None


def _pmc_traffic(args, world):
    """HBM-side bytes per launch (2*FETCH_SIZE + WRITE_SIZE, MI355X_MICROARCH.md's gfx950 correction) from the committed
    rocprofv3 --pmc passes (profiles/pmc_traffic.json, written by tools/pmc.sh), or null: PMC counters cannot be read
    inside the timed run, so the figure is only reported while the sources it was profiled at are the ones running."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        e = json.load(open(p)).get(f"{args.workload}_k{args.k}_{args.order}_n{world}")
        if not isinstance(e, dict) or e.get("source_hash") != traffic_source_hash():
            return None
        return int(e["bytes"])
    except Exception:
        return None


def vendor_baseline(a, k, B, C):
    """hipSPARSE SpMM (CSR_ALG3, row-major; the reference's cuSpmm protocol: 5 warm-up + 10 timed,
    flex.cu:5766-5789) on the same device-resident operands -- a reported side-by-side, not the metric."""
    import ctypes as ct

    import torch

    import flex_amd
    try:
        V = ct.CDLL(os.path.join(os.path.dirname(flex_amd.lib_path()), "libflex_vendor.so"))
        V.flex_vendor_spmm_create.argtypes = [ct.POINTER(ct.c_void_p), ct.c_int32, ct.c_int32, ct.c_int64, ct.c_void_p,
                                              ct.c_void_p, ct.c_void_p, ct.c_int, ct.c_void_p, ct.c_void_p]
        V.flex_vendor_spmm_run.argtypes = [ct.c_void_p, ct.c_void_p]
        V.flex_vendor_spmm_destroy.argtypes = [ct.c_void_p]
        rp = torch.from_numpy(a.rowPtr.astype(np.int32)).cuda()
        col = torch.from_numpy(a.col.astype(np.int32)).cuda()
        val = torch.from_numpy(a.vals).cuda()
        V.flex_vendor_spmm_create_alg.argtypes = V.flex_vendor_spmm_create.argtypes + [ct.c_int]
        s = torch.cuda.current_stream().cuda_stream

        def timed(alg):
            h = ct.c_void_p()
            if V.flex_vendor_spmm_create_alg(ct.byref(h), a.m, a.n, a.nnz, rp.data_ptr(), col.data_ptr(), val.data_ptr(), k,
                                             B.data_ptr(), C.data_ptr(), alg) != 0:
                return None
            for _ in range(5):
                V.flex_vendor_spmm_run(h, s)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                V.flex_vendor_spmm_run(h, s)
            e1.record()
            torch.cuda.synchronize()
            V.flex_vendor_spmm_destroy(h)
            return e0.elapsed_time(e1) / 10

        ms = timed(3)  # the reference's protocol
        if ms is None:
            return None
        others = {name: timed(alg) for alg, name in ((0, "default"), (1, "alg1"), (2, "alg2"))}
        best = min([("alg3", ms)] + [(n_, t) for n_, t in others.items() if t is not None], key=lambda x: x[1])
        return {"value": round(2.0 * a.nnz * k / (ms * 1e-3) / 1e9, 2), "unit": "GFLOPS", "ms_per_step": round(ms, 6),
                "algorithm": "hipsparseSpMM CSR_ALG3, row-major B/C, alpha=1, beta=0 (cuSpmm, flex.cu:5717-5804)",
                "other_algorithms_ms": {n_: (None if t is None else round(t, 6)) for n_, t in others.items()},
                "best_of_vendor": {"algorithm": best[0], "ms_per_step": round(best[1], 6),
                                   "value": round(2.0 * a.nnz * k / (best[1] * 1e-3) / 1e9, 2)}}
    except OSError:
        return None


def cpu_baseline(a, k, B):
    """The reference's CPU SpMM arithmetic (oracle port of aspt/sspmm_128.cu:1415-1422) timed on this host's cores (BASELINE.md 3:
    "all host cores of the GPU box").  The threaded leg is timed on every core this process may run on AND on 16 / 64 threads
    (`by_threads`); `value` is the fastest of them, `cores` the thread count that produced it; one thread beside it
    (`single_thread_value`).  Sample: the WHOLE workload -- every row of the same CSR and B -- whenever the fastest leg is predicted
    to fit ~10 s (the Amazon shape at k=128 does), else the first rows holding ~1e10 flops; the one-thread leg always runs on that
    bounded sample.  About 10-30 s of CPU work in all."""
    import oracle
    host_cores = os.cpu_count() or 1
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else host_cores
    budget = 1.0e10 / (2.0 * k)  # nonzeros of the bounded sample
    rp = a.rowPtr.astype(np.int64)
    rows = int(np.searchsorted(rp, budget, side="right")) - 1 if a.nnz > budget else a.m
    rows = max(rows, 1)
    nnz = int(rp[rows])
    rp_s = a.rowPtr[: rows + 1]
    Bh = B.cpu().numpy()

    def best_of(n, fn):
        best = 1e30
        for _ in range(n):
            t0 = time.perf_counter()
            fn()
            best = min(best, time.perf_counter() - t0)
        return best

    sample_run = lambda nt: (lambda: oracle.spmm(rp_s, a.col[:nnz], a.vals[:nnz], Bh, nthreads=nt))  # noqa: E731
    whole_run = lambda nt: (lambda: oracle.spmm(a.rowPtr, a.col, a.vals, Bh, nthreads=nt))  # noqa: E731
    # Thread counts: every core the process may use, and 16 / 64 beside it -- a row-parallel gather loop is bound by
    # the host's memory system, and on the 256-core GPU box all cores came out SLOWER than 16 threads (round 4: 12.9 vs 29.3 GFLOPS
    # on the Amazon sample).  `value` is the best of them and `cores` the thread count that produced it; `by_threads` has all.
    counts = sorted({c for c in (16, 64, cores) if c <= cores} or {cores})
    by_threads = {c: best_of(3 if c == counts[0] else 2, sample_run(c)) for c in counts}
    best_c = min(by_threads, key=by_threads.get)
    bestN = by_threads[best_c]
    best1 = best_of(1, sample_run(1))
    whole = rows < a.m and bestN * a.nnz / max(nnz, 1) <= 5.0  # the whole job, twice, within ~10 s
    if rows >= a.m:
        t_all, n_all, sample = bestN, nnz, f"the whole workload: all {a.m} rows ({a.nnz} nnz) of the same graph and B, best of 2-3"
    elif whole:
        t_all, n_all = best_of(2, whole_run(best_c)), a.nnz
        sample = f"the whole workload: all {a.m} rows ({a.nnz} nnz) of the same graph and B, best of 2 ({t_all:.2f} s)"
    else:
        t_all, n_all, sample = bestN, nnz, f"first {rows} rows ({nnz} nnz) of the same graph and B, best of 2-3"
    return {"value": round(2.0 * n_all * k / t_all / 1e9, 3), "unit": "GFLOPS", "cores": best_c, "host_cores": host_cores, "cores_available": cores,
            "kind": "port", "sample": sample,
            "by_threads": {str(c): round(2.0 * nnz * k / t / 1e9, 3) for c, t in by_threads.items()},
            "by_threads_sample": f"first {rows} rows ({nnz} nnz)",
            "single_thread_value": round(2.0 * nnz * k / best1 / 1e9, 3), "single_thread_sample": f"first {rows} rows ({nnz} nnz), one run"}


if __name__ == "__main__":
    main()
