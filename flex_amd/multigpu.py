"""Row sharding of one SpMM over the GPUs of a node (one process per GPU).

New work: the reference is single-GPU (flex.cu:4137).  Rows of A -- and of C -- are
independent units, B is replicated, so the data path has no collective: B is broadcast ONCE
(RCCL over xGMI via torch.distributed) before the timed region and every rank then runs
flex_spmm on its own contiguous, cost-balanced slice of the (re-ordered) rows.

Host logic only (numpy + torch.distributed); the per-rank compute is a flex_amd.Plan.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from . import binding as _b


@dataclass
class RowShard:
    a: "_b.HostCsr"         # the (re-ordered) matrix all ranks agree on
    vo_mp: np.ndarray | None  # vo_mp[new] = old when a was re-ordered, else None
    bounds: np.ndarray      # world+1 row boundaries in a's row order
    rank: int
    world: int

    @property
    def r0(self) -> int:
        return int(self.bounds[self.rank])

    @property
    def r1(self) -> int:
        return int(self.bounds[self.rank + 1])

    @property
    def nnz(self) -> int:
        return int(self.a.rowPtr[self.r1]) - int(self.a.rowPtr[self.r0])

    def plan(self, k: int, device: int, tuning: dict | None = None) -> "_b.Plan":
        """Plan for this rank's rows; columns are mapped back through vo_mp so the un-permuted B is used."""
        return _b.Plan(self.a, k, device=device, rows=(self.r0, self.r1), col_map=self.vo_mp, tuning=tuning)

    def local_csr(self):
        """(rowPtr, col, vals) of this rank's slice with columns in ORIGINAL numbering (what the plan computes)."""
        rp = self.a.rowPtr[self.r0:self.r1 + 1].astype(np.int64)
        cols = self.a.col[rp[0]:rp[-1]]
        if self.vo_mp is not None:
            cols = self.vo_mp[cols].astype(np.uint32)
        return (rp - rp[0]).astype(np.uint32), cols, self.a.vals[rp[0]:rp[-1]]

    def original_rows(self) -> np.ndarray:
        """Original row id of every local row (to scatter the C slice back)."""
        rows = np.arange(self.r0, self.r1)
        return rows if self.vo_mp is None else self.vo_mp[rows]


class OrderingFailed(RuntimeError):
    """shared_ordering: rank 0 failed to order (or to load / save the permutation cache); raised on EVERY rank of the job."""


ORDERINGS = {"rcm": _b.order_rcm, "cluster": _b.order_cluster, "deg": _b.order_deg, "gorder": _b.order_gorder, "dfs": _b.order_dfs}


def shared_ordering(a: "_b.HostCsr", order: str, timings: dict | None = None, cache: str | None = None) -> np.ndarray:
    """rank[old] = new for `a`, computed ONCE per job: with torch.distributed up, rank 0 runs the ordering (the community order
    of the Amazon shape is 2 s on 32 threads and 4 GB of scratch) and broadcasts the 4n bytes; the other ranks only receive.
    Without a process group (one process), computed here.  timings["order_s"] = seconds THIS rank spent ordering.
    cache: a permutation-cache file (flex_perm_save / _load, keyed by the matrix's fingerprint).  ONLY rank 0 touches it -- loads it if
    it matches, else orders and writes it -- so the ranks cannot disagree about whether there is a collective to join (a rank that
    found the file another rank had just written would skip the broadcast the others are waiting in).  timings["perm_cache"] =
    "loaded" / "written" (rank 0; the others report what rank 0 did).
    Failure: whatever rank 0 raises while ordering / loading / saving is caught there, a state word of -1 goes through the SAME
    broadcast, and every rank raises OrderingFailed -- nobody is left waiting in the collective."""
    import time
    t0 = time.perf_counter()
    spent = 0.0

    def order_or_load():
        nonlocal spent
        if cache:
            fp = _b.csr_fingerprint(a)
            try:
                return _b.perm_load(cache, a.m, fp), 1
            except _b.FlexError:
                pass
        r = ORDERINGS[order](a)
        spent = time.perf_counter() - t0
        if cache:
            _b.perm_save(cache, r, fp)
        return r, (2 if cache else 0)

    try:
        import torch
        import torch.distributed as dist
        group = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    except ImportError:
        group = False
    if not group:
        rank_arr, state = order_or_load()
    else:
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
        # Rank 0 can fail before it has anything to send (Gorder refuses isolated vertices, the 4 GB of scratch are not there, the
        # cache path is not writable, a cache file is damaged): it must still JOIN the collective, or the others wait in it until the
        # process-group timeout.  The last word of the one buffer is the state: -1 = rank 0 failed, and then every rank raises.
        failure = None
        if dist.get_rank() == 0:
            try:
                rank_arr, state = order_or_load()
                buf = torch.cat([torch.from_numpy(rank_arr.astype(np.int32)), torch.tensor([state], dtype=torch.int32)]).to(dev)
            except Exception as e:  # noqa: BLE001 -- whatever it was, the other ranks have to hear about it
                failure = e
                buf = torch.full((a.m + 1,), -1, dtype=torch.int32, device=dev)
        else:
            buf = torch.empty(a.m + 1, dtype=torch.int32, device=dev)
        dist.broadcast(buf, src=0)
        host = buf.cpu().numpy()
        rank_arr, state = host[: a.m].astype(np.uint32), int(host[a.m])
        if state < 0:
            if failure is not None:
                raise OrderingFailed(f"rank 0 could not produce the {order!r} ordering: {type(failure).__name__}: {failure}") from failure
            raise OrderingFailed(f"rank 0 could not produce the {order!r} ordering (its own message says why); rank {dist.get_rank()} stops with it")
    if timings is not None:
        timings["order_s"] = spent
        timings["perm_cache"] = {0: None, 1: "loaded", 2: "written"}[state]
    return rank_arr


def make_shard(a: "_b.HostCsr", k: int, rank: int, world: int, order: str = "cluster", rank_arr: np.ndarray | None = None,
               timings: dict | None = None) -> RowShard:
    """Every rank calls this with the same `a`: reorder first (so a shard's columns form a band /
    a set of communities), then cut contiguous row ranges of equal cost (flex_shard_rows).
    rank_arr: a precomputed rank[old] = new (a permutation cache, or shared_ordering); else the ordering is computed by rank 0
    and broadcast when a process group is up (shared_ordering)."""
    if timings is not None:
        timings["order_s"] = 0.0
    if order == "natural":
        vo, ap = None, a
    else:
        if rank_arr is None:
            rank_arr = shared_ordering(a, order, timings)
        vo, ap = _b.perm_csr(a, rank_arr)
    bounds = _b.shard_rows(ap, k, world)
    return RowShard(ap, vo, bounds, rank, world)


def broadcast_dense(B, src: int = 0, method: str = "broadcast"):
    """B (same-shaped tensor on every rank; contents valid on `src`) becomes valid everywhere.

    "broadcast" (default): one collective broadcast.  "scatter_allgather": the source scatters W equal row
    slices, one to each rank, and an all-gather completes every copy -- on xGMI, which is point-to-point,
    the source then feeds all of its links at once and every other link carries a slice, where a ring
    broadcast is paced by a single link (804 MB over 7 links vs 1).  Rows that do not divide
    by the world size go through a zero-padded staging tensor.  `nccl` (= RCCL) on GPUs, `gloo` in CPU tests.
    The second form is covered by 3-rank gloo tests only (no multi-GPU box this round), hence not the default."""
    import torch
    import torch.distributed as dist
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return B
    if method == "broadcast":
        dist.broadcast(B, src=src)
        return B
    if method != "scatter_allgather":
        raise ValueError(method)
    world, rank = dist.get_world_size(), dist.get_rank()
    n = B.shape[0]
    per = -(-n // world)
    row_elems = B[0].numel() if n else 0
    full = torch.empty((world * per,) + tuple(B.shape[1:]), dtype=B.dtype, device=B.device) if world * per != n else B
    if full is not B and rank == src:
        full[:n].copy_(B)
        full[n:].zero_()
    mine = torch.empty((per,) + tuple(B.shape[1:]), dtype=B.dtype, device=B.device)
    if per * row_elems:
        parts = list(full.split(per, dim=0)) if rank == src else None
        dist.scatter(mine, scatter_list=parts, src=src)
        dist.all_gather_into_tensor(full, mine)
    if full is not B:
        B.copy_(full[:n])
    return B
