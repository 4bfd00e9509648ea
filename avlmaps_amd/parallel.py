"""Multi-GPU sharding of the hot path: one process per GPU, torch.distributed (backend "nccl" == RCCL over xGMI).

Map creation shards FRAMES: rank r fuses the contiguous block shard_frames(F, r, ws) into its own
VoxelAccumulator, with no communication while frames stream in.  At the end ONE sparse merge runs:
  1. all_gather of the (small) per-rank voxel cell lists       -> every rank derives the same sorted union
  2. all_reduce(MIN) of the first-touch keys on the union       (8 B / voxel); sorting them gives every rank the
     reference's voxel-id order, so rows are scattered straight to their FINAL position
  3. the payload, in one of two forms:
     (a) ROW-SHARDED exchange (merge_accumulator_sharded, the default of VLMapBuilder): rank r owns the final rows
         shard_rows(M, r, ws); every rank sends each of its LOCAL voxel rows (D + 4 float64 + its final row index) to the
         owner of that row with one all_to_all_single (split sizes = reduce-scatter semantics).  Bytes moved = what a rank
         actually holds -- contiguous frame shards see mostly disjoint voxels -- not ws dense copies of the whole map, the
         owners finalise their rows in parallel, and the map is left row-sharded exactly as VLMap.shard_index_rows wants it
         for the index kernels.  A gather of the finished float32 rows to one rank is optional (file save).
     (b) ONE reduce(SUM) of a dense (M, D+4) float64 buffer to the destination rank (merge_accumulator; north_star's
         "single RCCL reduce"): simple, but every rank allocates and moves the whole map (9.3 GB at M = 2.25 M).
     Either way the rank that owns a voxel's global first touch (its key == the MIN) subtracts the reference's first-touch
     term a1 (1 - a1) f1 (vlmap_builder.py:166-174 closed form) from its own contribution first, so no first-touch rows
     are exchanged and the summed rows only need dividing by sum alpha.
  4. (optional, exact weight / grid_rgb) the sequential uint8 colour replay is a CHAIN over ranks: 24 B of state per
     voxel travel rank 0 -> 1 -> ... -> ws-1 (point-to-point), each rank continuing it with its own sample log.
The plan (steps 1-2) is tensor plumbing shared by two executions of step 3: merge_accumulator() drives the HIP
kernels of the library on the builder's own device arrays (the product path), merge_raw() does the same arithmetic on
exported torch tensors (CPU tensors with gloo in the tests; the cross-check of the device path on the GPU).

Landmark indexing shards VOXEL ROWS: each rank scores its own rows; only per-query (value, index) candidates
are exchanged (global_top1).

There is no reference counterpart: the upstream builder is single-process.
"""
from __future__ import annotations

import os
from typing import Dict, Optional, Tuple

import numpy as np

I64_MAX = np.iinfo(np.int64).max


def shard_frames(n_frames: int, rank: int, world_size: int) -> Tuple[int, int]:
    """contiguous block [start, stop) of frame indices for `rank` (keeps first-touch keys rank-monotone)"""
    per = (n_frames + world_size - 1) // world_size
    start = min(n_frames, rank * per)
    return start, min(n_frames, start + per)


def shard_rows(n_rows: int, rank: int, world_size: int) -> Tuple[int, int]:
    per = (n_rows + world_size - 1) // world_size
    start = min(n_rows, rank * per)
    return start, min(n_rows, start + per)


def init_distributed(backend: Optional[str] = None):
    """Initialise torch.distributed from the torchrun environment.  Returns (rank, world_size, local_rank)."""
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (ws > 1 or os.environ.get("AVLMAPS_FORCE_COLLECTIVES") == "1") and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:   # AVLMAPS_DIST_BACKEND=gloo lets several ranks share one GPU (testing the choreography)
            backend = os.environ.get("AVLMAPS_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            local = local % torch.cuda.device_count()
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=ws)
    return rank, ws, local


def _dist_on(group=None) -> bool:
    """collectives are used with more than one rank -- or with ONE rank when AVLMAPS_FORCE_COLLECTIVES=1 (a single MI355X
    box: RCCL refuses two ranks on one device, "Duplicate GPU detected", so this is how the real RCCL calls of the merge
    -- float64 sum-reduce of the payload, int64 MIN all-reduce, all_gather -- are exercised there)"""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or os.environ.get("AVLMAPS_FORCE_COLLECTIVES") == "1"


class _Coll:
    """the handful of collectives the merge uses; with the gloo backend (tests: several ranks on one GPU, or CPU tensors)
    device tensors are staged through the host, with nccl (= RCCL) they go as they are"""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.ws, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.stage = dist.get_backend(group) == "gloo"

    def _h(self, t):
        return t.cpu() if (self.stage and t.is_cuda) else t

    def all_gather(self, t):
        h = self._h(t)
        out = [h.new_empty(h.shape) for _ in range(self.ws)]
        self.dist.all_gather(out, h, group=self.group)
        return [o.to(t.device) for o in out]

    def all_reduce(self, t, op):
        h = self._h(t)
        self.dist.all_reduce(h, op=op, group=self.group)
        if h is not t:
            t.copy_(h)
        return t

    def reduce(self, t, dst, op):
        h = self._h(t)
        self.dist.reduce(h, dst=dst, op=op, group=self.group)
        if h is not t and self.rank == dst:
            t.copy_(h)
        return t

    def broadcast(self, t, src):
        h = self._h(t)
        self.dist.broadcast(h, src=src, group=self.group)
        if h is not t:
            t.copy_(h)
        return t

    def all_to_all(self, inp, in_splits, out_splits):
        """all_to_all_single along dim 0 with split sizes (rows); returns the received tensor (sum(out_splits), ...)"""
        h = self._h(inp).contiguous()
        out = h.new_empty((int(sum(out_splits)),) + tuple(h.shape[1:]))
        self.dist.all_to_all_single(out, h, [int(x) for x in out_splits], [int(x) for x in in_splits], group=self.group)
        return out.to(inp.device)

    def send(self, t, dst):
        self.dist.send(self._h(t).contiguous(), dst=dst, group=self.group)

    def recv(self, t, src):
        h = self._h(t)
        self.dist.recv(h, src=src, group=self.group)
        if h is not t:
            t.copy_(h)
        return t


class MergePlan:
    """what every rank knows after steps 1-2: M merged voxels in the reference's id order"""
    __slots__ = ("M", "cell", "key", "row_of_slot", "rank", "ws", "coll")

    def __init__(self, M, cell, key, row_of_slot, rank, ws, coll):
        self.M, self.cell, self.key, self.row_of_slot, self.rank, self.ws, self.coll = M, cell, key, row_of_slot, rank, ws, coll

    def grow_key(self, gs2: int) -> int:
        """first-touch key of the voxel with id gs2 - 1 (the reference re-allocates its arrays right after creating it,
        vlmap_builder.py:286-311), as an unsigned 64-bit value; all ones if the merged map is smaller"""
        return int(self.key[gs2 - 1].item()) if self.M >= gs2 else (1 << 64) - 1


def plan_merge(cell: "torch.Tensor", first_key: "torch.Tensor", group=None) -> MergePlan:
    """cell (n,) int32 linear cells of this rank's voxels, first_key (n,) int64 their first-touch keys (slot order).
    Collective over the group (a plain sort without torch.distributed)."""
    import torch
    dev = cell.device
    n = int(cell.shape[0])
    key = first_key.to(torch.int64)
    if not _dist_on(group):
        order = torch.argsort(key)
        pos = torch.empty_like(order)
        pos[order] = torch.arange(n, device=dev)
        return MergePlan(n, cell[order].to(torch.int32), key[order], pos, 0, 1, None)
    coll = _Coll(group)
    import torch.distributed as dist
    # 1. voxel cell lists -> identical sorted union on every rank
    counts = [int(c.item()) for c in coll.all_gather(torch.tensor([n], dtype=torch.int64, device=dev))]
    maxn = max(max(counts), 1)
    mine = torch.full((maxn,), -1, dtype=torch.int32, device=dev)
    mine[:n] = cell
    gathered = coll.all_gather(mine)
    union = torch.unique(torch.cat([g[:c] for g, c in zip(gathered, counts)]))        # sorted ascending
    M = int(union.shape[0])
    idx = torch.searchsorted(union, cell.to(union.dtype)) if n else torch.zeros(0, dtype=torch.int64, device=dev)
    # 2. global first touch = smallest key over ranks; its sort order is the reference's voxel-id order
    gkey = torch.full((M,), I64_MAX, dtype=torch.int64, device=dev)
    gkey[idx] = key
    coll.all_reduce(gkey, dist.ReduceOp.MIN)
    order = torch.argsort(gkey)
    pos = torch.empty_like(order)
    pos[order] = torch.arange(M, device=dev)
    return MergePlan(M, union[order].to(torch.int32), gkey[order], pos[idx], coll.rank, coll.ws, coll)


def merge_raw(raw: Dict[str, "torch.Tensor"], dst: int = 0, group=None):
    """Merge per-rank raw accumulators given as torch tensors (VoxelAccumulator.export_raw; CPU tensors + gloo in the tests).

    raw: cell (n,) int32 | first_key (n,) int64 | sum_feat (n,D) f64 | sum_w4 (n,4) f64 | first_feat (n,D) f32 | first_alpha (n,) f64
    Returns on rank `dst` dict(cell (M,) int32, first_key (M,) int64, acc (M, D+4) f64) in the reference's voxel-id order with
    the first-touch term already folded in: grid_feat = acc[:, :D] / acc[:, D] (ops.finalize_merged); None on the other ranks.
    """
    import torch
    plan = plan_merge(raw["cell"], raw["first_key"], group)
    D = raw["sum_feat"].shape[1]
    rows = plan.row_of_slot
    acc = torch.zeros((plan.M, D + 4), dtype=torch.float64, device=raw["cell"].device)
    a1 = raw["first_alpha"]
    own = raw["first_key"].to(torch.int64) == plan.key[rows]
    corr = torch.where(own, a1 * (1.0 - a1), torch.zeros_like(a1))
    acc[rows, :D] = raw["sum_feat"] - corr[:, None] * raw["first_feat"].double()
    acc[rows, D:] = raw["sum_w4"]
    if plan.coll is not None:
        import torch.distributed as dist
        plan.coll.reduce(acc, dst, dist.ReduceOp.SUM)
        if plan.rank != dst:
            return None
    return dict(cell=plan.cell, first_key=plan.key, acc=acc)


class ShardExchange:
    """who sends which local voxel row where in the row-sharded merge: final rows are dealt out in contiguous blocks
    shard_rows(M, r, ws); a rank's local slots, sorted by final row, are therefore already grouped by destination"""
    __slots__ = ("order", "send_pos", "rows_sorted", "send_counts", "recv_counts", "r0", "r1", "per")

    def __init__(self, plan: MergePlan):
        import torch
        rows = plan.row_of_slot.to(torch.int64)
        dev = rows.device
        ws, M = plan.ws, plan.M
        self.per = max(1, (M + ws - 1) // ws)
        self.r0, self.r1 = shard_rows(M, plan.rank, ws)
        self.order = torch.argsort(rows)                            # local slots in final-row order
        self.rows_sorted = rows[self.order].contiguous()
        self.send_pos = torch.empty_like(self.order)
        self.send_pos[self.order] = torch.arange(rows.shape[0], device=dev)
        dest = torch.clamp(self.rows_sorted // self.per, max=ws - 1)
        self.send_counts = torch.bincount(dest, minlength=ws)[:ws].to(torch.int64)
        if plan.coll is not None:
            allc = torch.stack(plan.coll.all_gather(self.send_counts.to(dev)))          # (ws, ws): [sender, receiver]
            self.recv_counts = allc[:, plan.rank].contiguous()
        else:
            self.recv_counts = self.send_counts.clone()
        self.send_counts, self.recv_counts = self.send_counts.cpu().tolist(), self.recv_counts.cpu().tolist()


def merge_raw_sharded(raw: Dict[str, "torch.Tensor"], group=None):
    """Row-sharded merge of per-rank raw accumulators given as torch tensors (the arithmetic of merge_accumulator_sharded on
    exported tensors: CPU + gloo in the tests, the cross-check of the device path on the GPU).  Returns on EVERY rank
    dict(M, rows=(r0, r1), cell (M,) int32, first_key (M,) int64, acc (r1 - r0, D + 4) float64 = this rank's block of final
    rows with the first-touch term folded in, bytes_sent)."""
    import torch
    plan = plan_merge(raw["cell"], raw["first_key"], group)
    D = raw["sum_feat"].shape[1]
    ex = ShardExchange(plan)
    a1 = raw["first_alpha"]
    own = raw["first_key"].to(torch.int64) == plan.key[plan.row_of_slot]
    corr = torch.where(own, a1 * (1.0 - a1), torch.zeros_like(a1))
    contrib = torch.cat([raw["sum_feat"] - corr[:, None] * raw["first_feat"].double(), raw["sum_w4"]], dim=1)[ex.order].contiguous()
    rows = ex.rows_sorted
    if plan.coll is not None:
        got_rows = plan.coll.all_to_all(rows, ex.send_counts, ex.recv_counts)
        got = plan.coll.all_to_all(contrib, ex.send_counts, ex.recv_counts)
    else:
        got_rows, got = rows, contrib
    acc = torch.zeros((ex.r1 - ex.r0, D + 4), dtype=torch.float64, device=raw["cell"].device)
    o = 0
    for c in ex.recv_counts:                        # peer by peer, in rank order: a reproducible float64 sum
        if c:
            acc.index_add_(0, got_rows[o:o + c] - ex.r0, got[o:o + c])
        o += c
    sent = sum(c for r, c in enumerate(ex.send_counts) if r != plan.rank)
    return dict(M=plan.M, rows=(ex.r0, ex.r1), cell=plan.cell, first_key=plan.key, acc=acc, bytes_sent=sent * ((D + 4) * 8 + 8))


def occupied_ids_from_cells(cell, n0: int, gs: int, vh: int):
    """the reference's dense (n0, gs, vh) voxel-id grid from the merged cell list (row i lives in linear cell cell[i]);
    every rank can derive it from the plan, nothing is exchanged"""
    import torch
    occ = torch.full((n0 * gs * vh,), -1, dtype=torch.int32, device=cell.device)
    occ[cell.to(torch.int64)] = torch.arange(cell.shape[0], dtype=torch.int32, device=cell.device)
    return occ.view(n0, gs, vh)


def merge_accumulator_sharded(acc, group=None, exact_rgb: bool = True, timings: Optional[dict] = None, gather_to: Optional[int] = None):
    """The product path of the multi-GPU build, row-sharded: merge the ranks' VoxelAccumulators with ONE sparse exchange and
    finalise every rank's block of final rows where it lives.

    Everything per-voxel runs in the HIP library on the builder's own device arrays (avl_builder_scatter_merge into the send
    buffer, avl_rows_add_f64 on the receiving side, avl_builder_replay_chain, avl_finalize_merged); torch.distributed carries
    the collectives (backend nccl == RCCL over xGMI).  Returns on every rank a dict of DEVICE tensors
        M, rows=(r0, r1), grid_feat (r1-r0, D) f32, grid_pos (r1-r0, 3) i32, weight (r1-r0,) f32, grid_rgb (r1-r0, 3) u8
    = this rank's block of the merged map in the reference's voxel-id order (the block VLMap.shard_index_rows scores), plus
    cell (M,) int32 of ALL rows (occupied_ids_from_cells gives the dense id grid anywhere).  gather_to = r: rank r also gets
    "full": the whole map (grid_feat, grid_pos, weight, grid_rgb, occupied_ids) as device tensors, e.g. to write the file.
    exact_rgb needs the replay log on every rank (the colour state is chained through the ranks in frame order)."""
    import time
    import torch
    from . import _lib
    from .device import torch_stream_ptr
    lib = _lib.load()
    st = torch_stream_ptr()
    dev = torch.device("cuda", torch.cuda.current_device())
    t0 = time.perf_counter()
    n = acc.num_voxels(st)
    cell = torch.empty((n,), dtype=torch.int32, device=dev)
    key = torch.empty((n,), dtype=torch.int64, device=dev)
    _lib.check(lib.avl_builder_export_raw(acc._h, n, cell.data_ptr(), key.data_ptr(), None, None, None, None, st), "avl_builder_export_raw")
    plan = plan_merge(cell, key, group)
    D, M = acc.D, plan.M
    ex = ShardExchange(plan)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    # every local voxel row, first-touch term folded in, straight into the send buffer in destination order
    W = D + 4
    send = torch.empty((max(n, 1), W), dtype=torch.float64, device=dev)
    key_by_pos = plan.key[ex.rows_sorted].contiguous() if n else torch.zeros((1,), dtype=torch.int64, device=dev)
    pos = ex.send_pos.contiguous()
    _lib.check(lib.avl_builder_scatter_merge(acc._h, n, pos.data_ptr(), key_by_pos.data_ptr(), send.data_ptr(), W, st),
               "avl_builder_scatter_merge")
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    if plan.coll is not None:
        got_rows = plan.coll.all_to_all(ex.rows_sorted, ex.send_counts, ex.recv_counts)
        got = plan.coll.all_to_all(send[:n], ex.send_counts, ex.recv_counts)
    else:
        got_rows, got = ex.rows_sorted, send[:n]
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    n_own = ex.r1 - ex.r0
    shard = torch.zeros((max(n_own, 1), W), dtype=torch.float64, device=dev)
    o = 0
    for c in ex.recv_counts:                        # peer by peer, in rank order: a reproducible float64 sum
        if c:
            _lib.check(lib.avl_rows_add_f64(c, W, got_rows[o:o + c].data_ptr(), ex.r0, n_own, got[o:o + c].data_ptr(), W, shard.data_ptr(), W, st),
                       "avl_rows_add_f64")
        o += c
    del got, send
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    # exact sequential weight / colour: the replay state is chained through the ranks in frame order (24 B per voxel and hop),
    # the last rank broadcasts the final state and every owner applies its block
    state = None
    have_log = 1 if (exact_rgb and acc.has_replay_log()) else 0
    chain_bytes = 0
    if plan.coll is not None:
        import torch.distributed as dist
        flag = torch.tensor([have_log], dtype=torch.int64, device=dev)
        plan.coll.all_reduce(flag, dist.ReduceOp.MIN)
        have_log = int(flag.item())
    if have_log and M > 0:
        state = torch.zeros((M, 3), dtype=torch.int64, device=dev)       # 24 B per voxel: {f64 w, f32 rgb[3], u32 started}
        if plan.coll is not None and plan.rank > 0:
            plan.coll.recv(state, plan.rank - 1)
        gk = plan.grow_key(acc.n_rows * acc.gs)
        rows64 = plan.row_of_slot.contiguous()
        _lib.check(lib.avl_builder_replay_chain(acc._h, n, rows64.data_ptr(), gk, state.data_ptr(), st), "avl_builder_replay_chain")
        if plan.coll is not None:
            torch.cuda.synchronize()
            if plan.rank < plan.ws - 1:
                plan.coll.send(state, plan.rank + 1)
                chain_bytes += M * 24
            plan.coll.broadcast(state, plan.ws - 1)
            if plan.rank == plan.ws - 1:
                chain_bytes += M * 24
    torch.cuda.synchronize()
    t5 = time.perf_counter()
    out = dict(M=M, rows=(ex.r0, ex.r1), cell=plan.cell,
               grid_feat=torch.empty((n_own, D), dtype=torch.float32, device=dev),
               grid_pos=torch.empty((n_own, 3), dtype=torch.int32, device=dev),
               weight=torch.empty((n_own,), dtype=torch.float32, device=dev),
               grid_rgb=torch.empty((n_own, 3), dtype=torch.uint8, device=dev))
    if n_own:
        own_cell = plan.cell[ex.r0:ex.r1].contiguous()
        _lib.check(lib.avl_finalize_merged(n_own, ex.r0, D, acc.gs, acc.vh, own_cell.data_ptr(), shard.data_ptr(), W, out["grid_feat"].data_ptr(),
                                           out["grid_pos"].data_ptr(), out["weight"].data_ptr(), out["grid_rgb"].data_ptr(), None, st),
                   "avl_finalize_merged")
        if state is not None:
            own_state = state[ex.r0:ex.r1].contiguous()
            _lib.check(lib.avl_replay_state_apply(n_own, own_state.data_ptr(), out["weight"].data_ptr(), out["grid_rgb"].data_ptr(), st),
                       "avl_replay_state_apply")
    del shard
    torch.cuda.synchronize()
    t6 = time.perf_counter()
    gather_bytes = 0
    if gather_to is not None:
        full = gather_row_shards(out, gather_to, plan.coll, plan.rank, plan.ws)
        if plan.rank != gather_to:
            gather_bytes = n_own * (D * 4 + 12 + 4 + 3)
        if full is not None:
            full["occupied_ids"] = occupied_ids_from_cells(plan.cell, acc.n_rows, acc.gs, acc.vh)
            out["full"] = full
    torch.cuda.synchronize()
    t7 = time.perf_counter()
    if timings is not None:
        sent_rows = sum(c for r, c in enumerate(ex.send_counts) if r != plan.rank)
        plan_bytes = (4 * n + 8 * M + 8 * plan.ws) if plan.coll is not None else 0
        timings.update(mode="row-sharded all_to_all", plan_s=t1 - t0, scatter_s=t2 - t1, exchange_s=t3 - t2, accumulate_s=t4 - t3,
                       replay_chain_s=t5 - t4, finalize_s=t6 - t5, gather_s=t7 - t6, merged_voxels=M, local_voxels=n, own_rows=n_own,
                       rows_sent=sent_rows, payload_bytes_sent=sent_rows * (W * 8 + 8), plan_bytes_sent=plan_bytes,
                       chain_bytes_sent=chain_bytes, gather_bytes_sent=gather_bytes,
                       bytes_sent_per_rank=sent_rows * (W * 8 + 8) + plan_bytes + chain_bytes + gather_bytes,
                       local_row_bytes=n * W * 8, dense_reduce_payload_bytes=M * W * 8, exact_rgb=bool(have_log),
                       world_size=plan.ws, backend=(plan.coll.dist.get_backend(plan.coll.group) if plan.coll is not None else "none"))
    return out


def gather_row_shards(shard: dict, dst: int, coll, rank: int, ws: int):
    """the ranks' blocks of finished rows (grid_feat / grid_pos / weight / grid_rgb device tensors, rank order = row order) ->
    the whole arrays on rank `dst`; None elsewhere.  One all_to_all_single per array in which only `dst` receives."""
    import torch
    names = ("grid_feat", "grid_pos", "weight", "grid_rgb")
    if coll is None:
        return {k: shard[k] for k in names}
    n_own = int(shard["grid_feat"].shape[0])
    counts = [int(c.item()) for c in coll.all_gather(torch.tensor([n_own], dtype=torch.int64, device=shard["grid_feat"].device))]
    full = {}
    for k in names:
        ins = [n_own if r == dst else 0 for r in range(ws)]
        outs = counts if rank == dst else [0] * ws
        got = coll.all_to_all(shard[k], ins, outs)
        if rank == dst:
            full[k] = got
    return full if rank == dst else None


def merge_raw_local(raws):
    """Same merge for several raw exports held by ONE process (e.g. two accumulators on one GPU): sums add, the smallest
    first-touch key owns the first-touch term, rows come out in key order."""
    import torch
    dev = raws[0]["cell"].device
    D = raws[0]["sum_feat"].shape[1]
    union = torch.unique(torch.cat([r["cell"] for r in raws]))
    M = union.shape[0]
    gkey = torch.full((M,), I64_MAX, dtype=torch.int64, device=dev)
    idxs = []
    for r in raws:
        idx = torch.searchsorted(union, r["cell"])
        idxs.append(idx)
        gkey[idx] = torch.minimum(gkey[idx], r["first_key"].to(torch.int64))
    acc = torch.zeros((M, D + 4), dtype=torch.float64, device=dev)
    for r, idx in zip(raws, idxs):
        own = r["first_key"].to(torch.int64) == gkey[idx]
        a1 = r["first_alpha"]
        corr = torch.where(own, a1 * (1.0 - a1), torch.zeros_like(a1))
        acc[idx, :D] += r["sum_feat"] - corr[:, None] * r["first_feat"].double()
        acc[idx, D:] += r["sum_w4"]
    order = torch.argsort(gkey)
    return dict(cell=union[order].to(torch.int32), first_key=gkey[order], acc=acc[order].contiguous())


def merge_accumulator(acc, dst: int = 0, group=None, exact_rgb: bool = True, timings: Optional[dict] = None):
    """The product path of the multi-GPU build: merge the ranks' VoxelAccumulators and finalise on rank `dst`.

    Everything per-voxel runs in the HIP library on the builder's own device arrays (avl_builder_scatter_merge,
    avl_builder_replay_chain, avl_finalize_merged); torch.distributed carries the collectives.  Returns on `dst` a dict of
    DEVICE tensors grid_feat (M,D) f32, grid_pos (M,3) i32, weight (M,) f32, grid_rgb (M,3) u8, occupied_ids (n0,gs,vh) i32
    in the reference's voxel-id order; None on the other ranks.  exact_rgb needs the replay log on every rank.
    """
    import time
    import torch
    from . import _lib
    from .device import torch_stream_ptr
    lib = _lib.load()
    st = torch_stream_ptr()
    dev = torch.device("cuda", torch.cuda.current_device())
    t0 = time.perf_counter()
    n = acc.num_voxels(st)
    cell = torch.empty((n,), dtype=torch.int32, device=dev)
    key = torch.empty((n,), dtype=torch.int64, device=dev)
    _lib.check(lib.avl_builder_export_raw(acc._h, n, cell.data_ptr(), key.data_ptr(), None, None, None, None, st), "avl_builder_export_raw")
    plan = plan_merge(cell, key, group)
    D, M = acc.D, plan.M
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    buf = torch.zeros((max(M, 1), D + 4), dtype=torch.float64, device=dev)
    rows = plan.row_of_slot.contiguous()
    _lib.check(lib.avl_builder_scatter_merge(acc._h, n, rows.data_ptr(), plan.key.data_ptr(), buf.data_ptr(), D + 4, st),
               "avl_builder_scatter_merge")
    if plan.coll is not None:
        import torch.distributed as dist
        plan.coll.reduce(buf, dst, dist.ReduceOp.SUM)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    # exact sequential weight / colour: chain the replay state through the ranks in frame order
    state = None
    have_log = 1 if (exact_rgb and acc.has_replay_log()) else 0
    if plan.coll is not None:
        import torch.distributed as dist
        flag = torch.tensor([have_log], dtype=torch.int64, device=dev)
        plan.coll.all_reduce(flag, dist.ReduceOp.MIN)
        have_log = int(flag.item())
    if have_log and M > 0:
        state = torch.zeros((M, 3), dtype=torch.int64, device=dev)       # 24 B per voxel: {f64 w, f32 rgb[3], u32 started}
        if plan.coll is not None and plan.rank > 0:
            plan.coll.recv(state, plan.rank - 1)
        gk = plan.grow_key(acc.n_rows * acc.gs)
        _lib.check(lib.avl_builder_replay_chain(acc._h, n, rows.data_ptr(), gk, state.data_ptr(), st), "avl_builder_replay_chain")
        if plan.coll is not None:
            torch.cuda.synchronize()
            if plan.rank < plan.ws - 1:
                plan.coll.send(state, plan.rank + 1)
            if plan.ws - 1 != dst:                                       # the last rank of the chain holds the final state
                if plan.rank == plan.ws - 1:
                    plan.coll.send(state, dst)
                elif plan.rank == dst:
                    plan.coll.recv(state, plan.ws - 1)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    out = None
    if plan.rank == dst:
        out = dict(grid_feat=torch.empty((M, D), dtype=torch.float32, device=dev),
                   grid_pos=torch.empty((M, 3), dtype=torch.int32, device=dev),
                   weight=torch.empty((M,), dtype=torch.float32, device=dev),
                   grid_rgb=torch.empty((M, 3), dtype=torch.uint8, device=dev),
                   occupied_ids=torch.full((acc.n_rows, acc.gs, acc.vh), -1, dtype=torch.int32, device=dev))
        _lib.check(lib.avl_finalize_merged(M, 0, D, acc.gs, acc.vh, plan.cell.data_ptr(), buf.data_ptr(), D + 4, out["grid_feat"].data_ptr(),
                                           out["grid_pos"].data_ptr(), out["weight"].data_ptr(), out["grid_rgb"].data_ptr(),
                                           out["occupied_ids"].data_ptr(), st), "avl_finalize_merged")
        if state is not None:
            _lib.check(lib.avl_replay_state_apply(M, state.data_ptr(), out["weight"].data_ptr(), out["grid_rgb"].data_ptr(), st),
                       "avl_replay_state_apply")
    torch.cuda.synchronize()
    if timings is not None:
        timings.update(plan_s=t1 - t0, scatter_reduce_s=t2 - t1, replay_chain_s=t3 - t2, finalize_s=time.perf_counter() - t3,
                       merged_voxels=M, local_voxels=n, payload_bytes=M * (D + 4) * 8, exact_rgb=bool(have_log))
    return out


def rank_world(group=None) -> Tuple[int, int]:
    """(rank, world_size) of the process group, (0, 1) without torch.distributed"""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(group), dist.get_world_size(group)
    except Exception:
        pass
    return 0, 1


def gather_rows(local: np.ndarray, n_rows: int, group=None) -> np.ndarray:
    """Row-sharded indexing: every rank holds the rows shard_rows(n_rows, rank, ws) of a per-voxel result (argmax (n,), scores
    (n, Q), ...); returns the full (n_rows, ...) host array on every rank.  One all_gather of equally padded shards -- the only
    exchange of the sharded index path, and it carries results (4 B per voxel and query), never features."""
    import torch
    rank, ws = rank_world(group)
    if ws == 1:
        return local
    coll = _Coll(group)
    per = (n_rows + ws - 1) // ws
    pad = np.zeros((per,) + local.shape[1:], dtype=local.dtype)
    pad[: local.shape[0]] = local
    dev = "cuda" if (coll.dist.get_backend(group) == "nccl") else "cpu"
    parts = coll.all_gather(torch.from_numpy(pad).to(dev))
    out = np.concatenate([p.cpu().numpy() for p in parts], axis=0)
    return out[:n_rows]


def global_top1(best_val: "torch.Tensor", best_row: "torch.Tensor", row_offset: int, group=None):
    """Per-query best voxel over row shards: (Q,) local max values and local row indices -> global (value, row).
    Ties go to the lowest global row index (np.argmax semantics)."""
    import torch
    import torch.distributed as dist
    rows = best_row.to(torch.int64) + row_offset
    if not _dist_on(group):
        return best_val, rows
    coll = _Coll(group)
    V, I = torch.stack(coll.all_gather(best_val)), torch.stack(coll.all_gather(rows))   # (ws, Q); ranks hold ascending rows
    vmax = V.max(dim=0).values
    cand = torch.where(V == vmax[None, :], I, torch.full_like(I, I64_MAX))
    return vmax, cand.min(dim=0).values
