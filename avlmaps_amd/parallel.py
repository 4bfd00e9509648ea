"""Multi-GPU sharding of the hot path: one process per GPU, torch.distributed (backend "nccl" == RCCL over xGMI).

Map creation shards FRAMES: rank r fuses the contiguous block shard_frames(F, r, ws) into its own
VoxelAccumulator, with no communication while frames stream in.  At the end ONE sparse merge runs:
  1. all_gather of the (small) per-rank voxel cell lists       -> every rank derives the same sorted union
  2. all_reduce(MIN) of the first-touch keys on the union       (8 B / voxel); sorting them gives every rank the
     reference's voxel-id order, so rows are scattered straight to their FINAL position
  3. ONE reduce(SUM) of the dense (M, D+4) float64 accumulators to the destination rank   <- the payload.
     The rank that owns a voxel's global first touch (its key == the MIN) subtracts the reference's first-touch term
     a1 (1 - a1) f1 (vlmap_builder.py:166-174 closed form) from its own contribution before the reduce, so no
     first-touch rows are exchanged and the reduced rows only need dividing by sum alpha.
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
