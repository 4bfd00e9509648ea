"""Multi-GPU sharding of the hot path: one process per GPU, torch.distributed (backend "nccl" == RCCL over xGMI).

Map creation shards FRAMES: rank r fuses the contiguous block shard_frames(F, r, ws) into its own
VoxelAccumulator, with no communication while frames stream in.  At the end ONE sparse merge runs:
  1. all_gather of the (small) per-rank voxel cell lists      -> every rank derives the same sorted union
  2. all_reduce(MIN) of the first-touch keys on the union      (8 B / voxel)
  3. ONE reduce(SUM) of the dense (M, D+4) float64 accumulators to the destination rank   <- the payload
  4. reduce(SUM) of the first-touch feature rows, non-zero only on the owning rank        (4 B*D / voxel)
followed by a first-touch-key sort on the destination so rows come out in the reference's voxel-id order.
Everything here is tensor plumbing (works on CPU tensors with gloo for the tests, on GPU tensors with RCCL);
the arithmetic that defines the map (accumulate / finalize) stays in the HIP library.

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
    if ws > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:   # AVLMAPS_DIST_BACKEND=gloo lets several ranks share one GPU (testing the choreography)
            backend = os.environ.get("AVLMAPS_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            local = local % torch.cuda.device_count()
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=ws)
    return rank, ws, local


def merge_raw(raw: Dict[str, "torch.Tensor"], dst: int = 0, group=None):
    """Merge per-rank raw accumulators (VoxelAccumulator.export_raw as torch tensors on one device).

    raw: cell (n,) int32 | first_key (n,) int64 | sum_feat (n,D) f64 | sum_w4 (n,4) f64 |
         first_feat (n,D) f32 | first_alpha (n,) f64
    Returns the merged dict ordered by first-touch key on rank `dst`, None on the other ranks.
    """
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        order = torch.argsort(raw["first_key"])
        return {k: v[order] for k, v in raw.items()}
    ws, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = raw["cell"].device
    D = raw["sum_feat"].shape[1]
    n = raw["cell"].shape[0]

    # 1. voxel cell lists -> identical sorted union on every rank
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(ws)]
    dist.all_gather(counts, torch.tensor([n], dtype=torch.int64, device=dev), group=group)
    counts = [int(c.item()) for c in counts]
    maxn = max(max(counts), 1)
    mine = torch.full((maxn,), -1, dtype=torch.int32, device=dev)
    mine[:n] = raw["cell"]
    gathered = [torch.empty_like(mine) for _ in range(ws)]
    dist.all_gather(gathered, mine, group=group)
    union = torch.unique(torch.cat([g[:c] for g, c in zip(gathered, counts)]))        # sorted ascending
    M = union.shape[0]
    idx = torch.searchsorted(union, raw["cell"]) if n else torch.zeros(0, dtype=torch.int64, device=dev)

    # 2. global first touch = smallest key over ranks
    gkey = torch.full((M,), I64_MAX, dtype=torch.int64, device=dev)
    gkey[idx] = raw["first_key"]
    dist.all_reduce(gkey, op=dist.ReduceOp.MIN, group=group)
    owner = raw["first_key"] == gkey[idx]

    # 3. the payload: ONE sum-reduce of the dense accumulators
    acc = torch.zeros((M, D + 4), dtype=torch.float64, device=dev)
    acc[idx, :D] = raw["sum_feat"]
    acc[idx, D:] = raw["sum_w4"]
    dist.reduce(acc, dst=dst, op=dist.ReduceOp.SUM, group=group)

    # 4. first-touch rows: non-zero on the owner only, so the sum is an exact copy
    ff = torch.zeros((M, D), dtype=torch.float32, device=dev)
    fa = torch.zeros((M,), dtype=torch.float64, device=dev)
    oi = idx[owner]
    ff[oi] = raw["first_feat"][owner]
    fa[oi] = raw["first_alpha"][owner]
    dist.reduce(ff, dst=dst, op=dist.ReduceOp.SUM, group=group)
    dist.reduce(fa, dst=dst, op=dist.ReduceOp.SUM, group=group)
    if rank != dst:
        return None
    order = torch.argsort(gkey)
    return dict(cell=union[order].to(torch.int32), first_key=gkey[order], sum_feat=acc[order, :D].contiguous(),
                sum_w4=acc[order, D:].contiguous(), first_feat=ff[order], first_alpha=fa[order])


def merge_raw_local(raws):
    """Same merge as merge_raw for several raw exports held by ONE process (e.g. two accumulators on one GPU):
    sums add, the smallest first-touch key owns the first-touch row, rows come out in key order."""
    import torch
    dev = raws[0]["cell"].device
    D = raws[0]["sum_feat"].shape[1]
    union = torch.unique(torch.cat([r["cell"] for r in raws]))
    M = union.shape[0]
    gkey = torch.full((M,), I64_MAX, dtype=torch.int64, device=dev)
    acc = torch.zeros((M, D + 4), dtype=torch.float64, device=dev)
    idxs = []
    for r in raws:
        idx = torch.searchsorted(union, r["cell"])
        idxs.append(idx)
        gkey[idx] = torch.minimum(gkey[idx], r["first_key"])
        acc[idx, :D] += r["sum_feat"]
        acc[idx, D:] += r["sum_w4"]
    ff = torch.zeros((M, D), dtype=torch.float32, device=dev)
    fa = torch.zeros((M,), dtype=torch.float64, device=dev)
    for r, idx in zip(raws, idxs):
        owner = r["first_key"] == gkey[idx]
        ff[idx[owner]] = r["first_feat"][owner]
        fa[idx[owner]] = r["first_alpha"][owner]
    order = torch.argsort(gkey)
    return dict(cell=union[order].to(torch.int32), first_key=gkey[order], sum_feat=acc[order, :D].contiguous(),
                sum_w4=acc[order, D:].contiguous(), first_feat=ff[order], first_alpha=fa[order])


def global_top1(best_val: "torch.Tensor", best_row: "torch.Tensor", row_offset: int, group=None):
    """Per-query best voxel over row shards: (Q,) local max values and local row indices -> global (value, row).
    Ties go to the lowest global row index (np.argmax semantics)."""
    import torch
    import torch.distributed as dist
    rows = best_row.to(torch.int64) + row_offset
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return best_val, rows
    ws = dist.get_world_size(group)
    vals = [torch.empty_like(best_val) for _ in range(ws)]
    idxs = [torch.empty_like(rows) for _ in range(ws)]
    dist.all_gather(vals, best_val, group=group)
    dist.all_gather(idxs, rows, group=group)
    V, I = torch.stack(vals), torch.stack(idxs)            # (ws, Q); ranks hold ascending row ranges
    vmax = V.max(dim=0).values
    cand = torch.where(V == vmax[None, :], I, torch.full_like(I, I64_MAX))
    return vmax, cand.min(dim=0).values
