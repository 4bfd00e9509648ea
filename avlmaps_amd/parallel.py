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
     Either way every rank folds its LOCAL first-touch sample (which the accumulators keep out of sum_feat) into its own
     contribution first -- with the reference's weight a1^2 on the rank that owns the voxel's global first touch (its key ==
     the MIN; vlmap_builder.py:166-174 closed form), with the plain weight a1 elsewhere -- so no first-touch rows
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
        if backend == "nccl" and torch.cuda.is_available():
            # RCCL builds its communicator lazily inside the FIRST collective (0.9-3.3 s on an MI355X box, profiles/r06_*): pay it
            # here, at process start-up, not inside the first merge of a build
            warm = torch.zeros(1, device="cuda")
            dist.all_reduce(warm)
            torch.cuda.synchronize()
    return rank, ws, local


def _dist_on(group=None) -> bool:
    """collectives are used with more than one rank -- or with ONE rank when AVLMAPS_FORCE_COLLECTIVES=1 (a single MI355X
    box: RCCL refuses two ranks on one device, "Duplicate GPU detected", so this is how the real RCCL calls of the merge
    -- float64 sum-reduce of the payload, int64 MIN all-reduce, all_gather -- are exercised there)"""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or os.environ.get("AVLMAPS_FORCE_COLLECTIVES") == "1"


class _SharedGpuLock:
    """Rehearsals that put several ranks on ONE GPU (AVLMAPS_SHARED_GPU_LOCK=<file>): a rank holds this cross-process lock
    whenever it computes inside a merge and drops it while it sits in a collective, so the ranks take turns on the device.

    Why: torch's large sorts / scans / selects are rocPRIM kernels whose workgroups spin on each other's look-back flags.  Eight
    processes launching them on one GPU at the same moment fill the CUs with spinning workgroups of different processes while
    the workgroups they wait for are not resident; progress then comes only from the scheduler's time slices -- 4 to 160 s
    inside a single torch.argsort of 1.25 M elements, on every rank at once (profiles/r04_shared_gpu_stall.txt), against
    < 1 ms alone.  One process per GPU -- the real launch -- cannot see this.  With the lock a rank's compute time is also what
    it would be on a GPU of its own; the time spent waiting for the lock is reported separately (shared_gpu_wait_s)."""
    _inst = None

    def __init__(self):
        self.path = os.environ.get("AVLMAPS_SHARED_GPU_LOCK")
        self.fd, self.held, self.wait_s = None, False, 0.0
        if self.path:
            self.fd = os.open(self.path, os.O_CREAT | os.O_RDWR, 0o666)

    @classmethod
    def get(cls):
        if cls._inst is None:
            cls._inst = cls()
        return cls._inst

    def acquire(self):
        if self.fd is None or self.held:
            return
        import fcntl
        import time
        t0 = time.perf_counter()
        fcntl.flock(self.fd, fcntl.LOCK_EX)
        self.wait_s += time.perf_counter() - t0
        self.held = True

    def release(self, device_sync=True):
        if self.fd is None or not self.held:
            return
        import fcntl
        if device_sync:
            try:
                import torch
                if torch.cuda.is_available():
                    torch.cuda.synchronize()
            except Exception:
                pass
        fcntl.flock(self.fd, fcntl.LOCK_UN)
        self.held = False


def torch_cat_rows(*parts):
    import torch
    return torch.cat(list(parts), dim=0)


class _Coll:
    """the handful of collectives the merge uses; with the gloo backend (tests: several ranks on one GPU, or CPU tensors)
    device tensors are staged through the host, with nccl (= RCCL) they go as they are.  Every call is bracketed by device
    synchronisations and adds its wall time to `comm_s` and its payload to `bytes_out`: merge_breakdown separates the time a
    rank spends INSIDE collectives (transfer + waiting for the peers) from its own compute."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.ws, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.stage = dist.get_backend(group) == "gloo"
        self.comm_s, self.bytes_out, self.calls = 0.0, 0, 0
        self.gpu_lock = _SharedGpuLock.get()
        self._relock = False

    def _h(self, t):
        return t.cpu() if (self.stage and t.is_cuda) else t

    def _tic(self, t):
        import time
        if t.is_cuda:
            import torch
            torch.cuda.synchronize()
        self._lw_tic = self.gpu_lock.wait_s
        return time.perf_counter()

    def _net(self):
        """context of the pure network part of a collective: a rank waiting for its peers does not keep the shared GPU of a
        one-GPU rehearsal (_SharedGpuLock); the device staging copies around it (gloo only) stay inside the lock, so that no rank
        ever computes while another one's copies are on the device"""
        coll = self

        class _N:
            def __enter__(self_n):
                coll._relock = coll.gpu_lock.held
                coll.gpu_lock.release(device_sync=True)

            def __exit__(self_n, *a):
                if coll._relock:
                    coll.gpu_lock.acquire()
        return _N()

    def _toc(self, t, t0, nbytes):
        import time
        if t.is_cuda:
            import torch
            torch.cuda.synchronize()
        self.comm_s += time.perf_counter() - t0 - (self.gpu_lock.wait_s - self._lw_tic)     # (waiting for the shared GPU is booked separately)
        self.bytes_out += int(nbytes)
        self.calls += 1

    def all_gather(self, t):
        t0 = self._tic(t)
        h = self._h(t)
        out = [h.new_empty(h.shape) for _ in range(self.ws)]
        with self._net():
            self.dist.all_gather(out, h, group=self.group)
        res = [o.to(t.device) for o in out]
        del out, h                              # host staging (gloo rehearsals) is released inside the timed bracket
        self._toc(t, t0, t.numel() * t.element_size() * (self.ws - 1))
        return res

    def all_gather_into(self, out, chunk):
        """every rank's equal-sized `chunk` into the flat `out` (ws x chunk), in place: `chunk` is this rank's slice of `out`"""
        t0 = self._tic(out)
        if self.stage or not out.is_cuda:
            h = self._h(chunk).contiguous()
            parts = [h.new_empty(h.shape) for _ in range(self.ws)]
            with self._net():
                self.dist.all_gather(parts, h, group=self.group)
            n = chunk.numel()
            for r, part in enumerate(parts):
                if r != self.rank:
                    out[r * n:(r + 1) * n].copy_(part)
            del parts, h
        else:
            with self._net():
                self.dist.all_gather_into_tensor(out, chunk, group=self.group)
        self._toc(out, t0, chunk.numel() * chunk.element_size() * (self.ws - 1))
        return out

    def all_reduce(self, t, op):
        t0 = self._tic(t)
        h = self._h(t)
        with self._net():
            self.dist.all_reduce(h, op=op, group=self.group)
        if h is not t:
            t.copy_(h)
        del h
        self._toc(t, t0, t.numel() * t.element_size())
        return t

    def reduce(self, t, dst, op):
        t0 = self._tic(t)
        h = self._h(t)
        with self._net():
            self.dist.reduce(h, dst=dst, op=op, group=self.group)
        if h is not t and self.rank == dst:
            t.copy_(h)
        del h
        self._toc(t, t0, 0 if self.rank == dst else t.numel() * t.element_size())
        return t

    def broadcast(self, t, src):
        t0 = self._tic(t)
        h = self._h(t)
        with self._net():
            self.dist.broadcast(h, src=src, group=self.group)
        if h is not t:
            t.copy_(h)
        del h
        self._toc(t, t0, t.numel() * t.element_size() if self.rank == src else 0)
        return t

    def all_to_all(self, inp, in_splits, out_splits):
        """all_to_all_single along dim 0 with split sizes (rows); returns the received tensor (sum(out_splits), ...)"""
        t0 = self._tic(inp)
        h = self._h(inp).contiguous()
        in_splits, out_splits = [int(x) for x in in_splits], [int(x) for x in out_splits]
        # a rank that sends or receives NOTHING in this exchange (the replay hops: only one rank receives) would hand the backend an
        # empty tensor; one dummy row travels from the rank to itself instead -- a self split is nobody else's business, so every
        # rank decides this alone
        pad = h.shape[0] == 0 or sum(out_splits) == 0
        if pad:
            o = sum(in_splits[:self.rank])
            h = torch_cat_rows(h[:o], h.new_zeros((1,) + tuple(h.shape[1:])), h[o:])
            in_splits[self.rank] += 1
            out_splits[self.rank] += 1
        out = h.new_empty((int(sum(out_splits)),) + tuple(h.shape[1:]))
        with self._net():
            self.dist.all_to_all_single(out, h, out_splits, in_splits, group=self.group)
        if pad:
            o = sum(out_splits[:self.rank])
            out = torch_cat_rows(out[:o], out[o + 1:])
            in_splits[self.rank] -= 1
            out_splits[self.rank] -= 1
        res = out.to(inp.device)
        del out, h                              # gloo rehearsals: GBs of pageable staging are unmapped here, not on the rank's compute clock
        row_b = inp.element_size() * (int(inp.numel() // inp.shape[0]) if inp.shape[0] else 0)
        self._toc(inp, t0, row_b * sum(int(c) for r, c in enumerate(in_splits) if r != self.rank))
        return res

    def all_to_all_start(self, inp, in_splits, out_splits, overlap=True):
        """all_to_all that may still be in flight when this returns: with nccl (= RCCL) and device tensors the exchange is queued on the
        backend's stream behind the work already on the current one and the caller goes on issuing kernels; all_to_all_finish makes the
        current stream wait for it and hands out the received tensor.  Staged (gloo) and empty exchanges complete here.  Only the host
        time of the two calls is booked to comm_s for an exchange in flight -- the transfer itself hides under the caller's kernels or
        shows up where the caller next drains the device."""
        in_splits, out_splits = [int(x) for x in in_splits], [int(x) for x in out_splits]
        if (not overlap or self.stage or not inp.is_cuda or os.environ.get("AVLMAPS_MERGE_ASYNC", "1") == "0"
                or inp.shape[0] == 0 or sum(out_splits) == 0):
            return ("done", self.all_to_all(inp, in_splits, out_splits))
        import time
        t0 = time.perf_counter()
        out = inp.new_empty((int(sum(out_splits)),) + tuple(inp.shape[1:]))
        work = self.dist.all_to_all_single(out, inp, out_splits, in_splits, group=self.group, async_op=True)
        self.comm_s += time.perf_counter() - t0
        row_b = inp.element_size() * int(inp.numel() // inp.shape[0])
        self.bytes_out += row_b * sum(c for r, c in enumerate(in_splits) if r != self.rank)
        self.calls += 1
        return ("flying", work, out, inp)

    def all_to_all_finish(self, handle):
        if handle[0] == "done":
            return handle[1]
        import time
        t0 = time.perf_counter()
        handle[1].wait()                            # the current stream waits for the backend's; the host does not
        self.comm_s += time.perf_counter() - t0
        return handle[2]

    def send(self, t, dst):
        t0 = self._tic(t)
        h = self._h(t).contiguous()
        with self._net():
            self.dist.send(h, dst=dst, group=self.group)
        del h
        self._toc(t, t0, t.numel() * t.element_size())

    def recv(self, t, src):
        t0 = self._tic(t)
        h = self._h(t)
        with self._net():
            self.dist.recv(h, src=src, group=self.group)
        if h is not t:
            t.copy_(h)
        self._toc(t, t0, 0)
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
    wf = torch.where(own, a1 * a1, a1)          # sum_feat leaves the local first touch out: a1^2 f1 for the global owner, a1 f1 otherwise
    acc[rows, :D] = wf[:, None] * raw["first_feat"].double() + raw["sum_feat"]
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
        self.order = _argsort_bits(rows, max(1, int(M).bit_length()))   # local slots in final-row order (rows are distinct)
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


def _merge_raw_sharded_general(raw: Dict[str, "torch.Tensor"], group=None):
    """(general plan: every rank sorts all M voxels) Row-sharded merge of per-rank raw accumulators given as torch tensors (the arithmetic of merge_accumulator_sharded on
    exported tensors: CPU + gloo in the tests, the cross-check of the device path on the GPU).  Returns on EVERY rank
    dict(M, rows=(r0, r1), cell (M,) int32, first_key (M,) int64, acc (r1 - r0, D + 4) float64 = this rank's block of final
    rows with the first-touch term folded in, bytes_sent)."""
    import torch
    plan = plan_merge(raw["cell"], raw["first_key"], group)
    D = raw["sum_feat"].shape[1]
    ex = ShardExchange(plan)
    a1 = raw["first_alpha"]
    own = raw["first_key"].to(torch.int64) == plan.key[plan.row_of_slot]
    wf = torch.where(own, a1 * a1, a1)
    contrib = torch.cat([wf[:, None] * raw["first_feat"].double() + raw["sum_feat"], raw["sum_w4"]], dim=1)[ex.order].contiguous()
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


def _merge_accumulator_sharded_general(acc, group=None, exact_rgb: bool = True, timings: Optional[dict] = None, gather_to: Optional[int] = None):
    """(general plan, round 3: every rank sorts all M voxels and the colour replay is a dense chain over all ranks -- kept as the
    fall-back for first-touch keys that are not rank-monotone and as the A/B baseline, AVLMAPS_MERGE_PLAN=general)
    The product path of the multi-GPU build, row-sharded: merge the ranks' VoxelAccumulators with ONE sparse exchange and
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
        timings.update(mode="row-sharded all_to_all", plan="general (every rank sorts all M voxels; dense replay chain)", plan_s=t1 - t0, scatter_s=t2 - t1, exchange_s=t3 - t2, accumulate_s=t4 - t3,
                       replay_chain_s=t5 - t4, finalize_s=t6 - t5, gather_s=t7 - t6, merged_voxels=M, local_voxels=n, own_rows=n_own,
                       rows_sent=sent_rows, payload_bytes_sent=sent_rows * (W * 8 + 8), plan_bytes_sent=plan_bytes,
                       chain_bytes_sent=chain_bytes, gather_bytes_sent=gather_bytes,
                       bytes_sent_per_rank=sent_rows * (W * 8 + 8) + plan_bytes + chain_bytes + gather_bytes,
                       local_row_bytes=n * W * 8, dense_reduce_payload_bytes=M * W * 8, exact_rgb=bool(have_log),
                       world_size=plan.ws, backend=(plan.coll.dist.get_backend(plan.coll.group) if plan.coll is not None else "none"))
    return out


# --------------------------------------------------------------------------------------------------------------------------------
# Round 4: the DIRECTORY plan.  Nothing a rank computes or holds is O(M): its critical path shrinks with the rank count.
# --------------------------------------------------------------------------------------------------------------------------------
U64_ALL_ONES = (1 << 64) - 1


def _mask_idx(mask, size):
    """Indices of the set elements of a boolean vector whose COUNT the host already knows (it rode on one of the plan's tiny
    all_gathers): torch.nonzero / boolean-mask indexing would read that count back from the device -- a host synchronisation
    per mask, a dozen per merge plan, each ~80-150 us (profiles/r05_merge_plan_ops.txt)"""
    import torch
    return torch.nonzero_static(mask, size=int(size)).reshape(-1)


def _argsort_bits(t, bits: int):
    """torch.argsort(t, stable=True) of non-negative integers below 2^bits.  On the GPU: the library's radix sort over just those
    bits (avl_argsort_bits) -- torch sorts every int64 vector with all 64 bits, eight passes where the merge plan's 3-bit
    destination ranks need one and its row numbers three."""
    import torch
    n = int(t.numel())
    if not t.is_cuda or n < 8192 or t.dtype not in (torch.int64, torch.int32):
        return torch.argsort(t, stable=True)
    from . import _lib
    from .device import torch_stream_ptr
    import ctypes as C
    lib = _lib.load()
    t = t.contiguous()
    perm = torch.empty(n, dtype=torch.int64, device=t.device)
    nb = C.c_size_t()
    bits = int(min(max(bits, 1), 8 * t.element_size() - 1))
    _lib.check(lib.avl_argsort_bits_work_bytes(n, t.element_size(), bits, C.byref(nb)), "avl_argsort_bits_work_bytes")
    work = torch.empty(int(nb.value), dtype=torch.uint8, device=t.device)      # torch's caching allocator: no hipMalloc per sort
    _lib.check(lib.avl_argsort_bits(n, t.data_ptr(), t.element_size(), bits, perm.data_ptr(), work.data_ptr(), int(nb.value),
                                    torch_stream_ptr()), "avl_argsort_bits")
    return perm


def _merge_hip(t):
    """the fused kernels of the plan's local steps (csrc/avl_merge.hip) for device tensors; AVLMAPS_MERGE_KERNELS=0 keeps the tensor
    code, which is also the CPU twin (gloo tests) and what the GPU tests compare the kernels with"""
    if not t.is_cuda or os.environ.get("AVLMAPS_MERGE_KERNELS", "1") == "0":
        return None
    from . import _lib
    return _lib.load()


def _merge_work(lib, n, dev, rows=False):
    import ctypes as C
    import torch
    from . import _lib
    nb = C.c_size_t()
    _lib.check((lib.avl_merge_rows_work_bytes if rows else lib.avl_merge_work_bytes)(int(n), C.byref(nb)), "avl_merge_work_bytes")
    return torch.empty(int(nb.value), dtype=torch.uint8, device=dev), int(nb.value)


def _group_counts(sorted_ids, k: int, masks=()):
    """Histogram of a SORTED vector of group ids in [0, k) -- and, per boolean mask, how many of each group's elements have it
    set -- from the groups' boundaries (one searchsorted, a running sum per mask).  torch.bincount reads the largest value back
    to size its result: a host synchronisation per histogram; scatter_add would be millions of atomics on a handful of words."""
    import torch
    i64 = torch.int64
    dev = sorted_ids.device
    edges = torch.searchsorted(sorted_ids, torch.arange(k + 1, dtype=sorted_ids.dtype, device=dev))
    out = [edges[1:] - edges[:-1]]
    for m in masks:
        run = torch.cat([torch.zeros(1, dtype=i64, device=dev), torch.cumsum(m.to(i64), 0)])
        out.append(run[edges[1:]] - run[edges[:-1]])
    return out


def _dir_owner(cell64, ws: int):
    """directory rank of a linear cell: a multiplicative hash, so a map that occupies one corner of the grid still spreads evenly"""
    return (((cell64 * 2654435761) & 0xFFFFFFFF) >> 12) % ws


class _Trace:
    """AVLMAPS_MERGE_TRACE=1: wall time of every step of the merge on every rank (device synchronised at each mark), to stderr;
    with `coll` also the step's own time = wall minus what it spent inside collectives and waiting for a shared GPU"""

    def __init__(self, what, dev, coll=None):
        import time
        self.on = os.environ.get("AVLMAPS_MERGE_TRACE") == "1"
        self.cuda = getattr(dev, "type", str(dev)) == "cuda"
        self.coll = coll
        self.what, self.t, self.marks, self.t_abs = what, time.perf_counter(), [], time.time()
        self.busy = self._busy()

    def _busy(self):
        return (self.coll.comm_s + self.coll.gpu_lock.wait_s) if self.coll is not None else 0.0

    def __call__(self, label):
        if not self.on:
            return
        import time
        if self.cuda:
            import torch
            torch.cuda.synchronize()
        t, b = time.perf_counter(), self._busy()
        own = f" (own {1e3 * ((t - self.t) - (b - self.busy)):.1f})" if self.coll is not None else ""
        self.marks.append(f"{label} {1e3 * (t - self.t):.1f}{own}")
        self.t, self.busy = t, b

    def done(self, rank):
        if self.on:
            import sys
            print(f"[merge trace] rank {rank} {self.what} @{self.t_abs % 1000:.2f}s: " + " | ".join(self.marks) + " (ms)", file=sys.stderr, flush=True)


class ShardPlan:
    """What a rank knows after plan_merge_directory -- about its OWN voxels only:
        row_of_slot (n,) int64  final row (= the reference's voxel id) of every local voxel
        is_new      (n,) bool   this rank holds the voxel's global first touch (no lower rank has the cell)
        prev / next (n,) int64  the neighbouring contributors of the voxel in rank order, -1 if none: the colour replay of a
                                voxel runs prev -> this rank -> next; a voxel with prev == next == -1 belongs to this rank alone
        M, bases, counts        merged voxel count, and per rank the first row / the number of its new voxels
        n_prev / n_next         per rank r: how many local voxels have prev == r / next == r (host lists: chain_replay's hop sizes)
        grow_key                first-touch key of row `grow_row` (plan_merge_directory(grow_row=...)), all ones if M is smaller
        aux_all                 (ws, k) the `aux` integers of every rank (rides on the plan's first all_gather)
        monotone                False: the first-touch keys are not ordered by rank (frames were not sharded contiguously);
                                only aux_all is valid and the caller falls back to the general plan (plan_merge)"""
    __slots__ = ("M", "n", "row_of_slot", "is_new", "prev", "next", "bases", "counts", "grow_key", "aux_all", "monotone", "rank", "ws",
                 "coll", "dir_entries", "n_prev", "n_next")


def plan_merge_directory(cell: "torch.Tensor", first_key: "torch.Tensor", group=None, grow_row: Optional[int] = None, aux=(),
                         local: bool = False) -> ShardPlan:
    """The merge plan without any O(M) step on any rank (plan_merge sorts / uniques ALL M voxels on EVERY rank).

    Contiguous frame shards make the first-touch keys rank-monotone (every key of rank r < every key of rank r + 1; checked, with
    a fall-back), so the reference's voxel-id order is: the voxels first touched by rank 0 in key order, then those first touched
    by rank 1, ...  A rank therefore only has to learn which of ITS voxels a lower rank already holds:
      1. every rank sends its cell list to DIRECTORY ranks (hash of the cell; one all_to_all, 4 B / voxel); a directory rank
         sorts what it received (~ M / ws entries) by (cell, source rank) and answers every entry with the neighbouring
         contributors (prev, next) of that cell (second all_to_all, 4 B / voxel);
      2. a rank sorts its NEW voxels (prev == -1) by key; one tiny all_gather of the counts gives every rank its row base;
      3. the final rows of shared voxels reach their other contributors through the directory (two all_to_alls over the shared
         voxels only).
    Collective over the group; plain local arithmetic without torch.distributed."""
    import torch
    i64 = torch.int64
    dev = cell.device
    n = int(cell.shape[0])
    key = first_key.to(i64)
    coll = _Coll(group) if (_dist_on(group) and not local) else None      # local: this process alone (warm_up_merge)
    ws, rank = (coll.ws, coll.rank) if coll is not None else (1, 0)

    def a2a(t, ins, outs):
        return coll.all_to_all(t, ins, outs) if coll is not None else t

    def gather(t):
        return coll.all_gather(t) if coll is not None else [t]

    plan = ShardPlan()
    plan.rank, plan.ws, plan.coll, plan.n = rank, ws, coll, n
    tr = _Trace('plan', dev)
    hip = _merge_hip(cell) if (ws <= 64 and cell.dtype == torch.int32) else None
    st = None
    if hip is not None:
        from . import _lib
        from .device import torch_stream_ptr
        st = torch_stream_ptr()
        cell = cell.contiguous()
        key = key.contiguous()
        # one entry point: directory rank of every voxel, the slots grouped by it (stable), their cells in that order, the counts
        # and the key range (avl_merge_partition)
        ordd = torch.empty(n, dtype=i64, device=dev)
        cell_sorted = torch.empty(n, dtype=torch.int32, device=dev)
        head0 = torch.empty(ws + 3, dtype=i64, device=dev)
        work, wb = _merge_work(hip, n, dev)
        _lib.check(hip.avl_merge_partition(n, cell.data_ptr(), key.data_ptr(), ws, ordd.data_ptr(), cell_sorted.data_ptr(), head0.data_ptr(),
                                           work.data_ptr(), wb, st), "avl_merge_partition")
        head = torch.cat([head0, torch.tensor([int(a) for a in aux], dtype=i64).to(dev)]) if len(aux) else head0
        dest_o = None
    else:
        cell64 = cell.to(i64)
        tr('to i64')
        dest = _dir_owner(cell64, ws)
        tr('hash')
        ordd = _argsort_bits(dest, max(1, (ws - 1).bit_length()))     # local slots grouped by directory rank (stable)
        tr('argsort')
        dest_o = dest[ordd]
        sc, = _group_counts(dest_o, ws)
        tr('bincount')
        kmin = key.min().reshape(1) if n else torch.full((1,), I64_MAX, dtype=i64, device=dev)
        kmax = key.max().reshape(1) if n else torch.full((1,), -1, dtype=i64, device=dev)
        tr('minmax')
        hv = torch.tensor([n] + [int(a) for a in aux], dtype=i64).to(dev)          # one small upload
        head = torch.cat([sc, hv[:1], kmin, kmax, hv[1:]])
        cell_sorted = None
    tr('hash+sort+head')
    allh_d = torch.stack(gather(head))
    allh = allh_d.cpu()
    tr('gather head')
    plan.aux_all = allh[:, ws + 3:].numpy()
    last = -1
    plan.monotone = True
    for r in range(ws):
        if int(allh[r, ws]) > 0:
            if int(allh[r, ws + 1]) <= last:
                plan.monotone = False
            last = int(allh[r, ws + 2])
    if not plan.monotone:
        return plan
    sc_l = allh[rank, :ws].tolist()
    rc_l = allh[:, rank].tolist()
    # 1. directory: (cell, source rank) order -> neighbouring contributors of every entry
    recv = a2a(cell_sorted if cell_sorted is not None else cell[ordd].contiguous(), sc_l, rc_l)
    R = int(recv.shape[0])
    plan.dir_entries = R
    tr('a2a cells')
    if hip is not None:
        # directory side in one entry point (avl_merge_dir_scan): (cell, source rank) order, the neighbours of every entry, the
        # packed reply, the masks and counts of the two later round trips
        recv = recv.contiguous()
        perm = torch.empty(R, dtype=i64, device=dev)
        first = torch.empty(R, dtype=torch.uint8, device=dev)
        prev_r = torch.empty(R, dtype=i64, device=dev)
        next_r = torch.empty(R, dtype=i64, device=dev)
        reply = torch.empty(R, dtype=torch.int32, device=dev)
        m3r = torch.empty(R, dtype=torch.uint8, device=dev)
        m4r = torch.empty(R, dtype=torch.uint8, device=dev)
        cnt_dir = torch.empty(2 * ws + 1, dtype=i64, device=dev)
        work, wb = _merge_work(hip, R, dev)
        _lib.check(hip.avl_merge_dir_scan(R, recv.data_ptr(), allh_d[:, rank].contiguous().data_ptr(), ws, 31, perm.data_ptr(), first.data_ptr(),
                                          prev_r.data_ptr(), next_r.data_ptr(), reply.data_ptr(), m3r.data_ptr(), m4r.data_ptr(),
                                          cnt_dir.data_ptr(), work.data_ptr(), wb, st), "avl_merge_dir_scan")
        first, m3r, m4r = first.view(torch.bool), m3r.view(torch.bool), m4r.view(torch.bool)
        tr('directory sort')
        back = a2a(reply, rc_l, sc_l).contiguous()
        tr('a2a reply')
        # ... and home again (avl_merge_classify): prev / next / is_new per slot, the masks in sending order, every count
        prev = torch.empty(n, dtype=i64, device=dev)
        nxt = torch.empty(n, dtype=i64, device=dev)
        is_new = torch.empty(n, dtype=torch.uint8, device=dev)
        m3 = torch.empty(n, dtype=torch.uint8, device=dev)
        m4 = torch.empty(n, dtype=torch.uint8, device=dev)
        cnt_me = torch.empty(1 + 4 * ws, dtype=i64, device=dev)
        _lib.check(hip.avl_merge_classify(n, back.data_ptr(), ordd.data_ptr(), cell_sorted.data_ptr(), ws, prev.data_ptr(), nxt.data_ptr(),
                                          is_new.data_ptr(), m3.data_ptr(), m4.data_ptr(), cnt_me.data_ptr(), st), "avl_merge_classify")
        is_new, m3, m4 = is_new.view(torch.bool), m3.view(torch.bool), m4.view(torch.bool)
        cnt = torch.cat([cnt_me[0:1], cnt_dir[2 * ws:], cnt_me[1:1 + ws], cnt_dir[:ws], cnt_dir[ws:2 * ws], cnt_me[1 + ws:1 + 2 * ws],
                         cnt_me[1 + 2 * ws:1 + 3 * ws], cnt_me[1 + 3 * ws:]])
    else:
        cnt = None
    if cnt is None:
        # source rank of every arrival (arrivals are grouped by source): a search in the running sum of the counts, which are on the
        # device already (repeat_interleave with a host-side count list cost 0.9 ms of a 4 ms plan at 2.25 M entries)
        src = torch.bucketize(torch.arange(R, dtype=i64, device=dev), torch.cumsum(allh_d[:, rank], 0), right=True)
        perm = torch.argsort(recv, stable=True)                      # arrival order is by source rank: stable = (cell, rank) order
        cs, ss = recv[perm], src[perm]
        first = torch.ones(R, dtype=torch.bool, device=dev)
        lastm = torch.ones(R, dtype=torch.bool, device=dev)
        if R > 1:
            first[1:] = cs[1:] != cs[:-1]
            lastm[:-1] = first[1:]
        neg = torch.full((R,), -1, dtype=i64, device=dev)
        prev_r = torch.empty(R, dtype=i64, device=dev)
        next_r = torch.empty(R, dtype=i64, device=dev)
        prev_r[perm] = torch.where(first, neg, torch.roll(ss, 1))
        next_r[perm] = torch.where(lastm, neg, torch.roll(ss, -1))
        tr('directory sort')
        back = a2a(((prev_r + 1) | ((next_r + 1) << 16)).to(torch.int32), rc_l, sc_l).to(i64)
        tr('a2a reply')
        prev = torch.empty(n, dtype=i64, device=dev)
        nxt = torch.empty(n, dtype=i64, device=dev)
        prev[ordd] = (back & 0xFFFF) - 1
        nxt[ordd] = (back >> 16) - 1
        # 2. new voxels in key order; row bases
        is_new = prev < 0
        # The size of EVERY data-dependent list below -- my new voxels, the directory's distinct cells, the four lists of the two
        # directory round trips -- rides on one tiny all_gather, so that no boolean mask is ever counted on the host (_mask_idx).
        # The per-rank list sizes come from the group boundaries of the (sorted) rank vectors (_group_counts), not from histograms.
        m3 = (is_new & (nxt >= 0))[ordd]                             # my new voxels that others share, in sending order
        m4 = (~is_new)[ordd]                                         # my voxels whose row somebody else assigns
        m3r = (prev_r < 0) & (next_r >= 0)
        m4r = prev_r >= 0
        _, n3, n4 = _group_counts(dest_o, ws, (m3, m4))              # my entries are grouped by directory rank,
        _, n3r, n4r = _group_counts(src, ws, (m3r, m4r))             # the directory's arrivals by source rank
        ranks = torch.arange(ws, dtype=i64, device=dev)
        cnt = torch.cat([is_new.sum().reshape(1), first.sum().reshape(1), n3, n3r, n4r, n4,
                         (prev[:, None] == ranks).sum(0), (nxt[:, None] == ranks).sum(0)]).to(i64)   # ... and the replay's hop sizes
    tr('new voxels + counts')
    allc = torch.stack(gather(cnt)).cpu()
    tr('gather counts')
    counts = allc[:, 0].tolist()
    bases = [0] * ws
    for r in range(1, ws):
        bases[r] = bases[r - 1] + counts[r - 1]
    plan.M, plan.bases, plan.counts = int(sum(counts)), bases, counts
    n_first = int(allc[rank, 1])
    s3, r3, s4, r4, plan.n_prev, plan.n_next = (allc[rank, 2 + k * ws:2 + (k + 1) * ws].tolist() for k in range(6))
    c = int(counts[rank])
    if hip is not None:
        # the second half in three entry points: my new voxels in key order and their rows + the rows I report (avl_merge_rows_new);
        # the directory hands every entry its cell's first row (avl_merge_dir_rows); the rows I learn (avl_merge_rows_other)
        u8 = lambda t: t.view(torch.uint8)
        kbits = max(1, int(allh[:, ws + 2].max()).bit_length()) if n else 1
        row = torch.empty(n, dtype=i64, device=dev)
        idx_new = torch.empty(c, dtype=i64, device=dev)
        send3 = torch.empty(int(sum(s3)), dtype=i64, device=dev)
        work, wb = _merge_work(hip, max(n, R), dev, rows=True)
        _lib.check(hip.avl_merge_rows_new(n, c, u8(is_new).data_ptr(), key.data_ptr(), min(kbits, 63), int(bases[rank]), ordd.data_ptr(),
                                          u8(m3).data_ptr(), int(sum(s3)), row.data_ptr(), idx_new.data_ptr(), send3.data_ptr(),
                                          work.data_ptr(), wb, st), "avl_merge_rows_new")
        recv3 = a2a(send3, s3, r3).contiguous()
        tr('a2a first rows')
        send4 = torch.empty(int(sum(s4)), dtype=i64, device=dev)
        _lib.check(hip.avl_merge_dir_rows(R, recv3.data_ptr(), u8(m3r).data_ptr(), u8(first).data_ptr(), perm.data_ptr(), u8(m4r).data_ptr(),
                                          int(sum(s4)), send4.data_ptr(), work.data_ptr(), wb, st), "avl_merge_dir_rows")
        tr('propagate')
        recv4 = a2a(send4, s4, r4).contiguous()
        _lib.check(hip.avl_merge_rows_other(n, u8(m4).data_ptr(), ordd.data_ptr(), recv4.data_ptr(), int(sum(r4)), row.data_ptr(),
                                            work.data_ptr(), wb, st), "avl_merge_rows_other")
        tr('a2a other rows')
    else:
        idx_new = _mask_idx(is_new, c)
        idx_new = idx_new[torch.argsort(key[idx_new])]
        row = torch.full((n,), -1, dtype=i64, device=dev)
        row[idx_new] = bases[rank] + torch.arange(c, dtype=i64, device=dev)
        # 3. rows of shared voxels: first contributor -> directory -> the other contributors
        recv3 = a2a(row[ordd[_mask_idx(m3, sum(s3))]].contiguous(), s3, r3)
        tr('a2a first rows')
        rowfirst_r = torch.full((R,), -1, dtype=i64, device=dev)
        rowfirst_r[_mask_idx(m3r, sum(r3))] = recv3                  # both sides keep the order of the first all_to_all
        seg = torch.cumsum(first.to(i64), 0) - 1
        row_s = rowfirst_r[perm[_mask_idx(first, n_first)]][seg] if R else rowfirst_r
        row_r = torch.empty(R, dtype=i64, device=dev)
        row_r[perm] = row_s
        tr('propagate')
        recv4 = a2a(row_r[_mask_idx(m4r, sum(s4))].contiguous(), s4, r4)
        row[ordd[_mask_idx(m4, sum(r4))]] = recv4
        tr('a2a other rows')
    plan.row_of_slot, plan.is_new, plan.prev, plan.next = row, is_new, prev, nxt
    # the key after which the reference's arrays have their post-growth dtypes (vlmap_builder.py:286-311)
    plan.grow_key = U64_ALL_ONES
    if grow_row is not None and plan.M > grow_row >= 0:
        holder = max(r for r in range(ws) if bases[r] <= grow_row)
        val = torch.zeros(1, dtype=i64, device=dev)
        if rank == holder:
            val[0] = key[idx_new[grow_row - bases[rank]]]
        if coll is not None:
            coll.broadcast(val, holder)
        plan.grow_key = int(val.item()) & U64_ALL_ONES
    tr('grow key')
    tr.done(rank)
    return plan


def warm_up_merge(n: int = 1 << 20, device="cuda", D: int = 0, n_exchange: Optional[int] = None) -> None:
    """Run the tensor plumbing of a merge once, locally, at a realistic size.  torch loads the code objects of its sort / unique /
    scan kernels lazily, per process and per size class (a 30 k-voxel warm-up merge takes other sort kernels than a 1 M-voxel
    merge): ~1 s in a single process, and tens of seconds when 8 processes on one box load them at the same moment (seen in the
    8-ranks-on-one-GPU rehearsal, profiles/r04_build_8ranks_one_gpu.json: 27-52 s inside the first large torch.argsort on every
    rank).  A benchmark calls this in its warm-up; a production build simply pays it in its first checkpoint merge."""
    import torch
    g = torch.Generator(device="cpu").manual_seed(0)
    cell = torch.randperm(max(int(n), 2) * 4, generator=g)[:n].to(torch.int32).to(device)
    key = torch.randperm(max(int(n), 2), generator=g).to(device)
    plan = plan_merge_directory(cell, key, grow_row=n // 2, local=True)
    ex = MixedExchange.__new__(MixedExchange)
    rows = plan.row_of_slot
    order = torch.argsort(rows)
    torch.bincount(torch.clamp(rows[order] // max(1, n // 8), max=7), minlength=8)
    u, inv = torch.unique(rows % max(1, n // 3), return_inverse=True)
    buf = torch.zeros((u.shape[0], 4), dtype=torch.float64, device=device)
    buf.index_add_(0, inv, torch.ones((n, 4), dtype=torch.float64, device=device))
    out = torch.zeros((n, 8), dtype=torch.float32, device=device)
    out.index_copy_(0, order, torch.ones((n, 8), dtype=torch.float32, device=device))
    st = torch.zeros((n, 3), dtype=torch.int64, device=device)
    st[rows[rows % 2 == 0]] = 1
    torch.where((plan.next < 0)[:, None], st, torch.zeros_like(st))
    idx = torch.nonzero(plan.prev == -1).reshape(-1)
    idx[torch.argsort(cell[idx])]
    for m in (3, 40):                       # chain_replay's grouping: composite int64 keys over a subset of the slots
        sub = torch.nonzero(rows % m == 0).reshape(-1)
        sub = sub[torch.argsort(((rows[sub] % 7) << 32) | cell[sub].to(torch.int64))]
        torch.bincount(rows[sub] % 7, minlength=8)
        st[sub] = st[sub.flip(0)].contiguous()
    del ex
    if str(device).startswith("cuda"):
        torch.cuda.synchronize()


def chain_replay(plan: ShardPlan, cell: "torch.Tensor", replay_fn, tr=None) -> "torch.Tensor":
    """The sequential weight / colour replay across ranks, restricted to what has to be sequential.

    replay_fn(state (n, 3) int64, sel (n,) bool) continues the 24-byte state of the selected local voxels with this rank's
    sample log (avl_builder_replay_chain).  Voxels no lower rank holds (prev == -1: almost all of them) are replayed at once on
    every rank in parallel; a voxel shared with lower ranks waits for its predecessor's state, which arrives point-to-point
    from rank `prev` (lists in cell order on both sides, so no indices travel) -- the round-3 chain moved a dense 24 B x M
    state through every rank in turn.  Returns the local states; they are FINAL where plan.next == -1."""
    import torch
    i64 = torch.int64
    n, ws, rank, coll = plan.n, plan.ws, plan.rank, plan.coll
    dev = cell.device
    state = torch.zeros((n, 3), dtype=i64, device=dev)
    tr = tr or (lambda label: None)
    sel_a = plan.prev < 0
    if n:
        replay_fn(state, sel_a)
    tr('phase A (no lower rank)')
    if coll is None or ws == 1:
        return state
    # ONE sort per direction groups the shared voxels by neighbour rank, in cell order inside a group (the same order on both
    # sides of a hop, so no indices travel); the per-rank lists are slices
    # (the list sizes came with the plan: nothing is counted on the host here; voxels without a neighbour sort behind rank ws - 1)
    def grouped(which, total):
        k = (torch.where(which >= 0, which, torch.full_like(which, ws)) << 32) | cell.to(i64)
        return _argsort_bits(k, 32 + max(1, int(ws).bit_length()))[:int(total)]
    n_prev, n_next = [int(v) for v in plan.n_prev], [int(v) for v in plan.n_next]
    idx_p = grouped(plan.prev, sum(n_prev))
    idx_n = grouped(plan.next, sum(n_next))
    tr('group by neighbour')
    if sum(n_prev):
        buf = torch.empty((int(idx_p.shape[0]), 3), dtype=i64, device=dev)
        o = 0
        for p in range(rank):
            if n_prev[p]:
                coll.recv(buf[o:o + n_prev[p]], p)
                o += n_prev[p]
        state[idx_p] = buf
    tr('recv from prev ranks')
    if sum(n_prev):
        replay_fn(state, ~sel_a)
    tr('phase B (continued)')
    if sum(n_next):
        out = state[idx_n].contiguous()
        o = 0
        for q in range(rank + 1, ws):
            if n_next[q]:
                coll.send(out[o:o + n_next[q]], q)
                o += n_next[q]
    tr('send to next ranks')
    return state


class MixedExchange:
    """Row-sharded exchange with the mixed payload.  Final rows are dealt out in contiguous blocks shard_rows(M, r, ws); a rank's
    voxels, sorted by final row, are grouped by destination.  Three lists travel (all_to_all_single with split sizes each):
        side    every local voxel: [row | cell << 32, sum_w4 (4 x f64), replay state (3 x i64)] = 64 B
        done    voxels of this rank ALONE (prev == next == -1): the finished float32 feature row, D x 4 B
        part    voxels shared between ranks: float64 partial sums of the features, D x 8 B"""
    __slots__ = ("order", "rows_sorted", "single_sorted", "send_all", "recv_all", "send_done", "recv_done", "send_part", "recv_part",
                 "r0", "r1", "per")

    def __init__(self, plan: ShardPlan):
        import torch
        i64 = torch.int64
        rows = plan.row_of_slot
        ws, M = plan.ws, plan.M
        self.per = max(1, (M + ws - 1) // ws)
        self.r0, self.r1 = shard_rows(M, plan.rank, ws)
        self.order = _argsort_bits(rows, max(1, int(M).bit_length()))   # rows are distinct: any sort is the stable one
        self.rows_sorted = rows[self.order].contiguous()
        single = (plan.prev < 0) & (plan.next < 0)
        self.single_sorted = single[self.order]
        dest = torch.clamp(self.rows_sorted // self.per, max=ws - 1)
        n_all, n_single = _group_counts(dest, ws, (self.single_sorted,))      # rows are sorted, so their destinations are
        cnt = torch.stack([n_all, n_single, n_all - n_single]).to(i64)
        if plan.coll is not None:
            allc = torch.stack(plan.coll.all_gather(cnt)).cpu()                         # (ws, 3, ws): [sender, list, receiver]
            recv = allc[:, :, plan.rank]
        else:
            recv = cnt.cpu().t()
        cnt = cnt.cpu()
        self.send_all, self.send_done, self.send_part = (cnt[k].tolist() for k in range(3))
        self.recv_all, self.recv_done, self.recv_part = (recv[:, k].tolist() for k in range(3))


def _fold_mixed(plan, ex, D, side, done, part, rows_add, tr=None, unpack=None):
    """owner side of the mixed exchange, shared by the device path and its torch twin: returns (own_cell, w4 (n_own, 4) f64,
    state (n_own, 3) i64, done_rows, done_feat, part_rows (k,), part_acc (k, D) f64, bad_rows (1,) int32 = some received row
    index lay outside the block) -- rows relative to this rank's block"""
    import torch
    i64 = torch.int64
    dev = side.device
    tr = tr or (lambda label: None)
    n_own = ex.r1 - ex.r0
    word = side[:, 0]
    if unpack is not None:                          # device path: one kernel (avl_merge_side_unpack; bad rows go to its error flag)
        rows, own_cell, state = unpack(side)
        bad_rows = torch.zeros(1, dtype=torch.int32, device=dev)
    else:
        rows = (word & 0xFFFFFFFF) - ex.r0
        # a row outside this rank's block would be a plan / exchange bug: never index with it (a device-side assert would take the
        # process down before anybody could report it) -- clamp, and let the caller raise on EVERY rank (bad_rows)
        bad_rows = ((rows < 0) | (rows >= n_own)).any().reshape(1).to(torch.int32) if rows.numel() else torch.zeros(1, dtype=torch.int32, device=dev)
        rows = rows.clamp(0, max(n_own - 1, 0))
        own_cell = torch.zeros(max(n_own, 1), dtype=torch.int32, device=dev)
        own_cell[rows] = ((word >> 32) & 0x7FFFFFFF).to(torch.int32)
        own_cell = own_cell[:n_own]
        st = side[:, 5:8]
        fin = (st[:, 2] >> 32) != 0                 # `started` of the 24-byte state: only a voxel's LAST contributor sends it
        # (no boolean-mask indexing here: every mask would be counted on the host.  States that are not final land in one spare row)
        state = torch.zeros((max(n_own, 1) + 1, 3), dtype=i64, device=dev)
        state[torch.where(fin, rows, torch.full_like(rows, max(n_own, 1)))] = st
        state = state[:max(n_own, 1)]
    tr('rows + cells + states')
    w4 = torch.zeros((max(n_own, 1), 4), dtype=torch.float64, device=dev)
    o = 0
    for c in ex.recv_all:                           # peer by peer, in rank order: a reproducible float64 sum
        if c:
            rows_add(rows[o:o + c], side[o:o + c, 1:5].view(torch.float64), w4)
        o += c
    tr('w4 adds')
    # rows of the two feature lists: the side list of a peer is in final-row order, and so are its done / part sublists
    single_flag = (word >> 63) != 0                 # bit 63 of the word: the voxel travelled as a finished row
    done_rows = rows[_mask_idx(single_flag, sum(ex.recv_done))]
    part_rows_all = rows[_mask_idx(~single_flag, sum(ex.recv_part))]
    tr('lists')
    part_rows, inv = (torch.unique(part_rows_all, return_inverse=True) if part_rows_all.numel() else
                      (part_rows_all, part_rows_all))
    tr('unique')
    acc = torch.zeros((max(int(part_rows.shape[0]), 1), D), dtype=torch.float64, device=dev)
    o = 0
    for c in ex.recv_part:
        if c:
            rows_add(inv[o:o + c], part[o:o + c], acc)
        o += c
    tr('part adds')
    return own_cell, w4, state, done_rows, done, part_rows, acc, bad_rows


def merge_raw_sharded(raw: Dict[str, "torch.Tensor"], group=None, replay_fn=None, gs2: Optional[int] = None):
    """Row-sharded merge of per-rank raw accumulators given as torch tensors: the arithmetic and the choreography of
    merge_accumulator_sharded on exported tensors (CPU + gloo in the tests, the cross-check of the device path on the GPU).
    Returns on EVERY rank dict(M, rows=(r0, r1), cell (n_own,) int32 and first_key (n_own,) int64 of this rank's block of final
    rows, grid_feat (n_own, D) float32, w4 (n_own, 4) float64 = summed [alpha, alpha rgb], part_rows (k,) = the block's rows that
    several ranks touched with part_acc (k, D) float64 their summed features (first-touch term folded in), bytes_sent,
    payload_bytes_fp64_form, plan="directory" | "general"; with replay_fn also state (n_own, 3) int64, the final replay states)."""
    import torch
    i64 = torch.int64
    cell, key = raw["cell"], raw["first_key"].to(i64)
    dev = cell.device
    D = raw["sum_feat"].shape[1]
    plan = plan_merge_directory(cell, key, group, grow_row=None if gs2 is None else gs2 - 1)
    if not plan.monotone or os.environ.get("AVLMAPS_MERGE_PLAN") == "general":
        out = _merge_raw_sharded_general(raw, group)
        r0, r1 = out["rows"]
        ar = torch.arange(r1 - r0, device=dev)
        out.update(cell=out["cell"][r0:r1], first_key=out["first_key"][r0:r1], plan="general", w4=out["acc"][:, D:], part_rows=ar,
                   part_acc=out["acc"][:, :D], grid_feat=(out["acc"][:, :D] / out["acc"][:, D:D + 1]).float())
        return out
    ex = MixedExchange(plan)
    n = plan.n
    a1 = raw["first_alpha"]
    wf = torch.where(plan.is_new, a1 * a1, a1)
    contrib = wf[:, None] * raw["first_feat"].double() + raw["sum_feat"]
    o = ex.order
    single = ex.single_sorted
    done = (contrib[o][single] / raw["sum_w4"][o][single][:, :1]).float().contiguous()
    part = contrib[o][~single].contiguous()
    state = torch.zeros((n, 3), dtype=i64, device=dev)
    if replay_fn is not None:
        state = chain_replay(plan, cell, replay_fn)
        state = torch.where((plan.next < 0)[:, None], state, torch.zeros_like(state))
    side = torch.empty((n, 8), dtype=i64, device=dev)
    side[:, 0] = ex.rows_sorted | (cell[o].to(i64) << 32) | (single.to(i64) << 63)
    side[:, 1:5] = raw["sum_w4"][o].contiguous().view(i64)
    side[:, 5:8] = state[o]
    keys_o = torch.where(plan.is_new, key, torch.full_like(key, I64_MAX))[o].contiguous()
    if plan.coll is not None:
        side = plan.coll.all_to_all(side, ex.send_all, ex.recv_all)
        done = plan.coll.all_to_all(done, ex.send_done, ex.recv_done)
        part = plan.coll.all_to_all(part, ex.send_part, ex.recv_part)
        keys_o = plan.coll.all_to_all(keys_o, ex.send_all, ex.recv_all)

    def rows_add(rows, src, dst):
        dst.index_add_(0, rows, src)
    own_cell, w4, own_state, done_rows, done_feat, part_rows, part_acc, bad_rows = _fold_mixed(plan, ex, D, side, done, part, rows_add)
    if int(bad_rows.item()):
        raise RuntimeError("multi-rank merge: a received row index lies outside its owner's block of final rows")
    n_own = ex.r1 - ex.r0
    grid_feat = torch.zeros((n_own, D), dtype=torch.float32, device=dev)
    grid_feat[done_rows] = done_feat
    if part_rows.numel():
        grid_feat[part_rows] = (part_acc / w4[part_rows, :1]).float()
    own_key = torch.full((max(n_own, 1),), I64_MAX, dtype=i64, device=dev)
    own_key.scatter_reduce_(0, (side[:, 0] & 0xFFFFFFFF) - ex.r0, keys_o, reduce="amin")
    sent_all = sum(c for r, c in enumerate(ex.send_all) if r != plan.rank)
    sent_done = sum(c for r, c in enumerate(ex.send_done) if r != plan.rank)
    sent_part = sum(c for r, c in enumerate(ex.send_part) if r != plan.rank)
    out = dict(M=plan.M, rows=(ex.r0, ex.r1), cell=own_cell, first_key=own_key[:n_own], grid_feat=grid_feat, w4=w4[:n_own],
               part_rows=part_rows, part_acc=part_acc[:int(part_rows.shape[0])], plan="directory",
               bytes_sent=sent_all * 64 + sent_done * D * 4 + sent_part * D * 8, payload_bytes_fp64_form=sent_all * ((D + 4) * 8 + 8),
               grow_key=plan.grow_key)
    if replay_fn is not None:
        out["state"] = own_state[:n_own]
    return out


def merge_accumulator_sharded(acc, group=None, exact_rgb: bool = True, timings: Optional[dict] = None, gather_to: Optional[int] = None,
                              status: int = 0):
    """The product path of the multi-GPU build, row-sharded (round 4: directory plan + mixed payload + point-to-point replay).

    Merges the ranks' VoxelAccumulators with ONE sparse exchange and finalises every rank's block of final rows where it lives.
    Nothing a rank computes is O(M): the plan (plan_merge_directory) touches its own voxels and ~M / ws directory entries; a voxel
    that only this rank touched (the bulk: contiguous frame shards see mostly disjoint space) is FINISHED here in float64 and
    ships as a float32 row (avl_builder_export_rows_f32: bit-identical to the single-process map), only voxels several ranks
    touched ship float64 partial sums (avl_builder_export_rows_f64 -> avl_rows_add_f64 in rank order at the owner); the exact
    sequential weight / colour replay runs in parallel for every voxel no lower rank holds and point-to-point, rank prev -> next,
    for the shared ones (chain_replay).  torch.distributed carries the collectives (backend nccl == RCCL over xGMI).

    Returns on every rank a dict of DEVICE tensors
        M, rows=(r0, r1), grid_feat (r1-r0, D) f32, grid_pos (r1-r0, 3) i32, weight (r1-r0,) f32, grid_rgb (r1-r0, 3) u8, cell (r1-r0,)
    = this rank's block of the merged map in the reference's voxel-id order (the block VLMap.shard_index_rows scores).
    gather_to = r: rank r also gets "full": the whole map (grid_feat, grid_pos, weight, grid_rgb, occupied_ids) as device tensors,
    e.g. to write the file.  exact_rgb needs the replay log on every rank.  status != 0 on ANY rank makes every rank raise
    (a rank that failed locally still joins this collective, so that nobody waits for it until the RCCL timeout)."""
    import torch
    torch.cuda.synchronize()
    glock = _SharedGpuLock.get()
    glock.acquire()                             # (a no-op unless several ranks share one GPU in a rehearsal)
    prof = None
    if timings is not None and os.environ.get("AVLMAPS_MERGE_PROFILE") == str(rank_world(group)[0]):
        from torch.profiler import ProfilerActivity, profile      # developer aid: where a rank's host + device time goes
        prof = profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA])
        prof.__enter__()
    try:
        which = os.environ.get("AVLMAPS_MERGE_PLAN", "gather")
        if which == "gather":
            # the product path: two all_gathers, ONE payload all_to_all, every local stretch one HIP entry point (avlmaps_amd/merge2.py)
            from . import merge2
            out = merge2.merge_accumulator_v2(acc, group, exact_rgb, timings, gather_to, status, glock)
            if out is not None:
                return out
            out = _merge_accumulator_sharded_general(acc, group, exact_rgb, timings, gather_to)      # keys not ordered by rank
            r0, r1 = out["rows"]
            out["cell"] = out["cell"][r0:r1]
            return out
        return _merge_accumulator_sharded_directory(acc, group, exact_rgb, timings, gather_to, status, glock)
    finally:
        glock.release()
        if prof is not None:
            import sys
            prof.__exit__(None, None, None)
            print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=40), file=sys.stderr, flush=True)
            try:
                print(prof.key_averages().table(sort_by="self_device_time_total", row_limit=60, max_name_column_width=110), file=sys.stderr, flush=True)
            except Exception:       # (older torch: the key is self_cuda_time_total)
                print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=60, max_name_column_width=110), file=sys.stderr, flush=True)


def _merge_accumulator_sharded_directory(acc, group, exact_rgb, timings, gather_to, status, glock):
    import time
    import torch
    from . import _lib
    from .device import torch_stream_ptr
    lib = _lib.load()
    st = torch_stream_ptr()
    dev = torch.device("cuda", torch.cuda.current_device())
    i64 = torch.int64
    null_launch_us = None
    if os.environ.get("AVLMAPS_MERGE_TRACE") == "1":
        # what ONE tiny kernel launch costs in this set-up (a dedicated GPU: 4-6 us; eight processes taking turns on one GPU:
        # 30-40 us -- the merge's tensor plumbing is a few hundred such launches, so this factor scales its compute time)
        z = torch.zeros(8, device=dev)
        torch.cuda.synchronize()
        tq = time.perf_counter()
        for _ in range(64):
            z.add_(1.0)
        torch.cuda.synchronize()
        null_launch_us = (time.perf_counter() - tq) / 64 * 1e6
    t0 = time.perf_counter()
    lw0 = glock.wait_s
    tr = _Trace('merge', dev)
    n = acc.num_voxels(st) if not status else 0
    tr('num_voxels')
    cell = torch.empty((n,), dtype=torch.int32, device=dev)
    key = torch.empty((n,), dtype=i64, device=dev)
    w4_loc = torch.empty((max(n, 1), 4), dtype=torch.float64, device=dev)
    if n:
        _lib.check(lib.avl_builder_export_raw(acc._h, n, cell.data_ptr(), key.data_ptr(), None, w4_loc.data_ptr(), None, None, st),
                   "avl_builder_export_raw")
    have_log = 1 if (exact_rgb and acc.has_replay_log()) else 0
    tr('export cell/key/w4')
    plan = plan_merge_directory(cell, key, group, grow_row=acc.n_rows * acc.gs - 1, aux=(have_log, int(status)))
    tr('plan')
    bad = [r for r in range(plan.ws) if int(plan.aux_all[r, 1]) != 0]
    if bad:
        raise RuntimeError(f"multi-rank merge aborted: rank(s) {bad} reported a failure (status {[int(plan.aux_all[r, 1]) for r in bad]})")
    if not plan.monotone or os.environ.get("AVLMAPS_MERGE_PLAN") == "general":
        out = _merge_accumulator_sharded_general(acc, group, exact_rgb, timings, gather_to)
        r0, r1 = out["rows"]
        out["cell"] = out["cell"][r0:r1]
        return out
    have_log = int(plan.aux_all[:, 0].min())
    coll = plan.coll
    D, M = acc.D, plan.M
    ex = MixedExchange(plan)
    tr('exchange bookkeeping')
    tr.done(plan.rank)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    c1 = coll.comm_s if coll is not None else 0.0
    lw1 = glock.wait_s
    # ---- local rows -> the three send lists (final-row order = destination order)
    o32 = ex.order.to(torch.int32)
    single = ex.single_sorted
    n_done, n_part = int(sum(ex.send_done)), int(sum(ex.send_part))     # known on the host: no mask is counted there (_mask_idx)
    i_part = _mask_idx(~single, n_part)
    slots_done = o32[_mask_idx(single, n_done)].contiguous()
    slots_part = o32[i_part].contiguous()
    done = torch.empty((max(n_done, 1), D), dtype=torch.float32, device=dev)
    part = torch.empty((max(n_part, 1), D), dtype=torch.float64, device=dev)
    _lib.check(lib.avl_builder_export_rows_f32(acc._h, n_done, slots_done.data_ptr(), done.data_ptr(), D, st), "avl_builder_export_rows_f32")
    own_part = plan.is_new[ex.order[i_part]].to(torch.uint8).contiguous()
    _lib.check(lib.avl_builder_export_rows_f64(acc._h, n_part, slots_part.data_ptr(), own_part.data_ptr(), part.data_ptr(), D, st),
               "avl_builder_export_rows_f64")
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    c2 = coll.comm_s if coll is not None else 0.0
    lw2 = glock.wait_s
    # ---- exact sequential weight / colour: parallel where it can be, point-to-point where it has to be sequential
    state = torch.zeros((n, 3), dtype=i64, device=dev)
    chain_bytes0 = coll.bytes_out if coll is not None else 0
    if have_log:
        ar = torch.arange(n, dtype=i64, device=dev)
        minus = torch.full((n,), -1, dtype=i64, device=dev)

        def replay_fn(stt, sel):
            idx = torch.where(sel, ar, minus).contiguous()
            _lib.check(lib.avl_builder_replay_chain(acc._h, n, idx.data_ptr(), plan.grow_key, stt.data_ptr(), st), "avl_builder_replay_chain")
        state = chain_replay(plan, cell, replay_fn, tr2 := _Trace('replay', dev, coll))
        tr2.done(plan.rank)               # (only a voxel's last contributor sends its state: avl_merge_side_pack looks at plan.next)
    chain_bytes = (coll.bytes_out - chain_bytes0) if coll is not None else 0
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    c3 = coll.comm_s if coll is not None else 0.0
    lw3 = glock.wait_s
    side = torch.empty((n, 8), dtype=i64, device=dev)
    _lib.check(lib.avl_merge_side_pack(n, ex.order.contiguous().data_ptr(), ex.rows_sorted.data_ptr(), single.contiguous().view(torch.uint8).data_ptr(),
                                       cell.data_ptr(), w4_loc.data_ptr(), state.contiguous().data_ptr() if have_log else None,
                                       plan.next.contiguous().data_ptr(), side.data_ptr(), st), "avl_merge_side_pack")
    if coll is not None:
        side = coll.all_to_all(side, ex.send_all, ex.recv_all)
        done = coll.all_to_all(done[:n_done], ex.send_done, ex.recv_done)
        part = coll.all_to_all(part[:n_part], ex.send_part, ex.recv_part)
    else:
        done, part = done[:n_done], part[:n_part]
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    c4 = coll.comm_s if coll is not None else 0.0
    lw4 = glock.wait_s
    # ---- owner side: fold what arrived into this rank's block of final rows
    n_own = ex.r1 - ex.r0

    err_flag = torch.zeros(1, dtype=torch.int32, device=dev)

    def rows_add(rows, src, dst):
        if src.stride(1) != 1:
            src = src.contiguous()                  # (a column window of wider rows goes as it is: ld = its row stride)
        rows = rows.contiguous()
        _lib.check(lib.avl_rows_add_f64_async(int(rows.shape[0]), int(src.shape[1]), rows.data_ptr(), 0, int(dst.shape[0]), src.data_ptr(),
                                              int(src.stride(0)), dst.data_ptr(), int(dst.shape[1]), err_flag.data_ptr(), st), "avl_rows_add_f64_async")
    tr3 = _Trace('fold', dev, coll)
    def unpack(side):
        R = int(side.shape[0])
        rows = torch.empty((R,), dtype=i64, device=dev)
        own_cell = torch.zeros(max(n_own, 1), dtype=torch.int32, device=dev)
        own_state = torch.zeros((max(n_own, 1), 3), dtype=i64, device=dev)
        _lib.check(lib.avl_merge_side_unpack(R, side.data_ptr(), ex.r0, n_own, rows.data_ptr(), own_cell.data_ptr(), own_state.data_ptr(),
                                             err_flag.data_ptr(), st), "avl_merge_side_unpack")
        return rows, own_cell[:n_own], own_state
    own_cell, w4, own_state, done_rows, done_feat, part_rows, part_acc, bad_rows = _fold_mixed(plan, ex, D, side, done, part, rows_add, tr3, unpack)
    # every rank learns whether ANY rank saw a bad row and raises with it: a rank raising alone would leave the others in
    # gather_row_shards' collectives
    err_flag = torch.maximum(err_flag, bad_rows.to(err_flag.device))
    if coll is not None:
        err_flag = coll.all_reduce(err_flag, coll.dist.ReduceOp.MAX)
    if int(err_flag.item()):
        raise RuntimeError("multi-rank merge: a received row index lies outside its owner's block of final rows")
    tr3('fold lists')
    out = dict(M=M, rows=(ex.r0, ex.r1), cell=own_cell,
               grid_feat=torch.empty((n_own, D), dtype=torch.float32, device=dev),
               grid_pos=torch.empty((n_own, 3), dtype=torch.int32, device=dev),
               weight=torch.empty((n_own,), dtype=torch.float32, device=dev),
               grid_rgb=torch.empty((n_own, 3), dtype=torch.uint8, device=dev))
    tr3('alloc out')
    if n_own:
        if D % 4 == 0 and done_rows.numel():
            # the finished rows that arrived, to their final positions: one pass at the copy rate (torch's index_copy_ took 3.3 ms
            # for the 4.6 GB of a 2.25 M-voxel map, this kernel 1.6)
            df = done_feat.contiguous()
            _lib.check(lib.avl_scatter_rows(df.data_ptr(), D * 4, done_rows.contiguous().data_ptr(), int(done_rows.shape[0]),
                                            out["grid_feat"].data_ptr(), n_own, err_flag.data_ptr(), st), "avl_scatter_rows")
        else:
            out["grid_feat"].index_copy_(0, done_rows, done_feat)
        tr3('copy done rows')
        if part_rows.numel():
            # (one kernel, no (k, D) float64 / float32 temporaries: the tensor expression's allocations were the erratic part of a
            # rank's fold when eight processes share one device)
            _lib.check(lib.avl_rows_div_f32(int(part_rows.shape[0]), D, part_acc.data_ptr(), part_rows.contiguous().data_ptr(), w4.data_ptr(),
                                            out["grid_feat"].data_ptr(), n_own, err_flag.data_ptr(), st), "avl_rows_div_f32")
        tr3('divide shared rows')
        _lib.check(lib.avl_finalize_side(n_own, ex.r0, acc.gs, acc.vh, own_cell.data_ptr(), w4.data_ptr(), out["grid_pos"].data_ptr(),
                                         out["weight"].data_ptr(), out["grid_rgb"].data_ptr(), None, st), "avl_finalize_side")
        if have_log:
            _lib.check(lib.avl_replay_state_apply(n_own, own_state.data_ptr(), out["weight"].data_ptr(), out["grid_rgb"].data_ptr(), st),
                       "avl_replay_state_apply")
    tr3('side + state apply')
    tr3.done(plan.rank)
    del side, done, part, part_acc
    torch.cuda.synchronize()
    t5 = time.perf_counter()
    c5 = coll.comm_s if coll is not None else 0.0
    lw5 = glock.wait_s
    gather_bytes = 0
    if gather_to is not None:
        full = gather_row_shards(out, gather_to, coll, plan.rank, plan.ws, names=("grid_feat", "grid_pos", "weight", "grid_rgb", "cell"))
        if plan.rank != gather_to:
            gather_bytes = n_own * (D * 4 + 12 + 4 + 3 + 4)
        if full is not None:
            full["occupied_ids"] = occupied_ids_from_cells(full.pop("cell"), acc.n_rows, acc.gs, acc.vh)
            out["full"] = full
    torch.cuda.synchronize()
    t6 = time.perf_counter()
    c6 = coll.comm_s if coll is not None else 0.0
    lw6 = glock.wait_s
    if timings is not None:
        W = D + 4
        sent_all = sum(c for r, c in enumerate(ex.send_all) if r != plan.rank)
        sent_done = sum(c for r, c in enumerate(ex.send_done) if r != plan.rank)
        sent_part = sum(c for r, c in enumerate(ex.send_part) if r != plan.rank)
        payload = sent_all * 64 + sent_done * D * 4 + sent_part * D * 8
        plan_bytes = (n * 8 + (plan.dir_entries or 0) * 4 + 16 * plan.ws * plan.ws) if coll is not None else 0
        comm = dict(plan=c1, export=c2 - c1, replay_chain=c3 - c2, exchange=c4 - c3, fold_finalize=c5 - c4, gather=c6 - c5)
        lockw = dict(plan=lw1 - lw0, export=lw2 - lw1, replay_chain=lw3 - lw2, exchange=lw4 - lw3, fold_finalize=lw5 - lw4, gather=lw6 - lw5)
        wall = dict(plan=t1 - t0, export=t2 - t1, replay_chain=t3 - t2, exchange=t4 - t3, fold_finalize=t5 - t4, gather=t6 - t5)
        wall = {k: wall[k] - lockw[k] for k in wall}          # rehearsals on one GPU: waiting for the shared device is not merge time
        timings.update(mode="row-sharded all_to_all", plan="directory (nothing O(M) per rank; mixed float32 / float64 payload; point-to-point replay)",
                       plan_s=wall["plan"], scatter_s=wall["export"], replay_chain_s=wall["replay_chain"], exchange_s=wall["exchange"],
                       accumulate_s=wall["fold_finalize"], finalize_s=0.0, gather_s=wall["gather"], null_launch_us=null_launch_us, wall_s=wall, in_collectives_s=comm, shared_gpu_wait_s=sum(lockw.values()),
                       compute_s={k: wall[k] - comm[k] for k in wall},
                       compute_total_s=sum(wall[k] - comm[k] for k in wall if k != "gather"),
                       in_collectives_total_s=sum(comm[k] for k in comm if k != "gather"),
                       merged_voxels=M, local_voxels=n, own_rows=n_own, new_voxels=int(plan.counts[plan.rank]), single_rank_voxels=n_done,
                       shared_voxels_local=n_part, shared_rows_owned=int(part_rows.numel()), directory_entries=int(plan.dir_entries or 0),
                       rows_sent=sent_all, payload_bytes_sent=payload, payload_bytes_fp64_form=sent_all * (W * 8 + 8),
                       plan_bytes_sent=plan_bytes, chain_bytes_sent=chain_bytes, gather_bytes_sent=gather_bytes,
                       bytes_sent_per_rank=payload + plan_bytes + chain_bytes + gather_bytes,
                       local_row_bytes=n * W * 8, dense_reduce_payload_bytes=M * W * 8, exact_rgb=bool(have_log), collectives=(coll.calls if coll else 0),
                       world_size=plan.ws, backend=(coll.dist.get_backend(coll.group) if coll is not None else "none"))
    return out


def gather_row_shards(shard: dict, dst: int, coll, rank: int, ws: int, names=("grid_feat", "grid_pos", "weight", "grid_rgb")):
    """the ranks' blocks of finished rows (grid_feat / grid_pos / weight / grid_rgb device tensors, rank order = row order) ->
    the whole arrays on rank `dst`; None elsewhere.  One all_to_all_single per array in which only `dst` receives."""
    import torch
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
        wf = torch.where(own, a1 * a1, a1)
        acc[idx, :D] += wf[:, None] * r["first_feat"].double() + r["sum_feat"]
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


def global_topk(vals: "torch.Tensor", rows: "torch.Tensor", row_offset: int, k: int, group=None):
    """Per-query k best voxels over row shards ("allgather of k * Q candidates"): vals / rows (Q, k_local) = this
    rank's candidates per query (k_local <= k; e.g. ops.topk_f32 per query column), local row indices; returns (values (Q, k),
    global rows (Q, k)) ordered like np.argsort(-v, kind="stable") over the whole map: descending value, ties by ascending global
    row (ranks hold ascending row blocks), NaN last.  Missing candidates (fewer than k voxels) are (-inf, -1).  The exchange is
    k * Q (value, row) pairs per rank -- never features."""
    import torch
    vals = vals.to(torch.float64) if vals.dtype == torch.float64 else vals.float()
    Q, kl = int(vals.shape[0]), int(vals.shape[1])
    g_rows = rows.to(torch.int64) + int(row_offset)
    pad = max(0, int(k) - kl)
    if pad:
        vals = torch.cat([vals, torch.full((Q, pad), float("-inf"), dtype=vals.dtype, device=vals.device)], dim=1)
        g_rows = torch.cat([g_rows, torch.full((Q, pad), -1, dtype=torch.int64, device=vals.device)], dim=1)
    vals, g_rows = vals[:, :k].contiguous(), g_rows[:, :k].contiguous()
    if _dist_on(group):
        coll = _Coll(group)
        vals = torch.cat(coll.all_gather(vals), dim=1)                      # (Q, ws * k)
        g_rows = torch.cat(coll.all_gather(g_rows), dim=1)
    # order: value descending with NaN last, then global row ascending; padding (-inf, -1) sorts behind every real candidate
    key_v = torch.where(torch.isnan(vals), torch.full_like(vals, float("-inf")), vals)
    is_pad = g_rows < 0
    big = torch.iinfo(torch.int64).max
    r_key = torch.where(is_pad, torch.full_like(g_rows, big), g_rows)
    nan_last = torch.isnan(vals) | is_pad
    # lexicographic sort by (nan_last asc, value desc, row asc): three stable sorts, least significant key first
    order = torch.argsort(r_key, dim=1, stable=True)
    order = order.gather(1, torch.argsort(key_v.gather(1, order), dim=1, descending=True, stable=True))
    order = order.gather(1, torch.argsort(nan_last.gather(1, order).to(torch.int8), dim=1, stable=True))
    order = order[:, :k]
    return vals.gather(1, order), g_rows.gather(1, order)
