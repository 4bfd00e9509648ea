"""The N-rank merge of the map build, "gather plan" form: two small all_gathers, the replay hops, ONE payload all_to_all.

What is merged: the ranks' voxel maps of the reference's builder loop (avlmaps/map/vlmap_builder.py:102-183), frames sharded
contiguously over the ranks (parallel.shard_frames).  A voxel's final row is the reference's voxel id = its position in first-touch
order (vlmap_builder.py:163-170); rank r ends up with the block parallel.shard_rows(M, r, ws) of the merged map in HBM.

Choreography (torch.distributed carries the collectives; every stretch of local work between two of them is ONE entry point of
csrc/avl_merge2.hip / avl_builder_m2_pack -- no torch sort / unique / index / nonzero on the device path, so the first merge of a
process costs what the tenth does):
    header      all_gather of [n, min key, max key, flags] per rank                           (32 B per rank)
    lists       all_gather of every rank's (first-touch key, cell) list                       (12 B per voxel)
    plan        avl_merge2_plan on every rank: union of the cells, reference row of every cell, and for the rank's own voxels the
                final row, the neighbouring contributors, the send order; the ws x ws size tables come back in ONE read-back
    pack        avl_builder_m2_pack: own voxels in final-row order straight from the accumulators into the send buffer -- a voxel of
                this rank alone as its FINISHED float32 row (bit-identical to the single-process map), a voxel several ranks touched
                as float64 partial sums; rows this rank owns itself never leave (single-rank ones are written into the block directly)
    replay      exact sequential weight / colour (vlmap_builder.py:164-178 dtypes): voxels no lower rank holds replay at once on
                every rank; for the shared ones 24 B of state hop rank -> next contributor, one small all_to_all per receiving rank
    exchange    ONE all_to_all_single: per destination [side records 64 B | float32 rows | float64 rows] -- or, for blocks above
                AVLMAPS_MERGE_CHUNK_MB, one per chunk of R rows of every owner's block: pack(c + 1) / exchange(c) / fold(c - 1) overlap
                (parallel._Coll.all_to_all_start / _finish), two send buffers of one chunk each
    fold        avl_merge2_fold: wave per row of the block, contributors summed in rank order, finished with finalize's expressions
The same choreography runs on CPU tensors with the NumPy twin of the kernels (HostKernels: gloo tests, and what the GPU tests
compare the HIP kernels with).  Keys that are not ordered by rank (frames not sharded contiguously) fall back to
parallel._merge_accumulator_sharded_general / the general plan of parallel.merge_raw_sharded.

There is no reference counterpart: the upstream builder is single-process.
"""
from __future__ import annotations

import os
import time
from typing import Optional

import numpy as np

U64_ALL_ONES = (1 << 64) - 1
SINGLE, DIRECT = 1 << 63, 1 << 62


def _bit_length(v: int) -> int:
    return max(1, int(v).bit_length())


def max_chunks(ws: int) -> int:
    """how many chunks of an owner's block the plan can size (avl_merge2_max_chunks: its per-chunk tables sit in the LDS histogram)"""
    return max(0, (12288 // (ws * ws) - 3) // 2)


class Plan:
    """what every rank knows after avl_merge2_plan's read-back: M, the growth key, the ws x ws size tables [sender][owner] of the whole
    exchange (A: voxels, Dn: the single-rank ones among them, H[prev][rank]: replay hops) and, when the exchange runs in chunks of
    chunk_rows rows of every owner's block, the same two tables per chunk"""

    def __init__(self, res: np.ndarray, rank: int, ws: int, D: int, chunk_rows: int = 0, nchunk: int = 0):
        W2 = ws * ws
        self.M = int(res[0])
        self.grow_key = int(res[1]) & U64_ALL_ONES
        self.A = res[2:2 + W2].reshape(ws, ws).astype(np.int64)
        self.Dn = res[2 + W2:2 + 2 * W2].reshape(ws, ws).astype(np.int64)
        self.H = res[2 + 2 * W2:2 + 3 * W2].reshape(ws, ws).astype(np.int64)
        self.rank, self.ws, self.D = rank, ws, D
        self.per = max(1, (self.M + ws - 1) // ws)
        self.r0 = min(self.M, rank * self.per)
        self.r1 = min(self.M, self.r0 + self.per)
        if nchunk:
            self.R = int(chunk_rows)
            self.C = max(1, (self.per + self.R - 1) // self.R)
            assert self.C <= nchunk, (self.C, nchunk)
            t = res[2 + 3 * W2:2 + (3 + 2 * nchunk) * W2].reshape(nchunk, 2, ws, ws).astype(np.int64)
            self.Ac, self.Dc = t[:self.C, 0], t[:self.C, 1]
            assert np.array_equal(self.Ac.sum(0), self.A) and np.array_equal(self.Dc.sum(0), self.Dn), "merge2: the chunk tables do not add up"
        else:
            self.R, self.C = self.per, 1
            self.Ac, self.Dc = self.A[None], self.Dn[None]
        self.start = np.concatenate([[0], np.cumsum(self.A[rank])]).astype(np.int64)       # own voxels (final-row order) by destination
        self.dstart = np.concatenate([[0], np.cumsum(self.Dn[rank])]).astype(np.int64)
        self.n_own = self.r1 - self.r0

    def layout(self, c: int) -> "Layout":
        return Layout(self, c)


class Layout:
    """where every list of ONE payload exchange lies (the whole exchange, or chunk c of it): identical arithmetic on every rank"""

    def __init__(self, P: Plan, c: int = 0):
        rank, ws, D = P.rank, P.ws, P.D
        self.P, self.c, self.rank, self.ws, self.D, self.per = P, c, rank, ws, D, P.per
        self.A, self.Dn = P.Ac[c], P.Dc[c]
        self.ldw = (D + 1) // 2                                                # words of a float32 row
        self.own_r0 = P.r0
        self.row_lo = min(P.r1, P.r0 + c * P.R)                                # this rank's rows of the chunk: [row_lo, row_lo + n_rows)
        self.n_rows = max(0, min(P.R, P.r1 - self.row_lo))
        before_a = P.Ac[:c, rank].sum(0) if c else np.zeros(ws, np.int64)
        before_d = P.Dc[:c, rank].sum(0) if c else np.zeros(ws, np.int64)
        self.lo = (P.start[:ws] + before_a).astype(np.int64)                    # first own voxel (final-row order) of the call per destination
        self.dlo = (P.dstart[:ws] + before_d).astype(np.int64)                  # single-rank voxels before it
        self.row0 = (np.arange(ws, dtype=np.int64) * P.per + c * P.R)           # first row of the chunk at every destination
        self.cum = np.concatenate([[0], np.cumsum(self.A[rank])]).astype(np.int64)
        self.n = int(self.cum[ws])

        def seg_words(p, q):
            a, d = int(self.A[p, q]), int(self.Dn[p, q])
            return a * 8 + d * self.ldw + (a - d) * D
        # send buffer: the remote destinations in rank order, this rank's own segment LAST (it never travels)
        self.send_words = np.array([seg_words(rank, q) for q in range(ws)], dtype=np.int64)
        off, o = np.zeros(ws, np.int64), 0
        for q in list(range(rank)) + list(range(rank + 1, ws)) + [rank]:
            off[q] = o
            o += int(self.send_words[q])
        self.send_off = off
        self.send_total = o
        self.remote_words = o - int(self.send_words[rank])
        self.side_off = off.copy()
        self.done_off = off + self.A[rank] * 8
        self.part_off = self.done_off + self.Dn[rank] * self.ldw
        self.recv_words = np.array([seg_words(p, rank) if p != rank else 0 for p in range(ws)], dtype=np.int64)
        self.recv_off = np.concatenate([[0], np.cumsum(self.recv_words)])[:ws].astype(np.int64)
        self.recv_total = int(self.recv_words.sum())

    def peer_lists(self, p):
        """(buffer, side word offset, done word offset, part word offset, records) of what peer p contributes to this rank's rows"""
        a, d = int(self.A[p, self.rank]), int(self.Dn[p, self.rank])
        base = int(self.send_off[p]) if p == self.rank else int(self.recv_off[p])
        return ("send" if p == self.rank else "recv"), base, base + a * 8, base + a * 8 + d * self.ldw, a

    def in_splits(self):
        return [0 if q == self.rank else int(self.send_words[q]) for q in range(self.ws)]

    def out_splits(self):
        return [int(w) for w in self.recv_words]


# ----------------------------------------------------------------------------------------------------------------------------
# NumPy twin of csrc/avl_merge2.hip + avl_builder_m2_pack (CPU tensors: the gloo tests; the GPU tests compare the kernels with it)
# ----------------------------------------------------------------------------------------------------------------------------
class HostKernels:
    def __init__(self, raw, replay_fn=None):
        """raw: cell (n,) i32, first_key (n,) i64, sum_feat (n, D) f64, sum_w4 (n, 4) f64, first_feat (n, D) f32, first_alpha (n,) f64
        (torch CPU tensors or arrays: avl_builder_export_raw's lists); replay_fn(state (n, 3) int64 array, sel (n,) bool)"""
        a = lambda x: np.ascontiguousarray(x.numpy() if hasattr(x, "numpy") else x)
        self.raw = {k: a(v) for k, v in raw.items()}
        self.n = int(self.raw["cell"].shape[0])
        self.perm = np.argsort(self.raw["first_key"], kind="stable").astype(np.int32)      # avl_merge2_prepare: the list goes out in key order
        self.D = int(self.raw["sum_feat"].shape[1])
        self.replay_fn = replay_fn
        self.device = "cpu"

    def header(self, flags):
        import torch
        k = self.raw["first_key"]
        return torch.tensor([self.n, int(k.min()) if self.n else np.iinfo(np.int64).max, int(k.max()) if self.n else -1, flags], dtype=torch.int64)

    def new_words(self, words):
        import torch
        return torch.zeros(int(words), dtype=torch.int64)

    def fill_chunk(self, chunk, nmax):
        c = chunk.numpy()
        c[:self.n] = self.raw["first_key"][self.perm]
        c[nmax:].view(np.int32)[:self.n] = self.raw["cell"][self.perm]

    def plan(self, gathered, n_all, nmax, rank, ws, cell_bits, grow_row, want_lists, chunk_rows=0, nchunk=0):
        g = gathered.numpy()
        stride = nmax + (nmax + 1) // 2
        off = np.concatenate([[0], np.cumsum(n_all)]).astype(np.int64)
        E = int(off[-1])
        ecell = np.concatenate([g[p * stride + nmax:(p + 1) * stride].view(np.int32)[:n_all[p]] for p in range(ws)]) if E else np.zeros(0, np.int32)
        ekey = np.concatenate([g[p * stride:p * stride + n_all[p]] for p in range(ws)]) if E else np.zeros(0, np.int64)
        erank = np.repeat(np.arange(ws), n_all)
        se = np.argsort(ecell, kind="stable")                           # contributors of a cell stay in rank order
        scell = ecell[se]
        head = np.ones(E, bool)
        head[1:] = scell[1:] != scell[:-1]
        tail = np.ones(E, bool)
        tail[:-1] = head[1:]
        hp = np.maximum.accumulate(np.where(head, np.arange(E), 0)) if E else np.zeros(0, np.int64)
        rs = erank[se]
        prev = np.where(head, -1, np.roll(rs, 1)) if E else rs
        nxt = np.where(tail, -1, np.roll(rs, -1)) if E else rs
        # the ranks' lists are in key order and the keys ordered by rank: ENTRY order is key order, a first contributor's row is the
        # number of first contributors before it
        headflag = np.zeros(E, np.int64)
        headflag[se] = head
        rowscan = np.cumsum(headflag) - headflag
        M = int(head.sum())
        row = rowscan[se[hp]] if E else rowscan
        self.rowcell = np.zeros(M, np.int32)
        self.rowcell[row[head]] = scell[head]
        keyrow = np.zeros(M, np.int64)
        keyrow[row[head]] = ekey[se][head]
        per = max(1, (M + ws - 1) // ws)
        q = np.minimum(row // per, ws - 1)
        res = np.zeros(2 + (3 + 2 * nchunk) * ws * ws, np.int64)
        res[0] = M
        res[1] = int(keyrow[grow_row]) if 0 <= grow_row < M else -1
        W2 = ws * ws
        np.add.at(res, 2 + rs * ws + q, 1)
        single_e = (prev < 0) & (nxt < 0)
        np.add.at(res, 2 + W2 + (rs * ws + q)[single_e], 1)
        np.add.at(res, 2 + 2 * W2 + (prev * ws + rs)[prev >= 0], 1)
        if nchunk:
            c = np.minimum((row - q * per) // chunk_rows, nchunk - 1)
            np.add.at(res, 2 + (3 + 2 * c) * W2 + rs * ws + q, 1)
            np.add.at(res, 2 + (4 + 2 * c[single_e]) * W2 + (rs * ws + q)[single_e], 1)
        mine = rs == rank
        s = self.perm[se[mine] - off[rank]]
        n = self.n
        self.row, self.prev, self.next = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32)
        self.row[s], self.prev[s], self.next[s] = row[mine], prev[mine], nxt[mine]
        self.order = np.argsort(self.row, kind="stable").astype(np.int32)
        single = ((self.prev < 0) & (self.next < 0))[self.order]
        self.sidx = (np.cumsum(single) - single).astype(np.int32)
        ar = np.arange(n, dtype=np.int64)
        self.selA, self.selB = np.where(self.prev < 0, ar, -1), np.where(self.prev < 0, -1, ar)
        kp = np.where(self.prev[self.order] < 0, ws, self.prev[self.order])
        kn = np.where(self.next[self.order] < 0, ws, self.next[self.order])
        self.idx_prev = self.order[np.argsort(kp, kind="stable")]
        self.idx_next = self.order[np.argsort(kn, kind="stable")]
        return res

    def new_state(self):
        return np.zeros((self.n, 3), np.int64)

    def replay(self, phase, state, grow_key):
        sel = (self.selA if phase == "A" else self.selB) >= 0
        if self.replay_fn is not None and sel.any():
            self.replay_fn(state, sel)

    def state_gather(self, state, lo, hi):
        import torch
        return torch.from_numpy(np.ascontiguousarray(state[self.idx_next[lo:hi]]).reshape(-1))

    def state_scatter(self, state, buf, lo, hi):
        state[self.idx_prev[lo:hi]] = buf.numpy().reshape(-1, 3)

    def pack(self, send, L, own_feat):
        w = send.numpy()
        r, D = self.raw, self.D
        a1 = r["first_alpha"]
        for q in range(L.ws):
            i0 = int(L.lo[q])
            i1 = i0 + int(L.A[L.rank, q])
            if i1 == i0:
                continue
            sl = self.order[i0:i1]
            is_new = self.prev[sl] < 0
            single = is_new & (self.next[sl] < 0)
            didx = self.sidx[i0:i1].astype(np.int64) - int(L.dlo[q])
            pidx = np.arange(i1 - i0) - didx
            row = self.row[sl].astype(np.int64)
            row_rel = row - int(L.row0[q])
            direct = single & (q == L.rank) & (own_feat is not None)
            side = w[int(L.side_off[q]):int(L.side_off[q]) + 8 * (i1 - i0)].reshape(-1, 8)
            word = row_rel | (np.where(single, didx, pidx) << 32)
            word = word.astype(np.uint64) | np.where(single, np.uint64(SINGLE), np.uint64(0)) | np.where(direct, np.uint64(DIRECT), np.uint64(0))
            side[:, 0] = word.view(np.int64)
            side[:, 1:5] = r["sum_w4"][sl].view(np.int64)
            side[:, 5:8] = 0
            fin = ((a1[sl] ** 2)[:, None] * r["first_feat"][sl].astype(np.float64) + r["sum_feat"][sl]) / r["sum_w4"][sl][:, :1]
            nd = int(L.Dn[L.rank, q])
            done = w[int(L.done_off[q]):int(L.done_off[q]) + nd * L.ldw].view(np.float32).reshape(nd, 2 * L.ldw)
            if own_feat is not None and q == L.rank:
                own_feat[(row - L.own_r0)[direct]] = fin[direct].astype(np.float32)
                keep = single & ~direct
            else:
                keep = single
            done[didx[keep], :D] = fin[keep].astype(np.float32)
            npart = (i1 - i0) - nd
            part = w[int(L.part_off[q]):int(L.part_off[q]) + npart * D].view(np.float64).reshape(npart, D)
            sh = ~single
            wf = np.where(is_new, a1[sl] * a1[sl], a1[sl])
            part[pidx[sh]] = wf[sh][:, None] * r["first_feat"][sl][sh].astype(np.float64) + r["sum_feat"][sl][sh]

    def side_state(self, send, L, state):
        w = send.numpy()
        for q in range(L.ws):
            i0 = int(L.lo[q])
            i1 = i0 + int(L.A[L.rank, q])
            if i1 == i0:
                continue
            sl = self.order[i0:i1]
            side = w[int(L.side_off[q]):int(L.side_off[q]) + 8 * (i1 - i0)].reshape(-1, 8)
            last = (self.next[sl] < 0) if state is not None else np.zeros(i1 - i0, bool)
            side[:, 5:8] = np.where(last[:, None], state[sl] if state is not None else 0, 0)

    def new_block(self, n_own, D, own_feat):
        return dict(grid_feat=own_feat, grid_pos=np.zeros((n_own, 3), np.int32), weight=np.zeros(n_own, np.float32),
                    grid_rgb=np.zeros((n_own, 3), np.uint8), cell=np.zeros(n_own, np.int32),
                    w4=np.zeros((n_own, 4)), state=np.zeros((n_own, 3), np.int64), part_rows=[], part_acc=[])       # (the twin's intermediates: tests)

    def fold(self, send, recv, L, gs, vh, have_log, out):
        """rows [L.row_lo, L.row_lo + L.n_rows) of the rank's block from one exchange; also keeps the twin's intermediate sums for the tests"""
        n_own, D = L.n_rows, L.D
        if n_own == 0:
            return
        b0 = L.row_lo - L.own_r0
        bufs = dict(send=send.numpy(), recv=recv.numpy() if recv is not None else None)
        w4 = np.zeros((n_own, 4))
        acc = np.zeros((n_own, D))
        ncontrib = np.zeros(n_own, np.int64)
        state = np.zeros((n_own, 3), np.int64)
        feat = out["grid_feat"][b0:b0 + n_own]
        single_row = np.zeros(n_own, bool)
        seen = np.zeros(n_own, bool)
        for p in range(L.ws):                                            # rank order: reproducible float64 sums
            which, s_off, d_off, p_off, cnt = L.peer_lists(p)
            if cnt == 0:
                continue
            w = bufs[which]
            side = w[s_off:s_off + 8 * cnt].reshape(cnt, 8)
            word = side[:, 0].view(np.uint64)
            rows = (word & np.uint64(0xFFFFFFFF)).astype(np.int64)
            assert (rows >= 0).all() and (rows < n_own).all() and (np.diff(rows) > 0).all(), "merge2: a peer's rows are not inside the chunk"
            fidx = ((word >> np.uint64(32)) & np.uint64(0x3FFFFFFF)).astype(np.int64)
            sg = (word & np.uint64(SINGLE)) != 0
            dr = (word & np.uint64(DIRECT)) != 0
            w4[rows] += side[:, 1:5].view(np.float64)
            ncontrib[rows] += 1
            seen[rows] = True
            started = (side[:, 7].view(np.uint64) >> np.uint64(32)) != 0
            state[rows[started]] = side[started, 5:8]
            nd = int(L.Dn[p, L.rank])
            done = w[d_off:d_off + nd * L.ldw].view(np.float32).reshape(nd, 2 * L.ldw)
            cp = sg & ~dr
            feat[rows[cp]] = done[fidx[cp], :D]
            single_row[rows[sg]] = True
            part = w[p_off:p_off + (cnt - nd) * D].view(np.float64).reshape(cnt - nd, D)
            acc[rows[~sg]] += part[fidx[~sg]]
        assert seen.all(), "merge2: a row of the block nobody sent"
        assert not (single_row & (ncontrib > 1)).any()
        sh = ~single_row
        feat[sh] = (acc[sh] / w4[sh, :1]).astype(np.float32)
        cell = self.rowcell[L.row_lo:L.row_lo + n_own]
        out["cell"][b0:b0 + n_own] = cell
        out["grid_pos"][b0:b0 + n_own] = np.stack([cell // (gs * vh), (cell // vh) % gs, cell % vh], 1).astype(np.int32)
        out["weight"][b0:b0 + n_own] = w4[:, 0].astype(np.float32)
        out["grid_rgb"][b0:b0 + n_own] = np.clip(w4[:, 1:4] / w4[:, :1] + 1e-9, 0, 255).astype(np.uint8)
        out["w4"][b0:b0 + n_own] = w4
        out["state"][b0:b0 + n_own] = state
        out["part_rows"].append(np.nonzero(sh)[0] + b0)
        out["part_acc"].append(acc[sh])

    def finish_block(self, out):
        out["part_rows"] = np.concatenate(out["part_rows"]) if out["part_rows"] else np.zeros(0, np.int64)
        out["part_acc"] = np.concatenate(out["part_acc"]) if out["part_acc"] else np.zeros((0, self.D))
        return out


# ----------------------------------------------------------------------------------------------------------------------------
# the HIP kernels on the builder's own device arrays (the product path)
# ----------------------------------------------------------------------------------------------------------------------------
class HipKernels:
    def __init__(self, acc, n):
        import torch
        from . import _lib
        from .device import torch_stream_ptr
        self.acc, self.n, self.D = acc, int(n), int(acc.D)
        self.lib, self.check = _lib.load(), _lib.check
        self.st = torch_stream_ptr()
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.torch = torch

    def header(self, flags):
        """the rank's (key, cell) list sorted by key (avl_merge2_prepare) + its header; fill_chunk copies from there"""
        import ctypes as C
        t = self.torch
        m = max(self.n, 1)
        key = t.empty(m, dtype=t.int64, device=self.device)
        cell = t.empty(m, dtype=t.int32, device=self.device)
        self._key = t.empty(m, dtype=t.int64, device=self.device)
        self._cell = t.empty(m, dtype=t.int32, device=self.device)
        self.perm = t.empty(m, dtype=t.int32, device=self.device)
        if self.n:
            self.check(self.lib.avl_builder_export_raw(self.acc._h, self.n, cell.data_ptr(), key.data_ptr(), None, None, None, None, self.st),
                       "avl_builder_export_raw")
        nb = C.c_size_t()
        self.check(self.lib.avl_merge2_prepare_work_bytes(self.n, C.byref(nb)), "avl_merge2_prepare_work_bytes")
        work = t.empty(int(nb.value), dtype=t.uint8, device=self.device)
        hdr = t.empty(4, dtype=t.int64, device=self.device)
        self.check(self.lib.avl_merge2_prepare(self.n, key.data_ptr(), cell.data_ptr(), int(self.acc.key_bits()), int(flags), self._key.data_ptr(),
                                               self._cell.data_ptr(), self.perm.data_ptr(), hdr.data_ptr(), work.data_ptr(), int(nb.value), self.st),
                   "avl_merge2_prepare")
        return hdr

    def new_words(self, words):
        return self.torch.empty(max(int(words), 1), dtype=self.torch.int64, device=self.device)

    def fill_chunk(self, chunk, nmax):
        if self.n:
            chunk[:self.n].copy_(self._key[:self.n])
            chunk[nmax:].view(self.torch.int32)[:self.n].copy_(self._cell[:self.n])
        del self._key, self._cell

    def plan(self, gathered, n_all, nmax, rank, ws, cell_bits, grow_row, want_lists, chunk_rows=0, nchunk=0):
        import ctypes as C
        t = self.torch
        E = int(sum(n_all))
        nb = C.c_size_t()
        self.check(self.lib.avl_merge2_work_bytes(E, self.n, ws, int(nchunk), C.byref(nb)), "avl_merge2_work_bytes")
        self.work = t.empty(int(nb.value), dtype=t.uint8, device=self.device)
        h_n = (C.c_int64 * ws)(*[int(v) for v in n_all])
        h_off = (C.c_int64 * 11)()
        nres = 2 + (3 + 2 * int(nchunk)) * ws * ws
        h_res = (C.c_int64 * nres)()
        self.check(self.lib.avl_merge2_plan(ws, rank, h_n, int(nmax), gathered.data_ptr(), self.perm.data_ptr(), int(cell_bits), int(grow_row),
                                            1 if want_lists else 0, int(chunk_rows), int(nchunk), self.work.data_ptr(), int(nb.value), h_off, h_res,
                                            self.st), "avl_merge2_plan")
        base = self.work.data_ptr()
        names = ("row", "prev", "next", "order", "sidx", "selA", "selB", "idx_prev", "idx_next", "rowcell", "res")
        self.p = {k: base + int(h_off[i]) for i, k in enumerate(names)}
        return np.frombuffer(h_res, dtype=np.int64, count=nres).copy()

    def new_state(self):
        return self.torch.zeros((max(self.n, 1), 3), dtype=self.torch.int64, device=self.device)

    def replay(self, phase, state, grow_key):
        if self.n:
            self.check(self.lib.avl_builder_replay_chain(self.acc._h, self.n, self.p["selA" if phase == "A" else "selB"], int(grow_key), state.data_ptr(),
                                                         self.st), "avl_builder_replay_chain")

    def state_gather(self, state, lo, hi):
        out = self.torch.empty(3 * (hi - lo), dtype=self.torch.int64, device=self.device)
        self.check(self.lib.avl_merge2_state_gather(hi - lo, self.p["idx_next"] + 4 * lo, state.data_ptr(), out.data_ptr(), self.st), "avl_merge2_state_gather")
        return out

    def state_scatter(self, state, buf, lo, hi):
        buf = buf.contiguous()
        self.check(self.lib.avl_merge2_state_scatter(hi - lo, self.p["idx_prev"] + 4 * lo, buf.data_ptr(), state.data_ptr(), self.st), "avl_merge2_state_scatter")

    @staticmethod
    def _i64s(a):
        import ctypes as C
        return (C.c_int64 * len(a))(*[int(v) for v in a])

    def pack(self, send, L, own_feat):
        p = self.p
        self.check(self.lib.avl_builder_m2_pack(self.acc._h, L.n, L.ws, L.rank, L.own_r0, self._i64s(L.cum), self._i64s(L.lo), self._i64s(L.dlo),
                                                self._i64s(L.row0), self._i64s(L.side_off), self._i64s(L.done_off), self._i64s(L.part_off),
                                                p["order"], p["row"], p["prev"], p["next"], p["sidx"], send.data_ptr(),
                                                own_feat.data_ptr() if own_feat is not None else None, self.st), "avl_builder_m2_pack")

    def side_state(self, send, L, state):
        self.check(self.lib.avl_merge2_side_state(L.n, L.ws, self._i64s(L.cum), self._i64s(L.lo), self._i64s(L.side_off), self.p["order"], self.p["next"],
                                                  state.data_ptr() if state is not None else None, send.data_ptr(), self.st), "avl_merge2_side_state")

    def new_block(self, n_own, D, own_feat):
        t = self.torch
        self.err = t.zeros(1, dtype=t.int32, device=self.device)
        return dict(grid_feat=own_feat, grid_pos=t.empty((n_own, 3), dtype=t.int32, device=self.device),
                    weight=t.empty((n_own,), dtype=t.float32, device=self.device), grid_rgb=t.empty((n_own, 3), dtype=t.uint8, device=self.device),
                    cell=t.empty((n_own,), dtype=t.int32, device=self.device))

    def fold(self, send, recv, L, gs, vh, have_log, out):
        """rows [L.row_lo, L.row_lo + L.n_rows) of the rank's block from one exchange"""
        import ctypes as C
        t = self.torch
        n, D = L.n_rows, L.D
        if n == 0:
            return
        nb = C.c_size_t()
        self.check(self.lib.avl_merge2_fold_work_bytes(n, L.ws, C.byref(nb)), "avl_merge2_fold_work_bytes")
        if getattr(self, "_table", None) is None or self._table.numel() < int(nb.value):
            self._table = t.empty(int(nb.value), dtype=t.uint8, device=self.device)          # (torch's blocks are 512-byte aligned)
        ptr = dict(send=send.data_ptr(), recv=recv.data_ptr() if recv is not None else 0)
        side, done, part, cnt = [], [], [], []
        for p in range(L.ws):
            which, s_off, d_off, p_off, c = L.peer_lists(p)
            side.append(ptr[which] + 8 * s_off)
            done.append(ptr[which] + 8 * d_off)
            part.append(ptr[which] + 8 * p_off)
            cnt.append(c)
        vps = lambda a: (C.c_void_p * len(a))(*a)
        b0 = L.row_lo - L.own_r0
        self.check(self.lib.avl_merge2_fold(n, L.row_lo, L.ws, D, gs, vh, vps(side), vps(done), vps(part), self._i64s(cnt), self.p["rowcell"],
                                            1 if have_log else 0, out["grid_feat"].data_ptr() + 4 * D * b0, out["grid_pos"].data_ptr() + 12 * b0,
                                            out["weight"].data_ptr() + 4 * b0, out["grid_rgb"].data_ptr() + 3 * b0, out["cell"].data_ptr() + 4 * b0,
                                            self._table.data_ptr(), int(self._table.numel()), self.err.data_ptr(), self.st), "avl_merge2_fold")

    def finish_block(self, out):
        self._table = None
        return out


# ----------------------------------------------------------------------------------------------------------------------------
def merge_sharded_v2(K, coll, D, cell_bits, grow_row, gs, vh, have_log, status=0, timings: Optional[dict] = None, sync=None):
    """The choreography, shared by the device path (K = HipKernels) and its twin (K = HostKernels).  coll: parallel._Coll or None
    (one process).  Returns None when the keys are not ordered by rank (the caller falls back to the general plan), else
    (out dict, Plan, info dict)."""
    import torch
    rank, ws = (coll.rank, coll.ws) if coll is not None else (0, 1)
    sync = sync or (lambda: None)
    marks = []
    trace = timings is not None and os.environ.get("AVLMAPS_MERGE_TRACE") == "1"

    def mark(label):
        # phase boundaries are HOST clocks; the device is only drained here when a trace is asked for (AVLMAPS_MERGE_TRACE=1) -- otherwise a
        # phase's kernels run on while the host prepares the next one (pack under the replay's set-up, ...), as in a build that is not
        # being timed; every collective drains the device before it starts its own clock, so a rank's total compute stays exact
        if trace:
            sync()
        marks.append((label, time.perf_counter(), coll.comm_s if coll is not None else 0.0, coll.gpu_lock.wait_s if coll is not None else 0.0))
    mark("start")
    hdr = K.header((1 if have_log else 0) | (int(status) << 1))
    allh = torch.stack(coll.all_gather(hdr)).cpu().numpy() if coll is not None else hdr.cpu().numpy()[None]
    bad = [r for r in range(ws) if int(allh[r, 3]) >> 1]
    if bad:
        raise RuntimeError(f"multi-rank merge aborted: rank(s) {bad} reported a failure (status {[int(allh[r, 3]) >> 1 for r in bad]})")
    have_log = bool(min(int(allh[r, 3]) & 1 for r in range(ws)))
    n_all = [int(v) for v in allh[:, 0]]
    last = -1
    for r in range(ws):
        if n_all[r] > 0:
            if int(allh[r, 1]) <= last:
                return None                                             # keys not ordered by rank: general plan
            last = int(allh[r, 2])
    nmax = max(1, max(n_all))
    stride = nmax + (nmax + 1) // 2
    gathered = K.new_words(ws * stride)
    chunk = gathered[rank * stride:(rank + 1) * stride]
    K.fill_chunk(chunk, nmax)
    if trace:
        mark("plan: header + lists out")
    if coll is not None:
        coll.all_gather_into(gathered, chunk)
    # the payload goes in chunks of R rows of every owner's block once a block is larger than that: export -> all_to_all -> fold of
    # neighbouring chunks overlap (nccl: the exchange runs on the backend's stream) and the buffers are O(chunk), not O(local voxels).
    # R: AVLMAPS_MERGE_CHUNK_MB (default 1024) of float64 payload per owner and chunk -- a chunk costs a rank ~0.25 ms of launches and
    # hand-overs (profiles/r06_merge_chunks_probe.txt: 1 / 3 / 5 / 18 chunks at the bench's 8-rank merge), ~2 % of what it spends on the wire
    per_max = max(1, -(-sum(n_all) // ws))
    R = int(os.environ.get("AVLMAPS_MERGE_CHUNK_ROWS", "0")) or max(1024, (int(os.environ.get("AVLMAPS_MERGE_CHUNK_MB", "1024")) << 20) // (8 * D + 64))
    nchunk = 0
    if coll is not None and ws > 1 and 0 < R < per_max and max_chunks(ws) >= 2:
        nchunk = -(-per_max // R)
        if nchunk > max_chunks(ws):
            R = -(-per_max // max_chunks(ws))
            nchunk = -(-per_max // R)
    res = K.plan(gathered, n_all, nmax, rank, ws, cell_bits, grow_row, have_log and ws > 1, R, nchunk)
    if trace:
        mark("plan: kernels + read-back")
    P = Plan(res, rank, ws, D, R, nchunk)
    Ls = [P.layout(c) for c in range(P.C)]
    del gathered, chunk
    mark("plan")
    n_own = P.n_own
    if K.device != "cpu":
        own_feat = torch.empty((n_own, D), dtype=torch.float32, device=K.device)
    else:
        own_feat = np.zeros((n_own, D), np.float32)
    out = K.new_block(n_own, D, own_feat)
    words = max(L.send_total for L in Ls)
    sends = [K.new_words(words) for _ in range(min(2, P.C))]
    K.pack(sends[0], Ls[0], own_feat)
    mark("pack")
    # ---- exact sequential weight / colour: at once where no lower rank holds the voxel, hop by hop where ranks share it
    state = None
    chain_bytes = 0
    if have_log:
        state = K.new_state()
        K.replay("A", state, P.grow_key)
        if ws > 1:
            po = np.concatenate([[0], np.cumsum(P.H[:, rank])]).astype(np.int64)          # my voxels grouped by prev rank
            no = np.concatenate([[0], np.cumsum(P.H[rank, :])]).astype(np.int64)          # ... by next rank
            for q in range(1, ws):
                if not P.H[:, q].any():
                    continue                                            # nobody shares a voxel with q's predecessors: every rank skips the round
                k_out = int(P.H[rank, q]) if rank < q else 0
                hop = K.state_gather(state, int(no[q]), int(no[q]) + k_out) if k_out else torch.zeros(0, dtype=torch.int64, device=sends[0].device)
                ins = [0] * ws
                ins[q] = 3 * k_out
                outs = [3 * int(P.H[p, q]) if rank == q else 0 for p in range(ws)]
                got = coll.all_to_all(hop, ins, outs)
                chain_bytes += 24 * k_out
                if rank == q and got.numel():
                    K.state_scatter(state, got, 0, int(po[ws]))
                    K.replay("B", state, P.grow_key)
    K.side_state(sends[0], Ls[0], state)
    mark("replay")
    exchanging = coll is not None and ws > 1
    if P.C == 1:
        recv = coll.all_to_all(sends[0][:Ls[0].remote_words], Ls[0].in_splits(), Ls[0].out_splits()) if exchanging else None
        mark("exchange")
        K.fold(sends[0], recv, Ls[0], gs, vh, have_log, out)
    else:
        # chunk c + 1 is packed while chunk c travels and chunk c - 1 folds; a send buffer is reused two chunks later, after its
        # exchange was finished and its fold issued on this stream
        ex = [0.0, 0.0, 0.0]                                             # wall, in collectives, waiting for a shared GPU: of the exchange calls

        def clocked(fn, *a):
            t0, c0, l0 = time.perf_counter(), coll.comm_s, coll.gpu_lock.wait_s
            r = fn(*a)
            ex[0] += time.perf_counter() - t0
            ex[1] += coll.comm_s - c0
            ex[2] += coll.gpu_lock.wait_s - l0
            return r

        def start(c):
            return coll.all_to_all_start(sends[c % 2][:Ls[c].remote_words], Ls[c].in_splits(), Ls[c].out_splits(), overlap=not trace)
        loop0 = marks[-1]
        h = clocked(start, 0)
        for c in range(P.C):
            if c + 1 < P.C:
                K.pack(sends[(c + 1) % 2], Ls[c + 1], own_feat)
                K.side_state(sends[(c + 1) % 2], Ls[c + 1], state)
            recv = clocked(coll.all_to_all_finish, h)
            if c + 1 < P.C:
                h = clocked(start, c + 1)
            K.fold(sends[c % 2], recv, Ls[c], gs, vh, have_log, out)
        del h
        # the exchange calls' share of the loop, booked as the "exchange" phase (the rest of the loop is the fold's)
        marks.append(("exchange", loop0[1] + ex[0], loop0[2] + ex[1], loop0[3] + ex[2]))
    out = K.finish_block(out)
    if trace:
        mark("fold: kernels")
    err = getattr(K, "err", None)
    if err is not None:
        if coll is not None:
            err = coll.all_reduce(err, coll.dist.ReduceOp.MAX)      # every rank raises together (nobody is left inside a later collective)
        if int(err.item()):
            raise RuntimeError(f"multi-rank merge: the fold found an inconsistent exchange (flags {int(err.item())})")
    if trace:
        mark("fold: flags")
    del sends, recv
    if timings is not None and not trace:
        sync()
    mark("fold")
    if trace:
        import sys
        steps = [(b[0], 1e3 * ((b[1] - a[1]) - (b[2] - a[2]) - (b[3] - a[3]))) for a, b in zip(marks[:-1], marks[1:])]
        print(f"[merge2 trace] rank {rank}: " + " | ".join(f"{k} {v:.2f}" for k, v in steps) + " (own ms)", file=sys.stderr, flush=True)
        marks[:] = [m for m in marks if ":" not in m[0]]
    info = dict(marks=marks, n_all=n_all, have_log=have_log, chain_bytes=chain_bytes, n_own=n_own, chunks=P.C, chunk_rows=P.R,
                remote_words=sum(int(L.remote_words) for L in Ls), buffer_words=words * len(Ls[:2]))
    return out, P, info


def _phase_times(marks):
    wall, comm, lockw = {}, {}, {}
    for (_, t0, c0, l0), (label, t1, c1, l1) in zip(marks[:-1], marks[1:]):
        lockw[label] = l1 - l0
        wall[label] = (t1 - t0) - lockw[label]
        comm[label] = c1 - c0
    return wall, comm, lockw


def merge_accumulator_v2(acc, group, exact_rgb, timings, gather_to, status, glock):
    """merge_accumulator_sharded's body in the gather-plan form (see the module docstring); same result dict.  Returns None when the
    first-touch keys are not ordered by rank (the caller takes the general plan)."""
    import torch
    from . import parallel
    dev = torch.device("cuda", torch.cuda.current_device())
    t0 = time.perf_counter()
    lw0 = glock.wait_s
    coll = parallel._Coll(group) if parallel._dist_on(group) else None
    from .device import torch_stream_ptr
    n = acc.num_voxels(torch_stream_ptr()) if not status else 0
    K = HipKernels(acc, n)
    have_log = bool(exact_rgb and acc.has_replay_log())
    ncell = acc.n_rows * acc.gs * acc.vh
    r = merge_sharded_v2(K, coll, acc.D, _bit_length(ncell - 1), acc.n_rows * acc.gs - 1, acc.gs, acc.vh, have_log, status, timings,
                         sync=torch.cuda.synchronize)
    if r is None:
        return None
    out, L, info = r
    rank, ws = L.rank, L.ws
    res = dict(M=L.M, rows=(L.r0, L.r1), cell=out["cell"], grid_feat=out["grid_feat"], grid_pos=out["grid_pos"], weight=out["weight"],
               grid_rgb=out["grid_rgb"])
    torch.cuda.synchronize()
    t5 = time.perf_counter()
    c5 = coll.comm_s if coll is not None else 0.0
    lw5 = glock.wait_s
    gather_bytes = 0
    if gather_to is not None:
        full = parallel.gather_row_shards(res, gather_to, coll, rank, ws, names=("grid_feat", "grid_pos", "weight", "grid_rgb", "cell"))
        if rank != gather_to:
            gather_bytes = info["n_own"] * (acc.D * 4 + 12 + 4 + 3 + 4)
        if full is not None:
            full["occupied_ids"] = parallel.occupied_ids_from_cells(full.pop("cell"), acc.n_rows, acc.gs, acc.vh)
            res["full"] = full
    torch.cuda.synchronize()
    t6 = time.perf_counter()
    if timings is not None:
        wall, comm, lockw = _phase_times(info["marks"])
        t_first = info["marks"][0][1]
        wall["plan"] += t_first - t0 - (info["marks"][0][3] - lw0)           # num_voxels + set-up belong to the plan
        wall["gather"] = (t6 - t5) - (glock.wait_s - lw5)
        comm["gather"] = (coll.comm_s if coll is not None else 0.0) - c5
        lockw["gather"] = glock.wait_s - lw5
        D, W = acc.D, acc.D + 4
        A, Dn = L.A, L.Dn
        sent_all = int(A[rank].sum() - A[rank, rank])
        sent_done = int(Dn[rank].sum() - Dn[rank, rank])
        payload = 8 * int(info["remote_words"])
        plan_bytes = (12 * int(max(info["n_all"])) * (ws - 1) + 32 * (ws - 1)) if coll is not None else 0
        names = dict(plan="plan", pack="export", replay="replay_chain", exchange="exchange", fold="fold_finalize", gather="gather")
        wall_s = {names[k]: v for k, v in wall.items()}
        comm_s = {names[k]: v for k, v in comm.items()}
        timings.update(phase_clock="host clocks at the phase boundaries; the device is drained only by the collectives and at the end (per phase with "
                                   "AVLMAPS_MERGE_TRACE=1): a phase's kernels may finish under the next phase's host work, the total is exact",
                       mode="row-sharded all_to_all", plan="gather plan (two all_gathers, one radix-sorted union per rank; mixed float32 / float64 payload in ONE "
                       "all_to_all; replay hops as small all_to_alls)",
                       plan_s=wall_s["plan"], scatter_s=wall_s["export"], replay_chain_s=wall_s["replay_chain"], exchange_s=wall_s["exchange"],
                       accumulate_s=wall_s["fold_finalize"], finalize_s=0.0, gather_s=wall_s["gather"], null_launch_us=None, wall_s=wall_s,
                       in_collectives_s=comm_s, shared_gpu_wait_s=sum(lockw.values()),
                       compute_s={k: wall_s[k] - comm_s[k] for k in wall_s},
                       compute_total_s=sum(wall_s[k] - comm_s[k] for k in wall_s if k != "gather"),
                       in_collectives_total_s=sum(comm_s[k] for k in comm_s if k != "gather"),
                       merged_voxels=L.M, local_voxels=n, own_rows=info["n_own"], new_voxels=None, single_rank_voxels=int(Dn[rank].sum()),
                       shared_voxels_local=int(A[rank].sum() - Dn[rank].sum()), shared_rows_owned=None, directory_entries=int(sum(info["n_all"])),
                       rows_sent=sent_all, payload_bytes_sent=payload, payload_bytes_fp64_form=sent_all * (W * 8 + 8),
                       plan_bytes_sent=plan_bytes, chain_bytes_sent=info["chain_bytes"], gather_bytes_sent=gather_bytes,
                       bytes_sent_per_rank=payload + plan_bytes + info["chain_bytes"] + gather_bytes,
                       local_row_bytes=n * W * 8, dense_reduce_payload_bytes=L.M * W * 8, exact_rgb=bool(info["have_log"]),
                       collectives=(coll.calls if coll else 0), world_size=ws, backend=(coll.dist.get_backend(coll.group) if coll is not None else "none"),
                       rows_sent_single=sent_done, exchange_chunks=info["chunks"], exchange_chunk_rows=info["chunk_rows"],
                       exchange_buffer_bytes=8 * info["buffer_words"])
    return res


def merge_raw_sharded_v2(raw, group=None, replay_fn=None, gs2: Optional[int] = None, gs: int = 1 << 10, vh: int = 1 << 10, ncell: Optional[int] = None,
                         coll=None):
    """The same merge on exported raw accumulators (CPU tensors; gloo): the NumPy twin of the kernels under the same choreography.
    Returns the rank's block as torch tensors + the twin's intermediates, or None if the keys are not ordered by rank."""
    import torch
    from . import parallel
    if coll is None:                  # (tests pass an in-process stand-in with _Coll's interface)
        coll = parallel._Coll(group) if parallel._dist_on(group) else None
    K = HostKernels(raw, replay_fn)
    ncell = int(ncell or (1 << 31) - 1)
    r = merge_sharded_v2(K, coll, K.D, _bit_length(ncell - 1), (gs2 - 1) if gs2 else -1, gs, vh, replay_fn is not None)
    if r is None:
        return None
    out, L, info = r
    rank = L.rank
    sent_all = int(L.A[rank].sum() - L.A[rank, rank])
    res = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in out.items()}
    res.update(M=L.M, rows=(L.r0, L.r1), grow_key=L.grow_key, bytes_sent=8 * int(info["remote_words"]), plan="gather",
               payload_bytes_fp64_form=sent_all * ((K.D + 4) * 8 + 8), layout=L, chain_bytes=info["chain_bytes"], chunks=info["chunks"])
    return res
