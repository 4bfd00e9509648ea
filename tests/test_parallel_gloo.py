"""world_size-2 gloo tests (CPU) of the multi-GPU choreography: frame sharding + sparse merge + row-sharded top-1."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from avlmaps_amd import parallel  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def make_rank_raw(seed, D, cells, frame_lo):
    """synthetic raw export of one rank: voxels `cells`, keys inside this rank's frame block"""
    rng = np.random.default_rng(seed)
    n = len(cells)
    keys = ((frame_lo + rng.integers(0, 50, n)).astype(np.int64) << 32) | rng.permutation(n).astype(np.int64)
    order = np.argsort(keys)
    return dict(cell=torch.from_numpy(np.asarray(cells, np.int32)[order].copy()),
                first_key=torch.from_numpy(keys[order].copy()),
                sum_feat=torch.from_numpy(rng.standard_normal((n, D))), sum_w4=torch.from_numpy(rng.random((n, 4)) + 0.1),
                first_feat=torch.from_numpy(rng.standard_normal((n, D)).astype(np.float32)),
                first_alpha=torch.from_numpy(rng.random(n)))


def expected_merge(raws):
    """per cell: summed accumulators, the smallest first-touch key and ITS first-touch feature / alpha"""
    D = raws[0]["sum_feat"].shape[1]
    table = {}
    for r in raws:
        for i, c in enumerate(r["cell"].tolist()):
            e = table.setdefault(c, dict(key=parallel.I64_MAX, sf=np.zeros(D), w4=np.zeros(4), ff=None, fa=None))
            e["sf"] = e["sf"] + r["sum_feat"][i].numpy()
            e["w4"] = e["w4"] + r["sum_w4"][i].numpy()
            e.setdefault("firsts", []).append((int(r["first_key"][i]), float(r["first_alpha"][i]), r["first_feat"][i].numpy().astype(np.float64)))
            if int(r["first_key"][i]) < e["key"]:
                e["key"], e["ff"], e["fa"] = int(r["first_key"][i]), r["first_feat"][i].numpy(), float(r["first_alpha"][i])
    for e in table.values():
        # sum_feat leaves every rank's LOCAL first touch out: the global first touch counts with the reference's a1^2
        # (vlmap_builder.py:166-174), the local first touches of the other ranks with their plain weight
        e["want"] = e["sf"] + sum((a * a if k == e["key"] else a) * f for k, a, f in e["firsts"])
    cells = sorted(table, key=lambda c: table[c]["key"])
    return cells, table


def _worker(rank, ws, port, tmpdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(ws), LOCAL_RANK=str(rank))
    r, w, _ = parallel.init_distributed("gloo")
    assert (r, w) == (rank, ws)
    D = 12
    cellsets = [[5, 9, 100, 7, 42, 77], [9, 3, 77, 1000, 5]]
    raws = [make_rank_raw(10 + k, D, cellsets[k], frame_lo=1000 * k) for k in range(ws)]
    merged = parallel.merge_raw(raws[rank], dst=0)
    plan = parallel.plan_merge(raws[rank]["cell"], raws[rank]["first_key"])      # collective: every rank calls it
    cells, table = expected_merge(raws)
    # the plan alone: every rank knows every voxel's final row
    assert plan.M == len(cells) and plan.cell.tolist() == cells
    assert [cells[r] for r in plan.row_of_slot.tolist()] == raws[rank]["cell"].tolist()
    assert plan.grow_key(3) == table[cells[2]]["key"] and plan.grow_key(10 ** 6) == (1 << 64) - 1
    if rank == 0:
        assert merged["cell"].tolist() == cells
        assert set(merged) == {"cell", "first_key", "acc"} and merged["acc"].shape == (len(cells), D + 4)
        local = parallel.merge_raw_local(raws)                   # single-process variant: same result
        for k in merged:
            assert torch.allclose(local[k].double(), merged[k].double(), rtol=1e-14, atol=1e-14), k
        for i, c in enumerate(cells):
            e = table[c]
            assert int(merged["first_key"][i]) == e["key"]
            # ONE reduce: every rank has already folded its first-touch sample in (a1^2 f1 for the global owner)
            want = e["want"]
            np.testing.assert_allclose(merged["acc"][i, :D].numpy(), want, rtol=1e-13, atol=1e-13)
            np.testing.assert_allclose(merged["acc"][i, D:].numpy(), e["w4"], rtol=1e-15, atol=1e-15)
        # rank 0's frames precede rank 1's: shared voxels must be owned by rank 0
        for c in set(cellsets[0]) & set(cellsets[1]):
            i = cells.index(c)
            assert int(merged["first_key"][i]) >> 32 < 1000
    else:
        assert merged is None
    # empty shard on one rank
    empty = {k: v[:0] for k, v in raws[1].items()}
    merged = parallel.merge_raw(raws[0] if rank == 0 else empty, dst=0)
    if rank == 0:
        assert sorted(merged["cell"].tolist()) == sorted(cellsets[0])
    # point-to-point chain of the replay state (24 B per voxel) as merge_accumulator does it
    coll = parallel._Coll()
    state = torch.zeros((4, 3), dtype=torch.int64)
    if rank > 0:
        coll.recv(state, rank - 1)
    state += rank + 1
    if rank < ws - 1:
        coll.send(state, rank + 1)
    else:
        assert int(state[0, 0]) == ws * (ws + 1) // 2
    # row-sharded indexing: per-voxel results of every rank's row block -> the full array on every rank
    full = np.arange(7 * 3, dtype=np.float32).reshape(7, 3)
    lo, hi = parallel.shard_rows(7, rank, ws)
    assert np.array_equal(parallel.gather_rows(full[lo:hi], 7), full)
    assert np.array_equal(parallel.gather_rows(np.arange(7, dtype=np.int32)[lo:hi], 7), np.arange(7))
    # row-sharded per-query top-1 with a cross-rank tie -> lowest global row wins
    vals = torch.tensor([[1.0, 5.0, 2.0], [4.0, 5.0, 0.5]])[rank]
    rows = torch.tensor([[3, 1, 2], [0, 4, 9]])[rank]
    v, i = parallel.global_top1(vals, rows, row_offset=100 * rank)
    assert v.tolist() == [4.0, 5.0, 2.0] and i.tolist() == [100, 1, 2]
    # row-sharded per-query top-k: k * Q candidates per rank; order = np.argsort(-v, kind="stable") of the whole map
    whole = np.array([[0.5, 3.0, 3.0, np.nan, 1.0, 3.0, -1.0, 2.0], [1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0]], np.float32)     # (Q, N)
    lo, hi = parallel.shard_rows(8, rank, ws)
    loc = torch.from_numpy(whole[:, lo:hi])
    lv, li = [], []
    for qrow in loc:                                # local top-3 per query, stable, NaN last (what ops.topk_f32 returns)
        o = np.argsort(-np.where(np.isnan(qrow.numpy()), -np.inf, qrow.numpy()), kind="stable")[:3]
        lv.append(qrow[o])
        li.append(torch.from_numpy(o))
    gv, gi = parallel.global_topk(torch.stack(lv), torch.stack(li), lo, 3)
    want = [np.argsort(-np.where(np.isnan(w), -np.inf, w), kind="stable")[:3] for w in whole]
    assert gi.tolist() == [w.tolist() for w in want], gi
    assert gv[0].tolist() == [3.0, 3.0, 3.0] and gv[1].tolist() == [1.0, 1.0, 1.0]
    Path(tmpdir, f"ok{rank}").write_text("ok")
    dist.barrier()
    dist.destroy_process_group()


def test_merge_world_size_2_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


def _fake_replay(rank, cell):
    """stand-in for avl_builder_replay_chain on CPU tensors: an order-dependent update of the 24-byte state of the selected voxels"""
    def fn(state, sel):
        c = cell.to(torch.int64)
        state[sel, 0] = state[sel, 0] * 31 + (rank + 1) * 1000 + c[sel]
        state[sel, 1] = state[sel, 1] + 1
        state[sel, 2] = 1 << 32                    # `started`
    return fn


def _sharded_worker(rank, ws, port, tmpdir, monotone=True):
    """row-sharded merge (merge_raw_sharded): every rank ends with ITS block of final rows, equal to the dense single-reduce
    result; the plan is the directory plan (nothing O(M) per rank), voxels of one rank alone travel as finished float32 rows,
    shared voxels as float64 partial sums, and the order-dependent replay state reaches the owner through the contributors in
    rank order"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(ws), LOCAL_RANK=str(rank))
    parallel.init_distributed("gloo")
    D = 12
    rng = np.random.default_rng(7)
    # contiguous frame shards see mostly disjoint voxels, a few shared ones (some by three and more ranks); one rank may hold nothing
    cellsets = [sorted(set(rng.integers(0, 400, 40 + 10 * (k % 3)).tolist())) for k in range(ws)]
    if ws >= 3:
        cellsets[2] = []
    lo = [1000 * k for k in range(ws)]
    if not monotone:
        lo[0], lo[1] = lo[1], lo[0]                # frames NOT sharded contiguously: the general plan must take over
    raws = [make_rank_raw(20 + k, D, cellsets[k], frame_lo=lo[k]) for k in range(ws)]
    cells, table = expected_merge(raws)
    dense = parallel.merge_raw(raws[rank], dst=0)
    mycell = raws[rank]["cell"]
    sh = parallel.merge_raw_sharded(raws[rank], replay_fn=_fake_replay(rank, mycell) if monotone else None, gs2=7)
    M = len(cells)
    assert sh["plan"] == ("directory" if monotone else "general")
    assert sh["M"] == M and sh["rows"] == parallel.shard_rows(M, rank, ws)
    r0, r1 = sh["rows"]
    assert sh["cell"].tolist() == cells[r0:r1] and sh["grid_feat"].shape == (r1 - r0, D)
    holders = {c: [k for k in range(ws) if c in cellsets[k]] for c in cells}
    part = {int(r): i for i, r in enumerate(sh["part_rows"].tolist())}
    for i in range(r0, r1):
        e = table[cells[i]]
        want = e["want"]
        np.testing.assert_allclose(sh["w4"][i - r0].numpy(), e["w4"], rtol=1e-15, atol=1e-15)
        assert int(sh["first_key"][i - r0]) == e["key"]
        if monotone and len(holders[cells[i]]) == 1:
            # a voxel of ONE rank: finished where it was accumulated, in float64, rounded once -- the single-process value
            assert (i - r0) not in part
            assert np.array_equal(sh["grid_feat"][i - r0].numpy(), (want / e["w4"][0]).astype(np.float32))
        else:
            np.testing.assert_allclose(sh["part_acc"][part[i - r0]].numpy(), want, rtol=1e-13, atol=1e-13)
            np.testing.assert_allclose(sh["grid_feat"][i - r0].numpy(), (want / e["w4"][0]).astype(np.float32), rtol=2e-7)
    # the blocks, gathered, are the dense reduce of merge_raw
    blocks = [None] * ws
    dist.all_gather_object(blocks, (sh["grid_feat"].numpy(), sh["w4"].numpy()))
    if rank == 0:
        gf = np.concatenate([b[0] for b in blocks], axis=0)
        w4 = np.concatenate([b[1] for b in blocks], axis=0)
        dacc = dense["acc"].numpy()
        np.testing.assert_allclose(w4, dacc[:, D:], rtol=1e-15, atol=1e-15)
        np.testing.assert_allclose(gf, (dacc[:, :D] / dacc[:, D:D + 1]).astype(np.float32), rtol=2e-7, atol=0)
    # the key after which the reference's arrays change dtype (vlmap_builder.py:286-311): first-touch key of voxel id gs2 - 1
    if monotone:
        assert sh["grow_key"] == (table[cells[6]]["key"] if M >= 7 else (1 << 64) - 1)
        # replay state: continued by every contributor of a voxel in rank order, delivered to the row's owner
        for i in range(r0, r1):
            c, s0 = cells[i], 0
            for k in holders[c]:
                s0 = s0 * 31 + (k + 1) * 1000 + c
            assert sh["state"][i - r0].tolist() == [s0, len(holders[c]), 1 << 32], (i, holders[c])
        # traffic: 64 B of side record per row that leaves, + 4 B x D (single-rank voxels) or 8 B x D (shared) -- about half of
        # the all-float64 form
        n_local = len(cellsets[rank])
        away = [c for c in cellsets[rank] if not (r0 <= cells.index(c) < r1)]
        single = sum(1 for c in away if len(holders[c]) == 1)
        assert sh["bytes_sent"] == len(away) * 64 + single * D * 4 + (len(away) - single) * D * 8
        assert sh["payload_bytes_fp64_form"] == len(away) * ((D + 4) * 8 + 8)
        total = [None] * ws
        dist.all_gather_object(total, (sh["bytes_sent"], n_local))
        if rank == 0:
            assert sum(t[0] for t in total) <= 1.3 * sum(t[1] for t in total) * (D + 4) * 8     # <= 1.3 x the local rows
    # the dense voxel-id grid from the gathered cell blocks
    allcells = [None] * ws
    dist.all_gather_object(allcells, sh["cell"].tolist())
    occ = parallel.occupied_ids_from_cells(torch.tensor(sum(allcells, []), dtype=torch.int32), 1, 20, 20)
    assert occ.shape == (1, 20, 20) and int((occ >= 0).sum()) == M and int(occ.view(-1)[cells[3]]) == 3
    # gather of finished row blocks to one rank
    shard = dict(grid_feat=sh["grid_feat"], grid_pos=torch.arange(r0, r1, dtype=torch.int32)[:, None].repeat(1, 3),
                 weight=sh["w4"][:, 0].float().contiguous(), grid_rgb=torch.full((r1 - r0, 3), rank, dtype=torch.uint8))
    fullmap = parallel.gather_row_shards(shard, ws - 1, parallel._Coll(), rank, ws)
    if rank == ws - 1:
        assert fullmap["grid_pos"][:, 0].tolist() == list(range(M)) and fullmap["grid_feat"].shape == (M, D)
        assert fullmap["grid_rgb"][:, 0].tolist() == [min(i // max(1, (M + ws - 1) // ws), ws - 1) for i in range(M)]
    else:
        assert fullmap is None
    Path(tmpdir, f"sh{rank}").write_text("ok")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("ws", [2, 3, 8])
def test_row_sharded_merge_gloo(tmp_path, ws):
    port = _free_port()
    mp.spawn(_sharded_worker, args=(ws, port, str(tmp_path)), nprocs=ws, join=True)
    assert all((tmp_path / f"sh{r}").exists() for r in range(ws))


def test_row_sharded_merge_falls_back_when_keys_are_not_rank_monotone(tmp_path):
    """frames not sharded contiguously: the directory plan's shortcut (voxel-id order = rank order, then key order) does not
    hold, every rank sees that from the plan's first all_gather and the general plan (sort of all M keys) takes over"""
    port = _free_port()
    mp.spawn(_sharded_worker, args=(3, port, str(tmp_path), False), nprocs=3, join=True)
    assert all((tmp_path / f"sh{r}").exists() for r in range(3))


def test_row_sharded_merge_single_process():
    raw = make_rank_raw(3, 6, [4, 2, 9, 11], 0)
    sh = parallel.merge_raw_sharded(raw)
    dense = parallel.merge_raw(raw)
    assert sh["rows"] == (0, 4) and sh["bytes_sent"] == 0 and sh["plan"] == "directory" and sh["part_rows"].numel() == 0
    assert torch.equal(sh["w4"], dense["acc"][:, 4 + 2:]) and torch.equal(sh["cell"], dense["cell"])
    assert torch.equal(sh["grid_feat"], (dense["acc"][:, :6] / dense["acc"][:, 6:7]).float())


def test_sharding_helpers():
    assert [parallel.shard_frames(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert [parallel.shard_frames(2, r, 4) for r in range(4)] == [(0, 1), (1, 2), (2, 2), (2, 2)]
    assert parallel.shard_rows(2_000_000, 7, 8) == (1_750_000, 2_000_000)
    spans = [parallel.shard_frames(40_000, r, 8) for r in range(8)]
    assert spans[0][0] == 0 and spans[-1][1] == 40_000 and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def test_single_process_merge_is_a_sort():
    raw = make_rank_raw(1, 4, [4, 2, 9], 0)
    perm = torch.tensor([2, 0, 1])
    shuffled = {k: v[perm] for k, v in raw.items()}
    merged = parallel.merge_raw(shuffled)
    assert torch.equal(merged["cell"], raw["cell"]) and torch.equal(merged["first_key"], raw["first_key"])
    a1 = raw["first_alpha"]
    want = (a1 * a1)[:, None] * raw["first_feat"].double() + raw["sum_feat"]
    assert torch.equal(merged["acc"][:, :4], want) and torch.equal(merged["acc"][:, 4:], raw["sum_w4"])


def test_seeded_shards_sample_like_the_single_process_run():
    """rank r fast-forwards the global NumPy RNG past the frames before its shard (VLMapBuilder.skip_pixel_shuffles):
    its sample lists are the single-process (= reference, vlmap_builder.py:275-277) lists of the same frames"""
    from avlmaps_amd.map.vlmap_builder import VLMapBuilder
    n_pix, rate, n_frames = 48 * 64, 7, 7
    np.random.seed(99)
    whole = [VLMapBuilder.sample_pixels(n_pix, rate) for _ in range(n_frames)]
    for ws in (2, 3, 8):
        for rank in range(ws):
            lo, hi = parallel.shard_frames(n_frames, rank, ws)
            np.random.seed(99)
            VLMapBuilder.skip_pixel_shuffles(lo, n_pix)
            for i in range(lo, hi):
                assert np.array_equal(VLMapBuilder.sample_pixels(n_pix, rate), whole[i]), (ws, rank, i)


def test_frame_stream_skips_and_never_hangs_on_a_full_queue(tmp_path):
    """_frame_stream(skip_shuffles=...) yields the reference's lists for a shard, inline and with the prefetch threads; a consumer
    that stops early (exception) does not leave the sampler thread blocked on a full queue"""
    import threading
    from avlmaps_amd.map.vlmap_builder import VLMapBuilder
    H, W, rate, n = 12, 16, 5, 9
    b = VLMapBuilder(tmp_path, {}, None, [None] * n, [None] * n, np.eye(4), np.eye(4))
    b.load_frame = lambda i: (np.zeros((H, W, 3), np.uint8), np.full((H, W), float(i), np.float32))
    np.random.seed(5)
    whole = [VLMapBuilder.sample_pixels(H * W, rate) for _ in range(n)]
    for prefetch in (0, 3):
        b.prefetch_frames = prefetch
        np.random.seed(5)
        got = list(b._frame_stream(4, 8, rate, skip_shuffles=4))
        assert [g[0] for g in got] == [4, 5, 6, 7]
        for i, _, depth, s, staged in got:
            assert depth[0, 0] == i and np.array_equal(s, whole[i]) and staged is None
    # the sampler thread walking the RNG (draws only) while workers compute the lists from snapshots of the state: the same
    # lists in the same order, and the global RNG ends where the serial loop leaves it
    H2, W2 = 90, 121
    b.load_frame = lambda i: (np.zeros((H2, W2, 3), np.uint8), np.full((H2, W2), float(i), np.float32))
    np.random.seed(11)
    want = []
    for _ in range(n):
        m = np.arange(H2 * W2)
        np.random.shuffle(m)                      # the reference's loop itself (vlmap_builder.py:275-277)
        want.append(m[::rate])
    end_state = np.random.get_state()
    for workers in (0, 2, "auto"):
        b.prefetch_frames, b.sampler_workers = 3, workers
        np.random.seed(11)
        got = list(b._frame_stream(0, n, rate))
        assert len(got) == n and all(np.array_equal(g[3], want[g[0]]) for g in got), workers
        st = np.random.get_state()
        assert st[2] == end_state[2] and np.array_equal(st[1], end_state[1]), workers
        assert b.pipeline_stats["sampler_workers"] == (b._n_sampler_workers() if workers == "auto" else workers)
    b.load_frame = lambda i: (np.zeros((H, W, 3), np.uint8), np.full((H, W), float(i), np.float32))
    b.prefetch_frames, b.sampler_workers = 1, 2
    before = threading.active_count()
    gen = b._frame_stream(0, n, rate)
    next(gen)
    gen.close()                                   # consumer gone with the queue full: the sampler must wind down
    assert threading.active_count() <= before + 1

    def boom(i):
        raise OSError("disk gone")
    b.load_frame = boom
    with pytest.raises(OSError):
        list(b._frame_stream(0, n, rate))
