"""The gather-plan merge (avlmaps_amd/merge2.py) over gloo on CPU tensors: the choreography of the product path -- two all_gathers,
the replay hops, ONE payload all_to_all -- with the NumPy twin of the HIP kernels, at world sizes 2 / 3 / 8 and in one process."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from avlmaps_amd import merge2, parallel  # noqa: E402
from test_parallel_gloo import _free_port, expected_merge, make_rank_raw  # noqa: E402

GS, VH = 20, 20          # cells < 400: a 1 x 20 x 20 grid


def fake_replay(rank, cell):
    """stand-in for avl_builder_replay_chain on the twin's arrays: an order-dependent update of the 24-byte state of the selected voxels"""
    c = np.asarray(cell, np.int64)

    def fn(state, sel):
        state[sel, 0] = state[sel, 0] * 31 + (rank + 1) * 1000 + c[sel]
        state[sel, 1] = state[sel, 1] + 1
        state[sel, 2] = 1 << 32                    # `started`
    return fn


def make_world(ws, monotone=True):
    D = 12
    rng = np.random.default_rng(7)
    # contiguous frame shards see mostly disjoint voxels, a few shared ones (some by three and more ranks); one rank may hold nothing
    cellsets = [sorted(set(rng.integers(0, 400, 40 + 10 * (k % 3)).tolist())) for k in range(ws)]
    if ws >= 3:
        cellsets[2] = []
    lo = [1000 * k for k in range(ws)]
    if not monotone:
        lo[0], lo[1] = lo[1], lo[0]
    raws = [make_rank_raw(20 + k, D, cellsets[k], frame_lo=lo[k]) for k in range(ws)]
    return D, cellsets, raws


def check_block(sh, rank, ws, D, cellsets, raws, with_replay=True):
    cells, table = expected_merge(raws)
    M = len(cells)
    assert sh["plan"] == "gather" and sh["M"] == M and sh["rows"] == parallel.shard_rows(M, rank, ws)
    r0, r1 = sh["rows"]
    assert sh["cell"].tolist() == cells[r0:r1] and sh["grid_feat"].shape == (r1 - r0, D)
    holders = {c: [k for k in range(ws) if c in cellsets[k]] for c in cells}
    part = {int(r): i for i, r in enumerate(sh["part_rows"].tolist())}
    for i in range(r0, r1):
        c = cells[i]
        e = table[c]
        want = e["want"]
        np.testing.assert_allclose(sh["w4"][i - r0].numpy(), e["w4"], rtol=1e-15, atol=1e-15)
        assert sh["grid_pos"][i - r0].tolist() == [c // (GS * VH), (c // VH) % GS, c % VH]
        assert float(sh["weight"][i - r0]) == np.float32(e["w4"][0])
        if len(holders[c]) == 1:
            # a voxel of ONE rank: finished where it was accumulated, in float64, rounded once -- the single-process value
            assert (i - r0) not in part
            assert np.array_equal(sh["grid_feat"][i - r0].numpy(), (want / e["w4"][0]).astype(np.float32))
        else:
            np.testing.assert_allclose(sh["part_acc"][part[i - r0]].numpy(), want, rtol=1e-13, atol=1e-13)
            np.testing.assert_allclose(sh["grid_feat"][i - r0].numpy(), (want / e["w4"][0]).astype(np.float32), rtol=2e-7)
    # the key after which the reference's arrays change dtype (vlmap_builder.py:286-311): first-touch key of voxel id gs2 - 1
    assert sh["grow_key"] == (table[cells[6]]["key"] if M >= 7 else (1 << 64) - 1)
    if with_replay:
        # replay state: continued by every contributor of a voxel in rank order, delivered to the row's owner by the last one
        for i in range(r0, r1):
            c, s0 = cells[i], 0
            for k in holders[c]:
                s0 = s0 * 31 + (k + 1) * 1000 + c
            assert sh["state"][i - r0].tolist() == [s0, len(holders[c]), 1 << 32], (i, holders[c])
    # traffic: 64 B of side record per row that leaves, + 4 B x D (single-rank voxels) or 8 B x D (shared)
    away = [c for c in cellsets[rank] if not (r0 <= cells.index(c) < r1)]
    single = sum(1 for c in away if len(holders[c]) == 1)
    assert sh["bytes_sent"] == len(away) * 64 + single * D * 4 + (len(away) - single) * D * 8
    return cells, table


def _worker(rank, ws, port, tmpdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(ws), LOCAL_RANK=str(rank))
    parallel.init_distributed("gloo")
    D, cellsets, raws = make_world(ws)
    sh = merge2.merge_raw_sharded_v2(raws[rank], replay_fn=fake_replay(rank, raws[rank]["cell"].numpy()), gs2=7, gs=GS, vh=VH, ncell=400)
    cells, table = check_block(sh, rank, ws, D, cellsets, raws)
    R = int(os.environ.get("AVLMAPS_MERGE_CHUNK_ROWS", "0"))
    assert sh["chunks"] == (-(-max(1, -(-len(cells) // ws)) // R) if 0 < R < -(-sum(len(c) for c in cellsets) // ws) else 1)
    # the blocks, gathered, are the dense single-reduce result of parallel.merge_raw
    dense = parallel.merge_raw(raws[rank], dst=0)
    blocks = [None] * ws
    dist.all_gather_object(blocks, (sh["grid_feat"].numpy(), sh["w4"].numpy(), sh["bytes_sent"], len(cellsets[rank])))
    if rank == 0:
        gf = np.concatenate([b[0] for b in blocks], axis=0)
        w4 = np.concatenate([b[1] for b in blocks], axis=0)
        dacc = dense["acc"].numpy()
        np.testing.assert_allclose(w4, dacc[:, D:], rtol=1e-15, atol=1e-15)
        np.testing.assert_allclose(gf, (dacc[:, :D] / dacc[:, D:D + 1]).astype(np.float32), rtol=2e-7, atol=0)
        assert sum(b[2] for b in blocks) <= 1.3 * sum(b[3] for b in blocks) * (D + 4) * 8          # <= 1.3 x the local rows
    # without a replay log on ONE rank nobody replays (the flag is the minimum over the ranks' headers)
    sh2 = merge2.merge_raw_sharded_v2(raws[rank], replay_fn=None if rank == ws - 1 else fake_replay(rank, raws[rank]["cell"].numpy()),
                                      gs2=7, gs=GS, vh=VH, ncell=400)
    assert not sh2["state"].any() and torch.equal(sh2["grid_feat"], sh["grid_feat"])
    # keys not ordered by rank: every rank sees it from the headers and returns None (the caller takes the general plan)
    _, _, raws_nm = make_world(ws, monotone=False)
    assert merge2.merge_raw_sharded_v2(raws_nm[rank], gs=GS, vh=VH, ncell=400) is None
    Path(tmpdir, f"m2_{rank}").write_text("ok")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("ws,chunk_rows", [(2, None), (3, None), (8, None), (3, 7), (8, 3)])
def test_gather_plan_merge_gloo(tmp_path, monkeypatch, ws, chunk_rows):
    """chunk_rows: the payload exchange in chunks of that many rows of every owner's block (merge2.merge_sharded_v2), same blocks"""
    if chunk_rows:
        monkeypatch.setenv("AVLMAPS_MERGE_CHUNK_ROWS", str(chunk_rows))
    port = _free_port()
    mp.spawn(_worker, args=(ws, port, str(tmp_path)), nprocs=ws, join=True)
    assert all((tmp_path / f"m2_{r}").exists() for r in range(ws))


def test_gather_plan_merge_single_process():
    D, cellsets, raws = make_world(1)
    sh = merge2.merge_raw_sharded_v2(raws[0], replay_fn=fake_replay(0, raws[0]["cell"].numpy()), gs2=7, gs=GS, vh=VH, ncell=400)
    check_block(sh, 0, 1, D, cellsets, raws)
    assert sh["bytes_sent"] == 0 and sh["part_rows"].numel() == 0
    dense = parallel.merge_raw(raws[0])
    assert torch.equal(sh["cell"], dense["cell"]) and torch.equal(sh["w4"], dense["acc"][:, D:])
    assert torch.equal(sh["grid_feat"], (dense["acc"][:, :D] / dense["acc"][:, D:D + 1]).float())


def test_layout_is_the_same_arithmetic_on_both_sides():
    """what rank p plans to send to q is what q expects from p, word for word"""
    rng = np.random.default_rng(3)
    ws, D = 5, 7
    A = rng.integers(0, 50, (ws, ws))
    Dn = (A * rng.random((ws, ws))).astype(np.int64)
    H = np.triu(rng.integers(0, 9, (ws, ws)), 1)
    res = np.concatenate([[int(A.sum()) // 2, -1], A.ravel(), Dn.ravel(), H.ravel()]).astype(np.int64)
    Ls = [merge2.Plan(res, r, ws, D).layout(0) for r in range(ws)]
    check_layouts(Ls, A, Dn, ws, D)
    # the same exchange in three chunks: per-chunk tables that add up to the whole ones
    parts = rng.random((3, ws, ws))
    Ac = np.floor(A[None] * parts / parts.sum(0)).astype(np.int64)
    Ac[2] = A - Ac[:2].sum(0)
    Dc = np.minimum(Ac, np.floor(Dn[None] * parts / parts.sum(0)).astype(np.int64))
    Dc[2] = Dn - Dc[:2].sum(0)
    Dc[2] = np.clip(Dc[2], 0, Ac[2])
    Dn3 = Dc.sum(0)
    M = int(A.sum()) // 2
    R = -(-(-(-M // ws)) // 3)
    res3 = np.concatenate([[M, -1], A.ravel(), Dn3.ravel(), H.ravel(), np.stack([Ac, Dc], 1).ravel()]).astype(np.int64)
    Ps = [merge2.Plan(res3, r, ws, D, R, 3) for r in range(ws)]
    assert all(P.C == 3 for P in Ps)
    for c in range(3):
        Lc = [P.layout(c) for P in Ps]
        check_layouts(Lc, Ac[c], Dc[c], ws, D)
        for r in range(ws):
            assert Lc[r].row_lo == min(Ps[r].r1, Ps[r].r0 + c * R) and (Lc[r].row0 == np.arange(ws) * Ps[r].per + c * R).all()
    for r in range(ws):
        assert sum(P.layout(c).n_rows for c in range(3) for P in [Ps[r]]) == Ps[r].n_own
        assert (Ps[r].layout(2).lo + Ac[2][r] == Ps[r].start[1:]).all()          # the chunks of a destination are consecutive stretches of the order


def check_layouts(Ls, A, Dn, ws, D):
    for p in range(ws):
        assert Ls[p].in_splits()[p] == 0 and Ls[p].out_splits()[p] == 0
        assert Ls[p].send_total == sum(Ls[p].send_words) and Ls[p].remote_words == sum(Ls[p].in_splits())
        for q in range(ws):
            if p != q:
                assert Ls[p].in_splits()[q] == Ls[q].out_splits()[p]
            which, s_off, d_off, p_off, cnt = Ls[q].peer_lists(p)
            assert cnt == A[p, q] and which == ("send" if p == q else "recv")
            assert d_off - s_off == 8 * A[p, q] and p_off - d_off == Dn[p, q] * ((D + 1) // 2)


@pytest.mark.parametrize("ws,seed", [(2, 1), (4, 2), (5, 3), (8, 4), (13, 5), (16, 6)])
def test_gather_plan_merge_random_worlds_over_threads(ws, seed):
    """the choreography + the NumPy twin for ws ranks as THREADS of this process (tests/thread_world.py: an in-process stand-in for
    parallel._Coll that moves the same tensors), random sharing patterns: empty ranks, voxels shared by up to all ranks, a rank whose
    voxels ALL belong to other ranks' blocks, odd feature widths -- every block against the brute-force merge of the raw lists"""
    from thread_world import run_ranks
    rng = np.random.default_rng(100 + seed)
    D = int(rng.choice([3, 8, 12, 17]))
    pool = rng.permutation(400)[: int(rng.integers(30, 200))]
    cellsets = []
    for k in range(ws):
        style = rng.integers(0, 4)
        if style == 0:
            cs = []                                                       # a rank without voxels
        elif style == 1:
            cs = rng.choice(pool, int(rng.integers(1, len(pool))), replace=False).tolist()      # heavy sharing
        else:
            cs = rng.integers(0, 400, int(rng.integers(5, 60))).tolist()
        cellsets.append(sorted(set(int(c) for c in cs)))
    if not any(cellsets):
        cellsets[0] = [7, 9]
    raws = [make_rank_raw(300 + 17 * seed + k, D, cellsets[k], frame_lo=1000 * k) for k in range(ws)]

    def rank_fn(r, coll):
        return merge2.merge_raw_sharded_v2(raws[r], group=None, replay_fn=fake_replay(r, raws[r]["cell"].numpy()), gs2=7, gs=GS, vh=VH, ncell=400,
                                           coll=coll)
    outs = run_ranks(ws, rank_fn)
    M = None
    for r in range(ws):
        cells, _ = check_block(outs[r], r, ws, D, cellsets, raws)
        M = len(cells)
    assert sum(o["rows"][1] - o["rows"][0] for o in outs) == M
    assert [c for o in outs for c in o["cell"].tolist()] == cells
    assert all(o["chunks"] == 1 for o in outs)
    # the same worlds with the exchange in chunks of a few rows: the same blocks, bit for bit, in buffers of a chunk
    R = int(rng.integers(1, 9))
    os.environ["AVLMAPS_MERGE_CHUNK_ROWS"] = str(R)
    try:
        outs_c = run_ranks(ws, rank_fn)
    finally:
        del os.environ["AVLMAPS_MERGE_CHUNK_ROWS"]
    per_max = -(-sum(len(c) for c in cellsets) // ws)
    if R < per_max and merge2.max_chunks(ws) >= 2:
        assert all(o["chunks"] == outs_c[0]["chunks"] for o in outs_c) and outs_c[0]["chunks"] >= min(2, -(-max(1, -(-M // ws)) // R))
    for a, b in zip(outs, outs_c):
        for k in ("cell", "grid_feat", "grid_pos", "weight", "grid_rgb", "w4", "state", "part_rows", "part_acc"):
            assert torch.equal(a[k], b[k]), k
        assert a["bytes_sent"] == b["bytes_sent"] and a["rows"] == b["rows"] and a["grow_key"] == b["grow_key"]
