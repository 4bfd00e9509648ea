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
    D = raws[0]["sum_feat"].shape[1]
    table = {}
    for r in raws:
        for i, c in enumerate(r["cell"].tolist()):
            e = table.setdefault(c, dict(key=parallel.I64_MAX, sf=np.zeros(D), w4=np.zeros(4), ff=None, fa=None))
            e["sf"] = e["sf"] + r["sum_feat"][i].numpy()
            e["w4"] = e["w4"] + r["sum_w4"][i].numpy()
            if int(r["first_key"][i]) < e["key"]:
                e["key"], e["ff"], e["fa"] = int(r["first_key"][i]), r["first_feat"][i].numpy(), float(r["first_alpha"][i])
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
    if rank == 0:
        cells, table = expected_merge(raws)
        assert merged["cell"].tolist() == cells
        local = parallel.merge_raw_local(raws)                   # single-process variant: same result
        for k in merged:
            assert torch.allclose(local[k].double(), merged[k].double(), rtol=1e-15, atol=1e-15), k
        for i, c in enumerate(cells):
            e = table[c]
            assert int(merged["first_key"][i]) == e["key"]
            np.testing.assert_allclose(merged["sum_feat"][i].numpy(), e["sf"], rtol=1e-15, atol=1e-15)
            np.testing.assert_allclose(merged["sum_w4"][i].numpy(), e["w4"], rtol=1e-15, atol=1e-15)
            assert np.array_equal(merged["first_feat"][i].numpy(), e["ff"])
            assert float(merged["first_alpha"][i]) == e["fa"]
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
    # row-sharded per-query top-1 with a cross-rank tie -> lowest global row wins
    vals = torch.tensor([[1.0, 5.0, 2.0], [4.0, 5.0, 0.5]])[rank]
    rows = torch.tensor([[3, 1, 2], [0, 4, 9]])[rank]
    v, i = parallel.global_top1(vals, rows, row_offset=100 * rank)
    assert v.tolist() == [4.0, 5.0, 2.0] and i.tolist() == [100, 1, 2]
    Path(tmpdir, f"ok{rank}").write_text("ok")
    dist.barrier()
    dist.destroy_process_group()


def test_merge_world_size_2_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


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
    for k in raw:
        assert torch.equal(merged[k], raw[k])
