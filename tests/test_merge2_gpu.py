"""The gather-plan merge on the GPU: the product choreography (avlmaps_amd/merge2.merge_sharded_v2) with the HIP kernels
(csrc/avl_merge2.hip, avl_builder_m2_pack) for several ranks of ONE process -- every rank a thread with its own VoxelAccumulator,
the collectives an in-process stand-in that moves the same bytes -- against (a) the single-process map of all frames and (b) the
NumPy twin of the kernels on the exported accumulators, bit for bit."""
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, str(Path(__file__).resolve().parent))


@pytest.fixture(scope="module")
def ops():
    from avlmaps_amd import _lib, ops
    _lib.load()
    _lib.require_gpu()
    return ops


from thread_world import run_ranks  # noqa: E402


def build_shards(ops, ws, D=64, nfr=24, seed=7, rate=5, cs=0.1, gs=400, replay=True):
    from oracle import avl_oracle as O
    from test_builder_gpu import synth_scene
    rng = np.random.default_rng(seed)
    H, W, Hf, Wf = 120, 160, 58, 77
    cam_h = 1.5
    calib = np.array([W / 2, 0, W / 2, 0, W / 2, H / 2, 0, 0, 1.0])
    depths, rgbs, feats, poses = synth_scene(rng, nfr, H, W, Hf, Wf, D)
    b2c, bt = O.setup_transforms([1, 0, 0, 0, -1, 0, 0, 0, -1], cam_h, [0, 0, -1], [-1, 0, 0], [0, 1, 0])
    Ts = O.pc_transforms(poses, bt, b2c)
    rs = np.random.RandomState(3)
    samples = [O.sample_indices(rs, H * W, rate) for _ in range(nfr)]
    fs = [np.ascontiguousarray(np.transpose(f, (1, 2, 0))) for f in feats]
    vh = int(cam_h / cs)
    from avlmaps_amd import parallel

    def build(lo, hi):
        acc = ops.VoxelAccumulator(gs, cs, vh, D, capacity=1 << 16)
        if replay:
            acc.enable_replay_log(max(1, (hi - lo) * len(samples[0])))
        for i in range(lo, hi):
            acc.integrate_frame(depths[i], calib, Ts[i], samples[i], fs[i], rgbs[i], frame_idx=i)
        return acc
    whole = build(0, nfr)
    shards = [build(*parallel.shard_frames(nfr, r, ws)) for r in range(ws)]
    return whole, shards, gs, vh


@pytest.mark.parametrize("ws,D", [(1, 64), (2, 64), (3, 30), (8, 64), (8, 5)])
def test_gather_plan_merge_on_device_equals_the_single_process_map_and_the_twin(ops, ws, D):
    check_merge_world(ops, ws, D)


@pytest.mark.parametrize("ws,D,chunk_rows", [(2, 64, 500), (3, 30, 97), (8, 64, 64), (8, 5, 1000)])
def test_gather_plan_merge_in_chunks(ops, ws, D, chunk_rows):
    """the payload exchange in chunks of chunk_rows rows of every owner's block (double-buffered send buffers, pack of chunk c + 1
    issued before chunk c folds): the same map"""
    check_merge_world(ops, ws, D, chunk_rows=chunk_rows)


def test_overlapped_all_to_all_over_rccl():
    """parallel._Coll.all_to_all_start / _finish on RCCL with the one rank a 1-GPU box can run (tests/dist_a2a_worker.py)"""
    import os
    import subprocess
    from test_parallel_gloo import _free_port
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               AVLMAPS_FORCE_COLLECTIVES="1")
    env.pop("AVLMAPS_DIST_BACKEND", None)
    r = subprocess.run([sys.executable, str(Path(__file__).resolve().parent / "dist_a2a_worker.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "A2A_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def check_merge_world(ops, ws, D, chunk_rows=None, **scene):
    """(also the body of tools/fuzz_merge2.py's random worlds)"""
    import os
    import torch
    from avlmaps_amd import merge2
    if chunk_rows:
        os.environ["AVLMAPS_MERGE_CHUNK_ROWS"] = str(chunk_rows)
        try:
            return check_merge_world(ops, ws, D, None, _chunked=chunk_rows, **scene)
        finally:
            del os.environ["AVLMAPS_MERGE_CHUNK_ROWS"]
    chunked = scene.pop("_chunked", None)
    whole, shards, gs, vh = build_shards(ops, ws, D=D, **scene)
    want = whole.finalize()
    M = len(want["grid_pos"])
    ncell = gs * gs * vh
    grow_row = M // 2                                    # (a growth inside the map: the replay switches dtypes at that voxel's key)

    def device_rank(r, coll):
        acc = shards[r]
        n = acc.num_voxels()
        K = merge2.HipKernels(acc, n)
        out, L, info = merge2.merge_sharded_v2(K, coll if ws > 1 else None, D, merge2._bit_length(ncell - 1), grow_row, gs, vh, True,
                                               timings={}, sync=torch.cuda.synchronize)
        torch.cuda.synchronize()
        return {k: v.cpu().numpy() for k, v in out.items()}, L, info

    def twin_rank(r, coll):
        raw = shards[r].export_raw()
        raw["first_key"] = raw["first_key"].astype(np.int64)
        K = merge2.HostKernels(raw)
        out, L, info = merge2.merge_sharded_v2(K, coll if ws > 1 else None, D, merge2._bit_length(ncell - 1), grow_row, gs, vh, False)
        return out, L, info

    dev = run_ranks(ws, device_rank)
    twin = run_ranks(ws, twin_rank)
    from avlmaps_amd import parallel
    pos = np.concatenate([d[0]["grid_pos"] for d in dev])
    feat = np.concatenate([d[0]["grid_feat"] for d in dev])
    assert np.array_equal(pos, want["grid_pos"])                                   # the reference's voxel ids, bit-exact, in id order
    cells = np.concatenate([d[0]["cell"] for d in dev])
    occ = parallel.occupied_ids_from_cells(torch.from_numpy(cells), gs, gs, vh).numpy()
    assert np.array_equal(occ, want["occupied_ids"])
    shared = 0
    for r in range(ws):
        (o, L, info), (t, Lt, _) = dev[r], twin[r]
        assert (L.r0, L.r1) == parallel.shard_rows(M, r, ws) == (Lt.r0, Lt.r1) and L.M == M == Lt.M
        assert np.array_equal(L.A, Lt.A) and np.array_equal(L.Dn, Lt.Dn) and np.array_equal(L.H, Lt.H) and L.grow_key == Lt.grow_key
        assert np.array_equal(o["grid_feat"], t["grid_feat"])                      # kernels == their NumPy twin, bit for bit
        assert np.array_equal(o["grid_pos"], t["grid_pos"]) and np.array_equal(o["cell"], t["cell"])
        shared += int(L.A[r].sum() - L.Dn[r].sum())
        assert info["have_log"]
        if chunked and ws > 1 and chunked < -(-sum(info["n_all"]) // ws) and merge2.max_chunks(ws) >= 2:
            assert info["chunks"] == L.C == twin[r][2]["chunks"], (info["chunks"], L.C, twin[r][2]["chunks"])
            assert L.C >= min(2, -(-max(1, -(-M // ws)) // chunked)), (L.C, M, ws, chunked)
            assert info["buffer_words"] == min(2, L.C) * max(int(L.layout(c).send_total) for c in range(L.C))     # two chunks, not the whole payload
    if ws > 1 and not scene:
        assert shared > 50, shared                                                 # the point of the test: voxels several ranks touched
    # against the single-process build: a voxel of one rank is bit-identical, a shared one differs by the float64 summation order
    np.testing.assert_allclose(feat, want["grid_feat"], rtol=1e-6, atol=1e-6)
    assert np.mean(feat == want["grid_feat"]) > 0.99
    # exact sequential weight / colour through the replay hops (growth key = first-touch key of voxel M // 2 on both sides)
    w = np.concatenate([d[0]["weight"] for d in dev])
    rgb = np.concatenate([d[0]["grid_rgb"] for d in dev])
    one = run_ranks(1, lambda r, coll: merge2.merge_sharded_v2(merge2.HipKernels(whole, whole.num_voxels()), None, D, merge2._bit_length(ncell - 1),
                                                               grow_row, gs, vh, True))[0][0]
    assert np.array_equal(w, one["weight"].cpu().numpy()) and np.array_equal(rgb, one["grid_rgb"].cpu().numpy())
    assert np.array_equal(one["grid_feat"].cpu().numpy(), want["grid_feat"])       # one rank: the plain finalisation, bit for bit


def test_gather_plan_merge_through_the_product_entry_point(ops):
    """parallel.merge_accumulator_sharded (one process): the gather plan is the default, its block is the whole map and equals
    VoxelAccumulator.finalize bit for bit, replay included; AVLMAPS_MERGE_PLAN=directory still gives the same map"""
    import os
    import torch
    from avlmaps_amd import parallel
    whole, _, gs, vh = build_shards(ops, 1, D=64)
    want = whole.finalize()
    tim = {}
    out = parallel.merge_accumulator_sharded(whole, timings=tim, gather_to=0)
    assert "gather plan" in tim["plan"] and tim["merged_voxels"] == len(want["grid_pos"]) and tim["exact_rgb"]
    for k in ("grid_feat", "grid_pos", "weight", "grid_rgb"):
        assert np.array_equal(out[k].cpu().numpy(), want[k]), k
        assert np.array_equal(out["full"][k].cpu().numpy(), want[k]), k
    assert np.array_equal(out["full"]["occupied_ids"].cpu().numpy(), want["occupied_ids"])
    # the merge's replay and the plain finalisation share ONE voxel-sorted form of the replay log (it lives in scratch allocated with
    # the log): merge -> finalize -> merge without a frame in between must give the same weights and colours each time
    again = whole.finalize()
    third = parallel.merge_accumulator_sharded(whole, timings={})
    for k in ("weight", "grid_rgb", "grid_feat"):
        assert np.array_equal(again[k], want[k]) and np.array_equal(third[k].cpu().numpy(), want[k]), k
    os.environ["AVLMAPS_MERGE_PLAN"] = "directory"
    try:
        old = parallel.merge_accumulator_sharded(whole, timings={})
    finally:
        del os.environ["AVLMAPS_MERGE_PLAN"]
    for k in ("grid_feat", "grid_pos", "weight", "grid_rgb"):
        assert torch.equal(old[k], out[k]), k


def test_random_worlds_slice(ops):
    """a fixed-seed slice of tools/fuzz_merge2.py (857 random worlds were clean in round 6): random rank counts, feature widths incl. odd
    ones, sample rates, cell sizes and grids"""
    rng = np.random.default_rng(2)
    for _ in range(10):
        ws = int(rng.integers(1, 9))
        check_merge_world(ops, ws, int(rng.choice([3, 5, 16, 30, 64, 256, 512, 768])), nfr=int(rng.integers(max(2, ws), 25)),
                          seed=int(rng.integers(0, 1 << 30)), rate=int(rng.choice([1, 3, 5, 11])), cs=float(rng.choice([0.05, 0.1, 0.3])),
                          gs=int(rng.choice([120, 400, 1000])))
