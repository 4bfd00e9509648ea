"""GPU box: what a periodic checkpoint of a large map costs on the builder's thread (finalisation + device-to-host copy),
full copy against the lean dirty-rows transfer.  probe_checkpoint.py [frames] [frames between checkpoints]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(__file__), "..")))
import bench  # noqa: E402
from avlmaps_amd import ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
step = int(sys.argv[2]) if len(sys.argv) > 2 else 100
H, W, Hf, Wf, D, rate, nbuf, B = 720, 1080, 347, 520, 512, 100, 4, 50
depths, rgbs, feats = bench.make_build_inputs(torch, H, W, Hf, Wf, D, nbuf, seed=99)
Ts = bench.pc_transforms(bench.trajectory(n + step))
calib = np.array([540, 0, 540, 0, 540, 360, 0, 0, 1.0])
rs = np.random.RandomState(5)
samples = []
for _ in range(nbuf):
    m = np.arange(H * W)
    rs.shuffle(m)
    samples.append(torch.from_numpy(m[::rate].astype(np.int32)).cuda())
P = int(samples[0].numel())
acc = ops.VoxelAccumulator(1000, 0.05, 30, D, capacity=2_600_000)
acc.enable_replay_log((n + step) * P)


def fuse(i0, i1):
    for j0 in range(i0, i1, B):
        idx = [i % nbuf for i in range(j0, min(i1, j0 + B))]
        acc.integrate_batch([depths[b] for b in idx], calib, Ts[j0:j0 + len(idx)], [samples[b] for b in idx], [feats[b] for b in idx],
                            [rgbs[b] for b in idx], frame_idx0=j0)


fuse(0, n)
torch.cuda.synchronize()
t0 = time.perf_counter()
full = acc.finalize(want_dirty=True)                       # what a checkpoint did in round 2's first version: the whole map to the host
t1 = time.perf_counter()
print(f"{acc.num_voxels()} voxels after {n} frames; full finalize + D2H: {1e3 * (t1 - t0):.1f} ms "
      f"({sum(v.nbytes for v in full.values() if v is not None) / 1e9:.2f} GB to the host)")
fuse(n, n + step)
torch.cuda.synchronize()
t0 = time.perf_counter()
lean = acc.finalize_rows(n_saved=len(full["grid_pos"]))
t1 = time.perf_counter()
nb = sum(v.nbytes for v in lean["rows"].values())
print(f"{step} frames later: lean checkpoint (finalize + changed / new rows only): {1e3 * (t1 - t0):.1f} ms, {len(lean['idx'])} rows, "
      f"{nb / 1e6:.1f} MB to the host")
ref = acc.finalize()
for name in ("grid_feat", "grid_pos", "weight", "grid_rgb"):
    assert np.array_equal(lean["rows"][name], ref[name][lean["idx"]]), name
print("the rows that came over equal the full finalisation")
t0 = time.perf_counter()
lean2 = acc.finalize_rows(n_saved=lean["n"])            # reuses the staging buffer: lean["rows"] is stale from here on
print(f"again at once (nothing fused since): {1e3 * (time.perf_counter() - t0):.1f} ms, {len(lean2['idx'])} rows")
acc.close()
