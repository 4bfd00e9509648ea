"""GPU box: frames/s of the builder at another feature width (bench.py builds at D = 512):  probe_build_width.py D [B] [frames]
B = 0: deferred fuse (one launch per frame), 1: frame by frame, > 1: frames per launch."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(__file__), "..")))
import bench  # noqa: E402
from avlmaps_amd import ops  # noqa: E402

D = int(sys.argv[1]) if len(sys.argv) > 1 else 1536
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
n = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
H, W, Hf, Wf, rate, nbuf = 720, 1080, 347, 520, 100, 4
depths, rgbs, feats = bench.make_build_inputs(torch, H, W, Hf, Wf, D, nbuf, seed=99)
Ts = bench.pc_transforms(bench.trajectory(n))
calib = np.array([540, 0, 540, 0, 540, 360, 0, 0, 1.0])
rs = np.random.RandomState(5)
samples = []
for _ in range(nbuf):
    m = np.arange(H * W)
    rs.shuffle(m)
    samples.append(torch.from_numpy(m[::rate].astype(np.int32)).cuda())
acc = ops.VoxelAccumulator(1000, 0.05, 30, D, capacity=700_000, deferred_fuse=B == 0)
for rep in range(2):
    acc.reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if B <= 1:
        for i in range(n):
            b = i % nbuf
            acc.integrate_frame(depths[b], calib, Ts[i], samples[b], feats[b], rgbs[b], frame_idx=i)
    else:
        for j0 in range(0, n, B):
            idx = [i % nbuf for i in range(j0, min(n, j0 + B))]
            acc.integrate_batch([depths[b] for b in idx], calib, Ts[j0:j0 + len(idx)], [samples[b] for b in idx], [feats[b] for b in idx],
                                [rgbs[b] for b in idx], frame_idx0=j0)
    acc.flush()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(f"D={D} B={B}: {n / dt:.0f} frames/s, {1e6 * dt / n:.2f} us/frame, {acc.num_voxels()} voxels")
acc.close()
