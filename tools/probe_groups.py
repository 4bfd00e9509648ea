"""GPU box: what a frame costs as a function of how its samples collide.  The bench scene (28 % of the samples own a voxel group,
the largest group of a frame has 8-9 members) against a far flat wall (few collisions) and a near one (many): frames/s of the
frame-by-frame C loop, plain and with deferred fuse.  If the launch were set by its longest group, the far wall would be much
faster.   usage: probe_groups.py [frames=3000]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import bench  # noqa: E402
from avlmaps_amd import ops  # noqa: E402

H, W, Hf, Wf, D, rate = 720, 1080, 347, 520, 512, 100
nbuf = 4
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
depths0, rgbs, feats = bench.make_build_inputs(torch, H, W, Hf, Wf, D, nbuf, seed=99)
Ts = bench.pc_transforms(bench.trajectory(n + 64))
calib = np.array([540, 0, 540, 0, 540, 360, 0, 0, 1.0])
rs = np.random.RandomState(5)
samples = []
for _ in range(nbuf):
    m = np.arange(H * W)
    rs.shuffle(m)
    samples.append(torch.from_numpy(m[::rate].astype(np.int32)).cuda())
scenes = {"bench scene": depths0,
          "flat wall at 5.8 m": [torch.full_like(d, 5.8) for d in depths0],
          "flat wall at 2.0 m": [torch.full_like(d, 2.0) for d in depths0],
          "flat wall at 0.6 m": [torch.full_like(d, 0.6) for d in depths0]}
for name, depths in scenes.items():
    for deferred in (False, True):
        acc = ops.VoxelAccumulator(1000, 0.05, 30, D, capacity=4_000_000, deferred_fuse=deferred)
        plan = acc.make_batch_plan([depths[i % nbuf] for i in range(64)], [samples[i % nbuf] for i in range(64)],
                                   [feats[i % nbuf] for i in range(64)], [rgbs[i % nbuf] for i in range(64)])
        done = 0
        for rep in range(2):          # the second pass over the same poses is the steady state (the voxels exist)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i0 in range(0, n - 63, 64):
                acc.integrate_frames(plan, calib, Ts[i0:i0 + 64], frame_idx0=done)
                done += 64
            acc.flush()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        frames = (n - 63 + 63) // 64 * 64
        st = dict(groups_per_frame=round(acc.num_groups() / max(done, 1), 1)) if hasattr(acc, "num_groups") else {}
        print(f"{name:20s} deferred={deferred!s:5s}: {dt / frames * 1e6:6.2f} us per frame  voxels {acc.num_voxels()}  {st}")
        acc.close()
