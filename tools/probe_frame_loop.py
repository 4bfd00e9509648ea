"""host time vs device time of the frame-by-frame build loop (scratch probe)"""
import sys, time
import numpy as np
import torch
import os; sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import bench
from avlmaps_amd import ops

H, W, Hf, Wf, D, rate = 720, 1080, 347, 520, 512, 100
nbuf, n = 4, 4000
depths, rgbs, feats = bench.make_build_inputs(torch, H, W, Hf, Wf, D, nbuf, seed=99)
Ts = bench.pc_transforms(bench.trajectory(n))
calib = np.array([540, 0, 540, 0, 540, 360, 0, 0, 1.0])
rs = np.random.RandomState(5)
samples = []
for _ in range(nbuf):
    m = np.arange(H * W); rs.shuffle(m)
    samples.append(torch.from_numpy(m[::rate].astype(np.int32)).cuda())
for deferred in (False, True):
    acc = ops.VoxelAccumulator(1000, 0.05, 30, D, capacity=2_500_000, deferred_fuse=deferred)
    for rep in range(2):
        acc.reset(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            b = i % nbuf
            acc.integrate_frame(depths[b], calib, Ts[i], samples[b], feats[b], rgbs[b], frame_idx=i)
        acc.flush()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    print(f"deferred={deferred}: host enqueue {1e6*(t1-t0)/n:.2f} us/frame, total {1e6*(t2-t0)/n:.2f} us/frame")
    acc.close()
# host cost alone: tiny launches (64 samples) so that the device is never the bottleneck
small = [s[:64].contiguous() for s in samples]
for deferred in (False, True):
    acc = ops.VoxelAccumulator(1000, 0.05, 30, D, capacity=2_500_000, deferred_fuse=deferred)
    for rep in range(2):
        acc.reset(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            b = i % nbuf
            acc.integrate_frame(depths[b], calib, Ts[i], small[b], feats[b], rgbs[b], frame_idx=i)
        acc.flush()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    print(f"64 samples, deferred={deferred}: host enqueue {1e6*(t1-t0)/n:.2f} us/frame, total {1e6*(t2-t0)/n:.2f} us/frame")
    acc.close()
