"""Eight ranks of the gather-plan merge in ONE process (threads + an in-process stand-in for the collectives, tests/test_merge2_gpu.py):
what the kernels of a rank cost at BASELINE's 10k-frame build, without the process hand-overs of the gloo rehearsal.
GPU box:  python tools/probe_merge2.py [ws] [frames] [reps]      (rocprofv3 --kernel-trace --stats in front for the kernel table)"""
import sys
import time
from pathlib import Path
from types import SimpleNamespace

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def main():
    ws = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    import torch
    import bench
    from avlmaps_amd import merge2, ops, parallel
    from thread_world import run_ranks
    H, W, Hf, Wf, D, rate, nbuf = 720, 1080, 347, 520, 512, 100, 4
    depths, rgbs, feats = bench.make_build_inputs(torch, H, W, Hf, Wf, D, nbuf, seed=99)
    Ts = bench.pc_transforms(bench.trajectory(frames, "spiral", 4.0))
    calib = np.array([540, 0, 540, 0, 540, 360, 0, 0, 1.0])
    rs = np.random.RandomState(5)
    samples = []
    for _ in range(nbuf):
        m = np.arange(H * W)
        rs.shuffle(m)
        samples.append(torch.from_numpy(m[::rate].astype(np.int32)).cuda())
    P = int(samples[0].numel())
    accs = []
    for r in range(ws):
        lo, hi = parallel.shard_frames(frames, r, ws)
        acc = ops.VoxelAccumulator(1000, 0.05, 30, D, capacity=max(600_000, 300_000 + 450 * (hi - lo)), deferred_fuse=True)
        acc.enable_replay_log(max(1, (hi - lo) * P))
        for j0 in range(lo, hi, 64):
            j1 = min(hi, j0 + 64)
            idx = [i % nbuf for i in range(j0, j1)]
            plan = acc.make_batch_plan([depths[b] for b in idx], [samples[b] for b in idx], [feats[b] for b in idx], [rgbs[b] for b in idx])
            acc.integrate_frames(plan, calib, Ts[j0:j1], frame_idx0=j0)
        acc.flush()
        accs.append(acc)
    torch.cuda.synchronize()
    ncell = 1000 * 1000 * 30

    import os
    prof_rank = int(os.environ.get("PROBE_CPROFILE_RANK", "-1"))      # cProfile of one rank's merges (host bookkeeping between the launches)
    prof = None
    if prof_rank >= 0:
        import cProfile
        prof = cProfile.Profile()

    def rank_fn(r, coll):
        if r == prof_rank:
            prof.enable()
            try:
                return rank_body(r, coll)
            finally:
                prof.disable()
        return rank_body(r, coll)

    def rank_body(r, coll):
        acc = accs[r]
        tim = {}
        t0 = time.perf_counter()
        n = acc.num_voxels()
        K = merge2.HipKernels(acc, n)
        out, L, info = merge2.merge_sharded_v2(K, coll if ws > 1 else None, D, merge2._bit_length(ncell - 1), 999_999, 1000, 30, True, timings=tim,
                                               sync=torch.cuda.synchronize)
        torch.cuda.synchronize()
        wall, comm, _ = merge2._phase_times(info["marks"])
        return dict(n=n, M=L.M, total_ms=1e3 * (time.perf_counter() - t0), compute_ms=round(1e3 * sum(wall[k] - comm[k] for k in wall), 3),
                    phases_ms={k: round(1e3 * (wall[k] - comm[k]), 3) for k in wall},
                    shared=int(L.A[r].sum() - L.Dn[r].sum()), sent_MB=round(8e-6 * info["remote_words"], 1), chunks=info["chunks"])

    for rep in range(reps):
        if rep == reps - 1:
            for a in accs:
                a.drop_replay_cache()
        res = run_ranks(ws, rank_fn, exclusive=True)
        print(f"--- merge {rep}" + (" (replay cache dropped)" if rep == reps - 1 else ""))
        for r, x in enumerate(res):
            print(r, x)
    if prof is not None:
        import pstats
        pstats.Stats(prof).sort_stats("tottime").print_stats(28)
    print("NOTE: one rank computes at a time (a lock dropped inside every stand-in collective); phases = the rank's own time, collectives and lock waits excluded")


if __name__ == "__main__":
    main()
