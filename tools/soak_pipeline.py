"""GPU box: soak of VLMapBuilder's threaded frame loop (decode pool, sampler thread, pinned stager + copy stream, deferred fuse on
probation, skipped checkpoints).  Every configuration must write the SAME map as the inline single-threaded build of the same
seeded sequence -- array for array.  soak_pipeline.py [seconds] [seed]"""
import os
import sys
import tempfile
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(os.environ.get("GRAFT_REPO_ROOT", Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import bench  # noqa: E402
from test_host_mirror import Cfg  # noqa: E402
from avlmaps_amd.map.map import Map  # noqa: E402
from avlmaps_amd.map.vlmap_builder import VLMapBuilder  # noqa: E402
from avlmaps_amd.utils.mapping_utils import load_3d_map  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t_end = time.time() + budget
cases = fails = 0
while time.time() < t_end:
    H, W = int(rng.choice([90, 120, 180])), int(rng.choice([120, 160, 240]))
    Hf, Wf, D = H // 2, W // 2, int(rng.choice([16, 64, 512]))
    n = int(rng.integers(20, 160))
    rate = int(rng.choice([3, 10, 25]))
    nbuf = 5
    g = torch.Generator(device="cuda").manual_seed(int(rng.integers(1 << 30)))
    feats = [torch.randn((Hf, Wf, D), device="cuda", generator=g).contiguous() for _ in range(nbuf)]
    yy, xx = np.meshgrid(np.linspace(-1, 1, H), np.linspace(-1, 1, W), indexing="ij")
    depths = [(2.6 + 1.6 * np.sin(2 * xx + 0.37 * i) * np.cos(1.5 * yy) + 0.7 * yy).astype(np.float32) for i in range(nbuf)]
    rgbs = [rng.integers(0, 256, (H, W, 3), dtype=np.uint8) for _ in range(nbuf)]
    cfg = Cfg(map_type="vlmap", grid_size=400, cell_size=0.05, depth_sample_rate=rate, cam_calib_mat=[W / 2, 0, W / 2, 0, W / 2, H / 2, 0, 0, 1],
              pose_info=Cfg(pose_type="mobile_base", camera_height=1.5, base2cam_rot=[1, 0, 0, 0, -1, 0, 0, 0, -1],
                            base_forward_axis=[0, 0, -1], base_left_axis=[-1, 0, 0], base_up_axis=[0, 1, 0]))
    traj = bench.trajectory(n)
    seed = int(rng.integers(1 << 30))
    sampling = str(rng.choice(["reference", "uniform"]))

    def build(tmp, **opts):
        m = Map(cfg)
        np.savetxt(tmp / "poses.txt", traj)
        k = {"i": 0}

        def extractor(rgb):
            k["i"] += 1
            return feats[k["i"] % nbuf]
        b = VLMapBuilder(tmp, cfg, tmp / "poses.txt", [None] * n, [None] * n, m.base2cam_tf, m.base_transform, feat_extractor=extractor)
        b.load_frame = lambda i: (rgbs[i % nbuf], depths[i % nbuf])
        b.pixel_sampling = sampling
        b.capacity = 4096
        for kk, v in opts.items():
            setattr(b, kk, v)
        np.random.seed(seed)
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):
            b.create_mobile_base_map()
        return load_3d_map(tmp / "vlmap" / "vlmaps.h5df"), b

    with tempfile.TemporaryDirectory() as t0, tempfile.TemporaryDirectory() as t1:
        ref, _ = build(Path(t0), prefetch_frames=0, deferred_fuse=False, save_every=0)
        opts = dict(prefetch_frames=int(rng.integers(1, 7)), batch_frames=int(rng.choice([1, 1, 2, 4])), deferred_fuse=rng.choice(["auto", False, True]).item(),
                    stage_frames=bool(rng.random() < 0.8), save_every=int(rng.choice([0, 7, 25])), skip_busy_checkpoints=bool(rng.random() < 0.7),
                    sampler_workers=int(rng.choice([0, 1, 3, 5])))
        if opts["deferred_fuse"] in ("True", "False"):
            opts["deferred_fuse"] = opts["deferred_fuse"] == "True"
        got, b = build(Path(t1), **opts)
    cases += 1
    # one launch pair per BATCH sums a voxel's samples of several frames in one list: its features equal the frame-by-frame ones to
    # float64 rounding (profiles/HISTORY.md 4.3), everything else and every non-batched configuration bit for bit
    feat_ok = (np.array_equal(ref[1], got[1]) if opts["batch_frames"] == 1 else
               (ref[1].shape == got[1].shape and np.allclose(ref[1], got[1], rtol=1e-6, atol=1e-30)))
    same = ref[0] == got[0] and feat_ok and all(np.array_equal(a, c) for a, c in zip(ref[2:6], got[2:6]))
    if not same:
        fails += 1
        which = [nm for nm, a, c in zip(("grid_feat", "grid_pos", "weight", "occupied_ids", "grid_rgb"), ref[1:6], got[1:6]) if not np.array_equal(a, c)]
        print(f"MISMATCH {which} H={H} W={W} D={D} n={n} rate={rate} sampling={sampling} seed={seed} opts={opts}", flush=True)
print(f"soak_pipeline: {cases} cases, {fails} mismatches (every threaded / staged / deferred / batched configuration against the inline build)")
sys.exit(1 if fails else 0)
