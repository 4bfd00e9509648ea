"""GPU box: frames/s of the whole VLMapBuilder pipeline (host threads + H2D + kernels + checkpoints) around a free feature
extractor, i.e. what the pipeline itself can sustain: probe_pipeline.py [frames]
Reference pixel sampling (np.random.shuffle of H*W indices per frame, serial) against pixel_sampling="uniform"."""
import os
import sys
import tempfile
import time
from pathlib import Path

import numpy as np
import torch

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
from test_host_mirror import Cfg  # noqa: E402
from avlmaps_amd.map.map import Map  # noqa: E402
from avlmaps_amd.map.vlmap_builder import VLMapBuilder  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
H, W, Hf, Wf, D, nbuf = 720, 1080, 347, 520, 512, 4
depths, rgbs, feats = bench.make_build_inputs(torch, H, W, Hf, Wf, D, nbuf, seed=99)
depths_h = [d.cpu().numpy() for d in depths]
rgbs_h = [r.cpu().numpy() for r in rgbs]
traj = bench.trajectory(n)
cfg = Cfg(map_type="vlmap", grid_size=1000, cell_size=0.05, depth_sample_rate=100, cam_calib_mat=[540, 0, 540, 0, 540, 360, 0, 0, 1],
          pose_info=Cfg(pose_type="mobile_base", camera_height=1.5, base2cam_rot=[1, 0, 0, 0, -1, 0, 0, 0, -1],
                        base_forward_axis=[0, 0, -1], base_left_axis=[-1, 0, 0], base_up_axis=[0, 1, 0]))
CASES = (("reference", False, 100), ("uniform", False, 100), ("uniform", True, 100), ("reference", False, 0), ("uniform", True, 0))
if os.environ.get("PROBE_CASE"):                      # e.g. PROBE_CASE=uniform,1,0 = sampling, deferred fuse, save_every
    a, b_, c_ = os.environ["PROBE_CASE"].split(",")
    CASES = ((a, bool(int(b_)), int(c_)),)
for sampling, deferred, save_every in CASES:
    with tempfile.TemporaryDirectory() as tmp:
        tmp = Path(tmp)
        m = Map(cfg)
        pose_path = tmp / "poses.txt"
        np.savetxt(pose_path, traj)
        k = {"i": 0}

        def extractor(rgb):
            k["i"] += 1
            return feats[k["i"] % nbuf]

        b = VLMapBuilder(tmp, cfg, pose_path, [None] * n, [None] * n, m.base2cam_tf, m.base_transform, feat_extractor=extractor)
        b.load_frame = lambda i: (rgbs_h[i % nbuf], depths_h[i % nbuf])
        b.pixel_sampling, b.deferred_fuse, b.save_every = sampling, deferred, save_every      # (the product default is deferred_fuse = "auto")
        if os.environ.get("AVL_SAMPLER_WORKERS"):
            b.sampler_workers = int(os.environ["AVL_SAMPLER_WORKERS"])
        np.random.seed(0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        b.create_mobile_base_map()
        dt = time.perf_counter() - t0
        nv = len(b.last_map["grid_pos"])
        print(f"pixel_sampling={sampling:9s} deferred_fuse={deferred!s:5s} save_every={save_every:3d}: {n / dt:7.1f} frames/s ({1e3 * dt / n:.2f} ms/frame), {nv} voxels, "
              f"{len(b._map_writer.stats)} saves", flush=True)
        bt = b.build_times
        print("   frame loop %.1f frames/s; final save %.3f s; %s" % (n / bt["frame_loop_s"], bt["final_save_s"],
              {k: round(v, 3) if isinstance(v, float) else v for k, v in bt.items() if k not in ("frame_loop_s", "final_save_s")}), flush=True)
        if save_every and sampling == "uniform" and deferred:
            for st in b._map_writer.stats:
                print("   save:", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items()}, flush=True)
