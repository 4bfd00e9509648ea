"""GPU box: the index side through the reference-shaped API at full size (2 M voxels x 512): where the wall-clock of
AVLMap.load_map / index_map / init_categories / index_object goes.  probe_index_api.py [voxels]"""
import cProfile
import os
import pstats
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_host_mirror import Cfg  # noqa: E402
from avlmaps_amd.apps.common import HashClip  # noqa: E402
from avlmaps_amd.map import AVLMap  # noqa: E402
from avlmaps_amd.utils.mapping_utils import save_3d_map  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
gs, vh, D = 1000, 30, 512
rng = np.random.default_rng(0)
cells = rng.choice(gs * gs * vh, size=N, replace=False)
pos = np.stack([cells // (gs * vh), (cells // vh) % gs, cells % vh], 1).astype(np.int32)
feat = rng.standard_normal((N, D), dtype=np.float32)
feat *= 14.0 / np.linalg.norm(feat, axis=1, keepdims=True)
occ = -np.ones((gs, gs, vh), np.int32)
occ[pos[:, 0], pos[:, 1], pos[:, 2]] = np.arange(N, dtype=np.int32)
mcfg = Cfg(map_type="vlmap", grid_size=gs, cell_size=0.05, depth_sample_rate=100, cam_calib_mat=[540, 0, 540, 0, 540, 360, 0, 0, 1],
          pose_info=Cfg(pose_type="mobile_base", camera_height=1.5, base2cam_rot=[1, 0, 0, 0, -1, 0, 0, 0, -1],
                        base_forward_axis=[0, 0, -1], base_left_axis=[-1, 0, 0], base_up_axis=[0, 1, 0]),
          potential_obstacle_names=["chair", "table", "sofa"], obstacle_names=["chair"], dilate_iter=3, gaussian_sigma=1.0,
          customize_obstacle_map=False)
cfg = Cfg(map_config=mcfg, params=Cfg(cs=0.05, gs=gs))


def timed(label, fn):
    t0 = time.perf_counter()
    r = fn()
    print(f"{label:58s} {1e3 * (time.perf_counter() - t0):9.1f} ms", flush=True)
    return r


with tempfile.TemporaryDirectory() as tmp:
    d = Path(tmp) / "vlmap"
    d.mkdir()
    timed("save_3d_map (4.2 GB)", lambda: save_3d_map(d / "vlmaps.h5df", feat, pos, np.ones(N, np.float32), occ, list(range(10)),
                                                      np.zeros((N, 3), np.uint8)))
    m = AVLMap(cfg, data_dir=tmp)
    if os.environ.get("PROBE_NO_PREFETCH"):
        m.vlmap.prefetch_device = False
    timed("AVLMap.load_map", lambda: m.load_map(tmp))
    vm = m.vlmap
    vm.clip_feat_dim, vm.clip_model = D, HashClip(D)
    if os.environ.get("PROBE_GAP"):       # what upstream does between load_map and the first query: load CLIP (seconds)
        timed(f"(host-side work for {os.environ['PROBE_GAP']} s, e.g. loading CLIP)", lambda: sum(i * i for i in range(int(2e7 * float(os.environ["PROBE_GAP"])))))
    timed("index_map('sofa', no categories)  first call (upload + prepare)", lambda: vm.index_map("sofa", with_init_cat=False))
    for _ in range(2):
        timed("index_map('chair', no categories)", lambda: vm.index_map("chair", with_init_cat=False))
    cats = ["void"] + [f"thing{i}" for i in range(62)] + ["sofa", "void"]
    timed("init_categories(64)  (scores_mat to the host, like upstream)", lambda: vm.init_categories(cats[1:-1]))
    timed("index_map('sofa') with categories", lambda: vm.index_map("sofa"))
    timed("AVLMap.index_object('sofa')  (mask + heat decay)", lambda: m.index_object("sofa", decay_rate=0.01))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        vm.index_map("table", with_init_cat=False)
        m.index_object("sofa", decay_rate=0.01)
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
