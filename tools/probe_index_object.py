"""GPU box: where AVLMap.index_object's time goes at 2 M voxels (cached query string): device work, the emptiness check, the 8 MB
heat back to the host.  probe_index_object.py [reps]"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from avlmaps_amd import _lib, ops  # noqa: E402
from avlmaps_amd.apps.common import HashClip  # noqa: E402
from avlmaps_amd.map.avlmap import AVLMap  # noqa: E402
from avlmaps_amd.map.vlmap import VLMap  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
N, D = 2_000_000, 512
feat, _ = bench.make_index_inputs(torch, N, D, 64, seed=1234)
feat_h = feat.cpu().numpy()
del feat


class Cfg(dict):
    __getattr__ = dict.__getitem__


cfg = Cfg(map_type="vlmap", grid_size=1000, cell_size=0.05, depth_sample_rate=100, cam_calib_mat=[540, 0, 540, 0, 540, 360, 0, 0, 1],
          pose_info=Cfg(pose_type="mobile_base", camera_height=1.5, base2cam_rot=[1, 0, 0, 0, -1, 0, 0, 0, -1],
                        base_forward_axis=[0, 0, -1], base_left_axis=[-1, 0, 0], base_up_axis=[0, 1, 0]))
vm = VLMap(cfg)
vm.grid_feat, vm.clip_feat_dim, vm.clip_model = feat_h, D, HashClip(D)
rng = np.random.default_rng(3)
lin = rng.choice(1000 * 1000 * 30, size=N, replace=False)
vm.grid_pos = np.stack([lin // 30000, (lin // 30) % 1000, lin % 30], 1).astype(np.int32)
am = AVLMap(Cfg(map_config=cfg, params=Cfg(cs=0.05, gs=1000)))
am.vlmap = vm
for _ in range(5):
    am.index_object("sofa", decay_rate=0.01)
lib = _lib.load()


def med(fn):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t)
    return 1e3 * float(np.median(ts))


q = vm._text_feats(["sofa"])
f = vm._device_feat()
state = {}


def device_part():
    _, a, _ = ops.sim_scores(f, q, want_scores=False, want_argmax=True, precision=vm._sim_precision)
    mask = ops.mask_from_argmax(a, 0)
    state["heat"] = vm.heatmap_from_mask(mask, 0.05, 0.01)
    _lib.check(lib.avl_stream_sync(None))


print(f"index_object end to end        {med(lambda: am.index_object('sofa', decay_rate=0.01)):.3f} ms")
print(f"  sim + mask + heat (synced)   {med(device_part):.3f} ms")
print(f"  emptiness check (argmax_f32) {med(lambda: ops.argmax_f32(state['heat'])):.3f} ms")
print(f"  heat.numpy() (8 MB)          {med(lambda: state['heat'].numpy()):.3f} ms")
print(f"  text features (cached)       {med(lambda: vm._text_feats(['sofa'])):.3f} ms")
