import sys, tempfile
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tools"))
import yaml
from make_synth_dataset import make
from avlmaps_amd.apps import create_map, index_map
from avlmaps_amd.utils.mapping_utils import load_3d_map
from avlmaps_amd import ops
tmp = Path(tempfile.mkdtemp())
scene = make(tmp / "scene", frames=6, H=96, W=128)
cfg = tmp / "cfg.yaml"
cfg.write_text(yaml.safe_dump({"map_config": {"cam_calib_mat": [64, 0, 64, 0, 64, 48, 0, 0, 1], "depth_sample_rate": 3,
                                              "grid_size": 400, "cell_size": 0.05}, "params": {"gs": 400, "cs": 0.05}}))
create_map.main(["--data-dir", str(scene), "--config", str(cfg), "--features", "hash", "--feat-dim", "64", "--seed", "3"])
it, gf, gp, w, occ, rgb = load_3d_map(scene / "vlmap" / "vlmaps.h5df")
print("gf", gf.shape, gf.dtype, "nan", np.isnan(gf).sum(), "absmax rows min/max", np.abs(gf).max(1).min(), np.abs(gf).max(1).max(), "norm", np.linalg.norm(gf, axis=1)[:5])
from avlmaps_amd.apps.common import HashClip
from avlmaps_amd.utils.clip_utils import landmark_text_feats
q, _ = landmark_text_feats(HashClip(64), ["sofa"], 64, use_multiple_templates=True, add_other=True)
print("q", q.shape, np.linalg.norm(q, axis=1))
ref = gf.astype(np.float64) @ q.astype(np.float64).T
print("fp64 argmax zeros frac", np.mean(ref.argmax(1) == 0), "gap stats", np.abs(ref[:, 0] - ref[:, 1]).min(), np.abs(ref[:,0]-ref[:,1]).mean())
for how in ("auto", "exact", "prep"):
    from avlmaps_amd.device import DeviceArray
    src = gf if how != "prep" else ops.prepare_map(DeviceArray.from_numpy(gf))
    sc, am, _ = ops.sim_scores(src, q, precision="exact" if how == "exact" else "auto")
    sc, am = (x.numpy() if not isinstance(x, np.ndarray) else x for x in (sc, am))
    print(how, "zeros", np.mean(am == 0), "err", np.abs(sc - ref).max(), "nan", np.isnan(sc).sum())
