"""CPU tests of the host-side mirror of the reference interface (no GPU compute)."""
import hashlib
import json
from pathlib import Path

import numpy as np
import pytest

from avlmaps_amd.map.map import Map
from avlmaps_amd.map.vlmap_builder import VLMapBuilder
from avlmaps_amd.utils import mapping_utils as mu
from avlmaps_amd.utils.clip_utils import multiple_templates

GOLDEN = Path(__file__).resolve().parent / "golden"


class Cfg(dict):
    __getattr__ = dict.__getitem__


def make_cfg(g):
    return Cfg(map_type="vlmap", grid_size=int(g["gs"]), cell_size=float(g["cs"]), depth_sample_rate=int(g["rate"]),
               cam_calib_mat=[float(x) for x in g["calib"]],
               pose_info=Cfg(pose_type="mobile_base", camera_height=float(g["camera_height"]),
                             base2cam_rot=[float(x) for x in g["base2cam_rot"]],
                             base_forward_axis=list(g["base_axes"][0]), base_left_axis=list(g["base_axes"][1]),
                             base_up_axis=list(g["base_axes"][2])))


def test_templates_match_reference_hash():
    meta = json.loads((GOLDEN / "templates.json").read_text())
    assert len(multiple_templates) == meta["n_templates"] == 63
    assert hashlib.sha256("\n".join(multiple_templates).encode()).hexdigest() == meta["sha256_of_newline_joined"]


def test_geometry_helpers_match_reference(golden):
    g = golden("g1_geometry.npz")
    for v, tf in zip(g["posevecs"], g["pose_tfs"]):
        assert np.array_equal(mu.cvt_pose_vec2tf(v), tf)
    assert np.array_equal(mu.get_sim_cam_mat(347, 520), g["simcam_347_520"])
    gs, cs = int(g["vox_gs"]), float(g["vox_cs"])
    got = np.array([mu.base_pos2grid_id_3d(gs, cs, *p) for p in g["vox_pts"]])
    assert np.array_equal(got, g["vox_ids"])


@pytest.mark.parametrize("name", ["g2a_builder_small.npz", "g2b_builder_growth.npz"])
def test_transforms_and_pose_chain(golden, name):
    from oracle import avl_oracle as O
    g = golden(name)
    m = Map(make_cfg(g))
    assert np.array_equal(m.base2cam_tf, g["base2cam_tf"]) and np.array_equal(m.base_transform, g["base_transform"])
    assert m.gs == int(g["gs"]) and m.cs == float(g["cs"])
    b = VLMapBuilder(Path("/tmp"), make_cfg(g), None, [], [], m.base2cam_tf, m.base_transform)
    Ts = b.frame_transforms(g["poses_rt"])
    assert np.array_equal(np.stack(Ts), O.pc_transforms(g["poses_rt"], g["base_transform"], g["base2cam_tf"]))


def test_sampling_follows_global_numpy_rng(golden):
    g = golden("g2a_builder_small.npz")
    H, W = g["depths"].shape[1:]
    np.random.seed(1234)            # the seed tools/gen_golden.py gave the reference run
    for i in range(len(g["samples"])):
        assert np.array_equal(VLMapBuilder.sample_pixels(H * W, int(g["rate"])), g["samples"][i])


def test_save_load_roundtrip(tmp_path):
    rng = np.random.default_rng(0)
    n, D = 17, 8
    arrays = dict(grid_feat=rng.standard_normal((n, D)).astype(np.float32), grid_pos=rng.integers(0, 9, (n, 3)).astype(np.int32),
                  weight=rng.random(n).astype(np.float32), occupied_ids=-np.ones((9, 9, 3), np.int32),
                  grid_rgb=rng.integers(0, 255, (n, 3)).astype(np.uint8))
    p = tmp_path / "vlmap" / "vlmaps.h5df"
    p.parent.mkdir()
    assert not mu.map_file_exists(p)
    mu.save_3d_map(p, arrays["grid_feat"], arrays["grid_pos"], arrays["weight"], arrays["occupied_ids"], [2, 0, 1], arrays["grid_rgb"])
    assert mu.map_file_exists(p)
    it, gf, gp, w, occ, rgb = mu.load_3d_map(p)
    assert it == [0, 1, 2]
    for a, b in ((gf, "grid_feat"), (gp, "grid_pos"), (w, "weight"), (occ, "occupied_ids"), (rgb, "grid_rgb")):
        assert np.array_equal(a, arrays[b]) and a.dtype == arrays[b].dtype


def test_map_attribute_surface_and_obstacles():
    cfg = Cfg(map_type="vlmap", grid_size=20, cell_size=0.05,
              pose_info=Cfg(camera_height=1.5, base2cam_rot=[1, 0, 0, 0, -1, 0, 0, 0, -1], base_forward_axis=[0, 0, -1],
                            base_left_axis=[-1, 0, 0], base_up_axis=[0, 1, 0]))
    m = Map.create(cfg)
    for a in ("grid_feat", "grid_pos", "weight", "occupied_ids", "grid_rgb", "mapped_iter_list", "obstacles_map",
              "obstacles_cropped", "scores_mat", "categories"):
        assert getattr(m, a) is None
    occ = -np.ones((20, 20, 30), np.int32)
    occ[5, 6, 3] = 4
    occ[9, 12, 10] = 7
    occ[2, 2, 3] = 0          # voxel id 0 is NOT an obstacle upstream (`> 0`, map.py:92)
    m.occupied_ids = occ
    obs = m.generate_obstacle_map()
    assert obs.shape == (20, 20) and not obs[5, 6] and not obs[9, 12] and obs[2, 2]
    assert (m.rmin, m.rmax, m.cmin, m.cmax) == (5, 9, 6, 12) and m.obstacles_cropped.shape == (5, 7)
    with pytest.raises(Exception, match="Categories are not preloaded"):
        m.index_map("sofa", with_init_cat=True)
    assert VLMapBuilder(Path("/tmp"), cfg, None, [], [], None, None).create_camera_map() is NotImplementedError


def test_get_lseg_feat_protocol_matches_reference(golden):
    """sliding-window evaluation of lseg_utils.py:20-119, device-resident channels-last output (CPU tensors here)"""
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tools"))
    from gen_golden import FakeLSeg
    from avlmaps_amd.utils.lseg_utils import get_lseg_feat
    g = golden("g5_lseg_protocol.npz")
    for name in ("pad_short", "grid_2x3", "tall"):
        crop, base = (int(x) for x in g[f"{name}_cfg"])
        ref = g[f"{name}_feat"]                                   # (1, D, Hf, Wf) as the reference returns it
        f = get_lseg_feat(FakeLSeg(), g[f"{name}_img"], ["example"], None, "cpu", crop, base)
        assert tuple(f.shape) == (ref.shape[2], ref.shape[3], ref.shape[1]) and f.is_contiguous()
        np.testing.assert_allclose(f.numpy(), np.transpose(ref[0], (1, 2, 0)), rtol=1e-6, atol=1e-6)
        f2 = get_lseg_feat(FakeLSeg(), g[f"{name}_img"], ["example"], None, "cpu", crop, base, channels_last=False)
        np.testing.assert_allclose(f2.numpy(), ref, rtol=1e-6, atol=1e-6)
