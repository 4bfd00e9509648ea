"""Pins the CPU oracle (oracle/) to golden vectors produced by executing the upstream reference
(tools/gen_golden.py).  CPU only."""
import numpy as np
import pytest

from oracle import avl_oracle as O


def test_cvt_pose_vec2tf(golden):
    g = golden("g1_geometry.npz")
    for v, tf in zip(g["posevecs"], g["pose_tfs"]):
        assert np.array_equal(O.cvt_pose_vec2tf(v), tf)


def test_base_pos2grid_id_3d_bit_exact(golden):
    g = golden("g1_geometry.npz")
    gs, cs = int(g["vox_gs"]), float(g["vox_cs"])
    got = np.array([O.base_pos2grid_id_3d(gs, cs, *p) for p in g["vox_pts"]])
    assert np.array_equal(got, g["vox_ids"])


def test_project_point_bit_exact(golden):
    g = golden("g1_geometry.npz")
    for K, ref in ((g["proj_calib"], g["proj_calib_xyz"]), (g["simcam_347_520"], g["proj_sim_xyz"])):
        got = np.array([O.project_point(K, p) for p in g["proj_pts"]])
        assert np.array_equal(got[:, :2], ref[:, :2])
        assert np.array_equal(got[:, 2], ref[:, 2])
    assert np.array_equal(O.get_sim_cam_mat(347, 520), g["simcam_347_520"])


def test_depth2pc_and_transform_bit_exact(golden):
    g = golden("g1_geometry.npz")
    depth, K = g["d2p_depth"], g["d2p_K"]
    pix = np.arange(depth.size)
    pc, mask = O.depth2pc_pixels(depth, np.linalg.inv(K), pix, 0.1, 6)
    assert np.array_equal(mask, g["d2p_mask"])
    assert np.array_equal(pc.T, g["d2p_pc"])
    assert np.array_equal(O.transform_points(g["tpc_T"], pc).T, g["tpc_out"])


def _run_oracle_builder(g):
    b2c, bt = O.setup_transforms(g["base2cam_rot"], float(g["camera_height"]), *g["base_axes"])
    assert np.array_equal(b2c, g["base2cam_tf"]) and np.array_equal(bt, g["base_transform"])
    Ts = O.pc_transforms(g["poses_rt"], bt, b2c)
    D = g["feats"].shape[1]
    m = O.OracleMap(int(g["gs"]), float(g["cs"]), float(g["camera_height"]), D)
    for i in range(len(g["depths"])):
        m.integrate(g["depths"][i], g["calib"], Ts[i], g["samples"][i], g["feats"][i], g["rgbs"][i])
    return m.export()


@pytest.mark.parametrize("name", ["g2a_builder_small.npz", "g2b_builder_growth.npz"])
def test_sequential_builder_matches_reference(golden, name):
    g = golden(name)
    out = _run_oracle_builder(g)
    assert len(out["grid_pos"]) == int(g["max_id"])
    assert np.array_equal(out["grid_pos"], g["grid_pos"])                    # voxel ids + id order: bit exact
    occ = out["occupied_ids"]
    assert tuple(occ.shape) == tuple(g["occ_shape"])
    nz = np.argwhere(occ != -1)
    assert np.array_equal(nz, g["occ_nz"])
    assert np.array_equal(occ[nz[:, 0], nz[:, 1], nz[:, 2]], g["occ_nz_vals"])
    assert out["weight"].dtype == g["weight"].dtype and out["grid_rgb"].dtype == g["grid_rgb"].dtype
    # same rounding sequence as the reference; exp() may differ by an ulp between libms
    np.testing.assert_allclose(out["weight"], g["weight"], rtol=2e-7, atol=0)
    np.testing.assert_allclose(out["grid_feat"], g["grid_feat"], rtol=1e-6, atol=1e-6)
    if g["grid_rgb"].dtype == np.uint8:
        assert np.abs(out["grid_rgb"].astype(int) - g["grid_rgb"].astype(int)).max() <= 1
    else:
        np.testing.assert_allclose(out["grid_rgb"], g["grid_rgb"], rtol=1e-5, atol=1e-4)
    frac_exact = np.mean(out["grid_feat"] == g["grid_feat"])
    assert frac_exact > 0.99, frac_exact


def test_growth_fixture_really_grew(golden):
    g = golden("g2b_builder_growth.npz")
    assert g["weight"].dtype == np.float64 and g["grid_rgb"].dtype == np.float32
    assert int(g["max_id"]) > int(g["gs"]) ** 2


def test_wide_query_similarity_matches_reference(golden):
    """g7: 100 / 128 query columns at D = 512 and a 1536-column fused map through the reference's get_lseg_score"""
    g = golden("g7_similarity_wide.npz")
    for tag, Q in (("d512_q100", 100), ("d512_q128", 128), ("d1536_q128", 128)):
        sc = O.sim_scores(g[f"{tag}_feat"], g[f"{tag}_mean_feats"])
        assert sc.shape == g[f"{tag}_scores"].shape and sc.shape[1] == Q
        np.testing.assert_allclose(sc, g[f"{tag}_scores"], rtol=0, atol=2e-5)
        sc2, am2 = O.sim_scores_scalar(g[f"{tag}_feat"], g[f"{tag}_mean_feats"])
        np.testing.assert_allclose(sc2, g[f"{tag}_scores"], rtol=0, atol=3e-5)
        ref = g[f"{tag}_scores"]
        assert np.all(ref[np.arange(len(ref)), am2] >= ref.max(axis=1) - 6e-5)


def test_similarity_matches_reference(golden):
    g = golden("g3_similarity.npz")
    feat = g["feat"]
    for name in ("q1", "q2"):
        tm = O.template_mean(g[f"{name}_template_feats"])
        assert np.array_equal(tm, g[f"{name}_mean_feats"])
    for name in ("q1", "q2", "q64", "q40_other_last"):
        sc = O.sim_scores(feat, g[f"{name}_mean_feats"])
        assert sc.dtype == np.float32 and sc.shape == g[f"{name}_scores"].shape
        np.testing.assert_allclose(sc, g[f"{name}_scores"], rtol=0, atol=1e-5)
        assert np.array_equal(np.argmax(sc, axis=1), g[f"{name}_argmax"])
        sc1 = O.sim_scores(feat, g[f"{name}_single_feats"])
        np.testing.assert_allclose(sc1, g[f"{name}_single_scores"], rtol=0, atol=1e-5)
        # scalar float64-accumulating port agrees within float32 round-off of the BLAS result
        sc2, _ = O.sim_scores_scalar(feat, g[f"{name}_mean_feats"])
        np.testing.assert_allclose(sc2, g[f"{name}_scores"], rtol=0, atol=2e-5)
    assert g["q1_scores"].shape[1] == 2 and g["q64_scores"].shape[1] == 65 and g["q40_other_last_scores"].shape[1] == 40
    mask, _ = O.argmax_mask(O.sim_scores(feat, g["q1_mean_feats"]), 0)
    assert np.array_equal(mask, g["index_map_sofa_mask"])
    # exact ties: first maximum wins
    sc = O.sim_scores(feat, g["tie_queries"])
    assert np.array_equal(np.argmax(sc, axis=1), g["tie_argmax"])
    _, am = O.sim_scores_scalar(feat, g["tie_queries"])
    assert np.array_equal(am, g["tie_argmax"])
    assert g["tie_argmax"][200] == 0

def config1_map():
    """BASELINE config 1's map: 50 000 x 512 standard normal float32 from seed 0 (tools/gen_golden.py:config1_inputs)"""
    return np.random.default_rng(0).standard_normal((50_000, 512)).astype(np.float32)


def test_config1_at_its_stated_size(golden):
    """BASELINE config 1 (50 000 x 512, one landmark + "other"; clip_utils.py:196-242, vlmap.py:104-125): the oracle against
    what the reference returned for the SAME seeded map (g9), at the size the config states -- the CPU plumbing case."""
    g = golden("g9_config1.npz")
    feat = config1_map()
    assert np.array_equal(feat[::997].sum(axis=1), g["feat_crc_rows"])          # the seed still produces the map g9 was made from
    assert np.array_equal(O.template_mean(g["template_feats"]), g["mean_feats"])
    sc = O.sim_scores(feat, g["mean_feats"])
    assert sc.shape == (50_000, 2) and sc.dtype == np.float32
    np.testing.assert_allclose(sc, g["scores"], rtol=0, atol=1e-6)
    mask, ids = O.argmax_mask(sc, 0)
    assert np.array_equal(mask, g["index_map_mask"]) and np.array_equal(ids, np.argmax(g["scores"], axis=1))
    np.testing.assert_allclose(O.sim_scores(feat, g["single_feats"]), g["single_scores"], rtol=0, atol=1e-5)
    sc2, am2 = O.sim_scores_scalar(feat, g["mean_feats"])
    np.testing.assert_allclose(sc2, g["scores"], rtol=0, atol=2e-6)
    gap = np.abs(g["scores"][:, 0] - g["scores"][:, 1])
    assert np.array_equal(am2[gap > 4e-6], ids[gap > 4e-6])



def test_heatmap_matches_reference(golden):
    g = golden("g4_heatmap.npz")
    for decay in (0.01, 0.1):
        h = O.heatmap_from_mask(g["grid_pos"], g["mask"], 0.05, decay)
        np.testing.assert_allclose(h, g[f"heat_{decay}"], rtol=0, atol=1e-6)
        assert np.array_equal(h == 1.0, g[f"heat_{decay}"] == 1.0)


def test_multi_floor_builder_matches_reference(golden):
    """vlmap_builder_multi_floor.py:60-199 (two passes: bounding box, then fusion with np.round voxel indices)"""
    g = golden("g6_multi_floor.npz")
    nfr = len(g["depths_u16"])
    depth_m = g["depths_u16"] / 1000.0                                   # :105 -- float64 metres
    minmax = np.array([np.inf] * 3 + [-np.inf] * 3)
    for i in range(nfr):
        O.points_bbox(minmax, depth_m[i], g["calib"], g["poses"][i] @ O.HABITAT2CAM_ROT, g["samples_pass1"][i])
    assert np.array_equal(minmax[:3], g["pcd_min"]) and np.array_equal(minmax[3:], g["pcd_max"])
    D = g["feats"].shape[1]
    m = O.OracleGlobalMap(minmax[:3], minmax[3:], float(g["cs"]), D)
    assert np.array_equal(m.grid_size, g["grid_size"])
    for i in range(nfr):
        m.integrate(depth_m[i], g["calib"], g["poses"][i] @ O.HABITAT2CAM_ROT, g["samples_pass2"][i], g["feats"][i], g["rgbs"][i])
    out = m.export()
    assert np.array_equal(out["grid_pos"], g["grid_pos"]) and len(out["grid_pos"]) == int(g["max_id"])
    occ = out["occupied_ids"]
    assert tuple(occ.shape) == tuple(g["occ_shape"])
    nz = np.argwhere(occ != -1)
    assert np.array_equal(nz, g["occ_nz"]) and np.array_equal(occ[nz[:, 0], nz[:, 1], nz[:, 2]], g["occ_nz_vals"])
    assert out["weight"].dtype == g["weight"].dtype == np.float64       # capacity doubled (n0*n1 rows < voxels)
    np.testing.assert_allclose(out["weight"], g["weight"], rtol=2e-7)
    np.testing.assert_allclose(out["grid_feat"], g["grid_feat"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(out["grid_rgb"], g["grid_rgb"], rtol=1e-5, atol=1e-4)


def test_oracle_map2d_matches_reference(golden):
    """G8: the oracle's restatements of pool_3d_label_to_2d, generate_obstacle_map (+ crop), generate_rgb_topdown_map and the
    scatter half of get_dynamic_obstacles_map_3d against outputs of the reference functions themselves"""
    g = golden("g8_map2d.npz")
    gs, vh, cs = int(g["gs"]), int(g["vh"]), float(g["cs"])
    pos = g["grid_pos"]
    occ = -np.ones((gs, gs, vh), np.int32)
    nz = g["occupied_ids_nz"]
    occ[nz[:, 0], nz[:, 1], nz[:, 2]] = g["occupied_ids_vals"]
    for name in ("sparse", "dense", "none"):
        assert np.array_equal(O.pool_3d_label_to_2d(g[f"mask3d_{name}"], pos, gs), g[f"mask2d_{name}"])
    for tag, band in (("default", (0, 1.5)), ("band", (0.3, 1.0))):
        om = O.obstacle_map(occ, cs, *band)
        assert np.array_equal(om, g[f"obstacles_{tag}"])
        assert list(O.crop_bounds(om)) == g[f"obstacles_{tag}_crop"].tolist()
    assert np.array_equal(O.rgb_topdown(pos, g["grid_rgb"], gs), g["rgb_topdown"])
    sc = O.sim_scores(g["grid_feat"], g["dyn_mean_feats"])
    np.testing.assert_allclose(sc, g["dyn_scores"], rtol=0, atol=1e-5)
    rmin, _, cmin, _ = g["obstacles_default_crop"].tolist()
    got = O.dynamic_obstacles(g["dyn_predict"], list(g["dyn_potential"]), list(g["dyn_obstacle_names"]), pos, rmin, cmin,
                              g["obstacles_default_cropped"])
    assert np.array_equal(got, g["dyn_new_obstacles"])
