#!/usr/bin/env python3
"""Generate golden vectors by EXECUTING the upstream reference in this container.

Run:  python tools/gen_golden.py            (needs /root/reference; never runs on the GPU box)
Writes small .npz fixtures (inputs + the reference's outputs) to tests/golden/.
The fixtures are data only; no reference source text is stored.

Reference entry points exercised (file:line in /root/reference):
  G1  avlmaps/utils/mapping_utils.py:18-26 cvt_pose_vec2tf, :226-251 depth2pc, :305-315 transform_pc,
      :345-349 base_pos2grid_id_3d, :591-596 get_sim_cam_mat, :599-605 project_point
  G2  avlmaps/map/vlmap_builder.py:54-185 VLMapBuilder.create_mobile_base_map (the real loop;
      only disk/model I/O is replaced by in-memory fakes), avlmaps/map/map.py:54-68 _setup_transforms
  G3  avlmaps/utils/clip_utils.py:196-242 get_lseg_score, avlmaps/map/vlmap.py:92-125
      VLMap.init_categories / index_map
  G4  avlmaps/utils/visualize_utils.py:29-49 get_heatmap_from_mask_3d
  G8  the 2-D consumers of the map: avlmaps/utils/visualize_utils.py:77-83 pool_3d_label_to_2d,
      avlmaps/map/map.py:79-113 Map.generate_obstacle_map / generate_cropped_obstacle_map / generate_rgb_topdown_map,
      avlmaps/utils/index_utils.py:138-184 get_dynamic_obstacles_map_3d, avlmaps/map/vlmap.py:158-187 VLMap.get_pos
      (up to the cv2.findContours call, which is stubbed: its input mask is recorded)
  G9  BASELINE config 1 at its stated size (50 000 x 512, one landmark + "other"): clip_utils.py:196-242 get_lseg_score and
      vlmap.py:104-125 VLMap.index_map(with_init_cat=False); the map is regenerated from its seed, outputs only are stored
"""
import os
import sys
import tempfile
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
from ref_import import import_reference  # noqa: E402

OUT = HERE.parent / "tests" / "golden"


class Cfg(dict):
    """attr+item access config, stands in for omegaconf.DictConfig"""
    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError as e:
            raise AttributeError(k) from e
        return v


def make_map_config(gs, cs, cam_h, calib, rate):
    return Cfg(
        map_type="vlmap",
        pose_info=Cfg(
            pose_type="mobile_base", camera_height=cam_h,
            base2cam_rot=[1, 0, 0, 0, -1, 0, 0, 0, -1],
            base_forward_axis=[0, 0, -1], base_left_axis=[-1, 0, 0], base_up_axis=[0, 1, 0],
        ),
        cam_calib_mat=list(map(float, calib)), grid_size=gs, cell_size=cs, depth_sample_rate=rate,
    )


# --------------------------------------------------------------------------------------
def gen_g1(m, rng):
    mu = m["mapping_utils"]
    out = {}
    # poses
    quats = rng.standard_normal((16, 4))
    quats /= np.linalg.norm(quats, axis=1, keepdims=True)
    pos = rng.uniform(-3, 3, (16, 3))
    posevecs = np.concatenate([pos, quats], axis=1)
    out["posevecs"] = posevecs
    out["pose_tfs"] = np.stack([mu.cvt_pose_vec2tf(p) for p in posevecs])

    # voxel ids incl. edge cases
    gs, cs = 1000, 0.05
    pts = rng.uniform(-30, 30, (1500, 3))
    pts[:, 2] = rng.uniform(-0.3, 1.8, 1500)
    k = rng.integers(-520, 520, (300, 3)).astype(np.float64)
    edge = k * cs                               # exactly-integral multiples of the cell size
    edge2 = np.nextafter(edge, np.inf)
    edge3 = np.nextafter(edge, -np.inf)
    neg = rng.uniform(-cs, 0, (100, 3))          # (-cs,0) truncates toward zero -> index 0
    allp = np.concatenate([pts, edge, edge2, edge3, neg,
                           np.array([[0.0, 0.0, 0.0], [-0.0, 25.0, 1.5], [24.999999, -25.0, 1.4999999]])])
    ids = np.array([mu.base_pos2grid_id_3d(gs, cs, p[0], p[1], p[2]) for p in allp], dtype=np.int64)
    out["vox_gs"] = gs
    out["vox_cs"] = cs
    out["vox_pts"] = allp
    out["vox_ids"] = ids

    # project_point with the two camera matrices the builder uses
    calib = np.array([540, 0, 540, 0, 540, 360, 0, 0, 1.0]).reshape(3, 3)
    simcam = mu.get_sim_cam_mat(347, 520)
    out["simcam_347_520"] = simcam
    pl = rng.uniform(-4, 4, (1200, 3))
    pl[:, 2] = rng.uniform(0.1, 6, 1200)
    # reproduce the builder's access pattern: p is a row of pc.T
    pc = np.ascontiguousarray(pl.T)
    pr = np.array([mu.project_point(calib, p) for p in pc.T])
    ps = np.array([mu.project_point(simcam, p) for p in pc.T])
    out["proj_pts"] = pl
    out["proj_calib"] = calib
    out["proj_calib_xyz"] = pr
    out["proj_sim_xyz"] = ps

    # depth2pc on a small image + transform_pc
    H, W = 20, 28
    depth = rng.uniform(0.0, 7.0, (H, W)).astype(np.float32)
    depth[0, 0] = 0.1
    depth[0, 1] = 6.0
    depth[0, 2] = np.float32(0.1) + np.float32(1e-6)
    K = np.array([W / 2, 0, W / 2, 0, W / 2, H / 2, 0, 0, 1.0]).reshape(3, 3)
    pc, mask = mu.depth2pc(depth, intr_mat=K, min_depth=0.1, max_depth=6)
    out["d2p_depth"] = depth
    out["d2p_K"] = K
    out["d2p_pc"] = pc
    out["d2p_mask"] = mask
    T = out["pose_tfs"][3]
    out["tpc_T"] = T
    out["tpc_out"] = mu.transform_pc(pc, T)
    np.savez_compressed(OUT / "g1_geometry.npz", **out)
    print("G1 written", {k: np.asarray(v).shape for k, v in out.items()})


# --------------------------------------------------------------------------------------
def run_reference_builder(m, cfg, poses, rgbs, depths, feats, seed):
    """Run the REAL VLMapBuilder.create_mobile_base_map with in-memory I/O fakes.
    Returns dict with the final map arrays + the per-frame sampled pixel lists."""
    vb = m["vlmap_builder"]
    Map = m["map"].Map
    tmp = Path(tempfile.mkdtemp(prefix="avl_golden_"))
    pose_path = tmp / "poses.txt"
    np.savetxt(pose_path, poses)           # same text round-trip a real dataset has
    poses_rt = np.loadtxt(pose_path)
    nfr = len(rgbs)
    rgb_paths = [tmp / f"{i:06d}.png" for i in range(nfr)]
    depth_paths = [tmp / f"{i:06d}.npy" for i in range(nfr)]

    mp = Map(cfg)  # real _setup_transforms
    builder = vb.VLMapBuilder(tmp, cfg, pose_path, rgb_paths, depth_paths, mp.base2cam_tf, mp.base_transform)

    D = feats[0].shape[1]
    captured = {}
    samples = []

    def fake_imread(p):
        i = int(Path(p).stem)
        return rgbs[i][:, :, ::-1].copy()          # "bgr"

    def fake_cvt(bgr, code):
        return bgr[:, :, ::-1].copy()

    def fake_depth(p):
        return depths[int(Path(p).stem)]

    frame_counter = {"i": 0}

    def fake_lseg(*a, **k):
        f = feats[frame_counter["i"]]
        frame_counter["i"] += 1
        return f

    def fake_init_lseg(self):
        self.device = "cpu"
        self.clip_feat_dim = D
        return None, None, 480, 520, [0.5] * 3, [0.5] * 3

    def fake_save(self, grid_feat, grid_pos, weight, grid_rgb, occupied_ids, mapped_iter_set, max_id):
        captured.update(
            grid_feat=np.array(grid_feat[:max_id]), grid_pos=np.array(grid_pos[:max_id]),
            weight=np.array(weight[:max_id]), grid_rgb=np.array(grid_rgb[:max_id]),
            occupied_ids=np.array(occupied_ids), mapped_iter_list=np.array(sorted(mapped_iter_set), dtype=np.int32),
            max_id=max_id,
        )

    orig_shuffle = np.random.shuffle
    rate = cfg.depth_sample_rate

    def rec_shuffle(x):
        orig_shuffle(x)
        samples.append(np.array(x[::rate], dtype=np.int32))

    vb.cv2.imread = fake_imread
    vb.cv2.cvtColor = fake_cvt
    vb.load_depth_npy = fake_depth
    vb.get_lseg_feat = fake_lseg
    vb.VLMapBuilder._init_lseg = fake_init_lseg
    vb.VLMapBuilder._save_3d_map = fake_save
    vb.tqdm = lambda it, **k: _NoBar(it)
    np.random.seed(seed)
    np.random.shuffle = rec_shuffle
    try:
        builder.create_mobile_base_map()
    finally:
        np.random.shuffle = orig_shuffle
    captured["samples"] = samples
    captured["poses_rt"] = poses_rt
    captured["base2cam_tf"] = mp.base2cam_tf
    captured["base_transform"] = mp.base_transform
    return captured


class _NoBar:
    def __init__(self, it):
        self.it = it

    def __iter__(self):
        return iter(self.it)

    def set_description(self, *a, **k):
        pass


def synth_sequence(rng, nfr, H, W, Hf, Wf, D, feat_scale=14.2857, dbase=2.5, damp=1.5):
    """smooth-ish depth scene, random rgb, smooth trajectory, random unit features * LSeg logit scale"""
    yy, xx = np.meshgrid(np.linspace(-1, 1, H), np.linspace(-1, 1, W), indexing="ij")
    depths, rgbs, feats, poses = [], [], [], []
    for i in range(nfr):
        d = dbase + damp * np.sin(2.0 * xx + 0.3 * i) * np.cos(1.5 * yy) + 0.3 * damp * yy
        d = d + rng.normal(0, 0.02, d.shape)
        d[rng.random(d.shape) < 0.03] = 0.0           # holes
        d[rng.random(d.shape) < 0.02] = 8.0           # beyond max_depth
        depths.append(d.astype(np.float32))
        rgbs.append(rng.integers(0, 256, (H, W, 3), dtype=np.uint8))
        f = rng.standard_normal((1, D, Hf, Wf)).astype(np.float32)
        f /= np.linalg.norm(f, axis=1, keepdims=True)
        # LSeg emits logit_scale * unit vector computed in fp16 then cast (lseg_net.py:318-324)
        f = (f * feat_scale).astype(np.float16).astype(np.float32)
        feats.append(f)
        yaw = 0.15 * i
        # habitat pose: y up; rotate about y, translate in x/z
        q = np.array([0.0, np.sin(yaw / 2), 0.0, np.cos(yaw / 2)])
        p = np.array([0.2 * i, 0.05, -0.1 * i])
        poses.append(np.concatenate([p, q]))
    return depths, rgbs, feats, np.array(poses)


def gen_g2(m, rng):
    # (a) regular small scene
    H, W, Hf, Wf, D, nfr = 48, 64, 23, 31, 16, 6
    calib = [W / 2, 0, W / 2, 0, W / 2, H / 2, 0, 0, 1]
    cfg = make_map_config(gs=200, cs=0.05, cam_h=1.5, calib=calib, rate=7)
    depths, rgbs, feats, poses = synth_sequence(rng, nfr, H, W, Hf, Wf, D)
    res = run_reference_builder(m, cfg, poses, rgbs, depths, feats, seed=1234)
    save_builder_fixture("g2a_builder_small.npz", cfg, depths, rgbs, feats, poses, res)
    # (b) coarse grid so that the reference's capacity doubling (_reserve_map_space) triggers:
    #     initial capacity = gs*gs = 400 rows < number of voxels
    cfg = make_map_config(gs=20, cs=0.1, cam_h=1.7, calib=calib, rate=3)
    depths, rgbs, feats, poses = synth_sequence(np.random.default_rng(5), 16, H, W, Hf, Wf, 8, dbase=0.7, damp=0.3)
    res = run_reference_builder(m, cfg, poses, rgbs, depths, feats, seed=99)
    save_builder_fixture("g2b_builder_growth.npz", cfg, depths, rgbs, feats, poses, res)


def save_builder_fixture(name, cfg, depths, rgbs, feats, poses, res):
    occ = res["occupied_ids"]
    nz = np.argwhere(occ != -1).astype(np.int32)
    out = dict(
        gs=cfg.grid_size, cs=cfg.cell_size, camera_height=cfg.pose_info.camera_height,
        rate=cfg.depth_sample_rate, calib=np.array(cfg.cam_calib_mat, dtype=np.float64),
        base2cam_rot=np.array(cfg.pose_info.base2cam_rot, dtype=np.float64),
        base_axes=np.array([cfg.pose_info.base_forward_axis, cfg.pose_info.base_left_axis,
                            cfg.pose_info.base_up_axis], dtype=np.float64),
        depths=np.stack(depths), rgbs=np.stack(rgbs), feats=np.concatenate(feats, 0),
        poses=poses, poses_rt=res["poses_rt"],
        base2cam_tf=res["base2cam_tf"], base_transform=res["base_transform"],
        samples=np.stack(res["samples"]),
        grid_feat=res["grid_feat"], grid_pos=res["grid_pos"], weight=res["weight"],
        grid_rgb=res["grid_rgb"], occ_shape=np.array(occ.shape), occ_nz=nz,
        occ_nz_vals=occ[nz[:, 0], nz[:, 1], nz[:, 2]].astype(np.int32),
        mapped_iter_list=res["mapped_iter_list"], max_id=res["max_id"],
        numpy_version=np.__version__,
    )
    np.savez_compressed(OUT / name, **out)
    print(name, "voxels", res["max_id"], "weight dtype", res["weight"].dtype, "rgb dtype", res["grid_rgb"].dtype,
          "pts/frame", [len(s) for s in res["samples"]])


# --------------------------------------------------------------------------------------
def gen_g3(m, rng):
    cu = m["clip_utils"]
    VLMap = m["vlmap"].VLMap
    ntmpl = len(cu.multiple_templates)
    out = {"n_templates": ntmpl}
    N, D = 1024, 512
    feat = rng.standard_normal((N, D)).astype(np.float32)
    feat *= (rng.uniform(0.2, 14.2857, (N, 1)) / np.linalg.norm(feat, axis=1, keepdims=True)).astype(np.float32)
    feat[100] = feat[7]                  # duplicate rows
    feat[200] = 0.0                      # all-zero row -> all scores tie at 0 -> argmax 0
    out["feat"] = feat

    table = {}

    def fake_text_feats(in_text, clip_model, clip_feat_dim, batch_size=64):
        r = np.zeros((len(in_text), clip_feat_dim), dtype=np.float32)
        for i, t in enumerate(in_text):
            r[i] = table[t]
        return r

    cu.get_text_feats = fake_text_feats
    m["vlmap"].get_lseg_score = cu.get_lseg_score

    def register(landmarks):
        for lm in landmarks + ["other"]:
            for t in cu.multiple_templates:
                s = t.format(lm)
                if s not in table:
                    v = rng.standard_normal(D).astype(np.float32)
                    table[s] = v / np.linalg.norm(v)
            if lm not in table:
                v = rng.standard_normal(D).astype(np.float32)
                table[lm] = v / np.linalg.norm(v)

    cases = {
        "q1": ["sofa"],
        "q2": ["chair", "table"],
        "q64": [f"thing{i}" for i in range(64)],
        "q40_other_last": [f"cat{i}" for i in range(39)] + ["other"],
    }
    for name, lms in cases.items():
        register(lms)
        sc = cu.get_lseg_score(None, list(lms), feat, D, use_multiple_templates=True, add_other=True)
        lm_other = lms if lms[-1] == "other" else lms + ["other"]
        tf = np.stack([np.stack([table[t.format(lm)] for t in cu.multiple_templates]) for lm in lm_other])
        if len(lm_other) <= 3:
            out[f"{name}_template_feats"] = tf.astype(np.float32)      # (Q, 63, D): pins the template mean
        # what the reference reduces the templates to (clip_utils.py:223-225: reshape + np.mean(axis=1))
        out[f"{name}_mean_feats"] = np.mean(tf.astype(np.float32), axis=1)
        out[f"{name}_scores"] = sc
        out[f"{name}_argmax"] = np.argmax(sc, axis=1).astype(np.int32)
        # single-template path
        sc1 = cu.get_lseg_score(None, list(lms), feat, D, use_multiple_templates=False, add_other=True)
        out[f"{name}_single_feats"] = np.stack([table[lm] for lm in lm_other]).astype(np.float32)
        out[f"{name}_single_scores"] = sc1
    # exact-tie case: two identical query columns -> first index must win
    register(["dup"])
    tf = np.stack([table[t.format("dup")] for t in cu.multiple_templates]).mean(0)
    q = np.stack([tf, tf, -tf]).astype(np.float32)
    sc = feat @ q.T
    out["tie_queries"] = q
    out["tie_scores"] = sc
    out["tie_argmax"] = np.argmax(sc, axis=1).astype(np.int32)

    # VLMap.index_map (real class) without preloaded categories
    cfg = make_map_config(1000, 0.05, 1.5, [540, 0, 540, 0, 540, 360, 0, 0, 1], 100)
    vm = VLMap(cfg)
    vm.grid_feat = feat
    vm.clip_model = None
    vm.clip_feat_dim = D
    out["index_map_sofa_mask"] = vm.index_map("sofa", with_init_cat=False)
    sm = vm.init_categories(list(cases["q40_other_last"]))
    out["init_categories_scores"] = sm
    np.savez_compressed(OUT / "g3_similarity.npz", **out)
    print("G3 written; Q64 scores", out["q64_scores"].shape, out["q64_scores"].dtype)


# --------------------------------------------------------------------------------------
def gen_g4(m, rng):
    vu = m["visualize_utils"]
    vu.tqdm = lambda it, **k: it
    N = 2000
    pos = np.stack([rng.integers(400, 470, N), rng.integers(400, 470, N), rng.integers(0, 30, N)], 1).astype(np.int32)
    pos = np.unique(pos, axis=0)
    rng.shuffle(pos)
    mask = rng.random(len(pos)) < 0.04
    out = {"grid_pos": pos, "mask": mask}
    for decay in (0.01, 0.1):
        out[f"heat_{decay}"] = vu.get_heatmap_from_mask_3d(pos, mask, cell_size=0.05, decay_rate=decay)
    np.savez_compressed(OUT / "g4_heatmap.npz", **out)
    print("G4 written", len(pos), mask.sum())


def gen_g8(m, rng):
    """2-D consumers (see module docstring).  One synthetic map: unique cells, several voxels per (row, col) column, ids in
    random spatial order (so 'last writer wins' in generate_rgb_topdown_map is not the spatially last voxel)."""
    cu, iu, vu = m["clip_utils"], m["index_utils"], m["visualize_utils"]
    Map, VLMap = m["map"].Map, m["vlmap"].VLMap
    gs, vh, cs, D = 64, 30, 0.05, 64
    cand = np.stack(np.meshgrid(np.arange(9, 52), np.arange(5, 61), np.arange(vh), indexing="ij"), -1).reshape(-1, 3)
    keep = rng.random(len(cand)) < 0.045
    # a few columns get many voxels, most get one or none
    pos = cand[keep].astype(np.int32)
    rng.shuffle(pos)
    N = len(pos)
    occ = -np.ones((gs, gs, vh), dtype=np.int32)
    occ[pos[:, 0], pos[:, 1], pos[:, 2]] = np.arange(N, dtype=np.int32)
    rgb = rng.integers(0, 256, (N, 3)).astype(np.uint8)
    feat = rng.standard_normal((N, D)).astype(np.float32)
    feat *= (rng.uniform(0.5, 14.2857, (N, 1)) / np.linalg.norm(feat, axis=1, keepdims=True)).astype(np.float32)
    out = dict(gs=gs, vh=vh, cs=cs, grid_pos=pos, occupied_ids_nz=np.argwhere(occ >= 0).astype(np.int32),
               occupied_ids_vals=occ[occ >= 0], grid_rgb=rgb, grid_feat=feat)
    # pool_3d_label_to_2d
    for name, frac in (("sparse", 0.03), ("dense", 0.5), ("none", 0.0)):
        mask = rng.random(N) < frac
        out[f"mask3d_{name}"] = mask
        out[f"mask2d_{name}"] = vu.pool_3d_label_to_2d(mask, pos, gs)
    # Map.generate_obstacle_map (+ crop) and rgb top-down
    cfg = make_map_config(gs, cs, 1.5, [540, 0, 540, 0, 540, 360, 0, 0, 1], 100)
    mp = Map(cfg)
    mp.occupied_ids, mp.grid_pos, mp.grid_rgb = occ, pos, rgb
    for tag, (h0, h1) in (("default", (0, 1.5)), ("band", (0.3, 1.0))):
        om = mp.generate_obstacle_map(h0, h1)
        out[f"obstacles_{tag}"] = om
        out[f"obstacles_{tag}_crop"] = np.array([mp.rmin, mp.rmax, mp.cmin, mp.cmax], dtype=np.int64)
        out[f"obstacles_{tag}_cropped"] = mp.obstacles_cropped
    out["rgb_topdown"] = mp.generate_rgb_topdown_map()
    # get_dynamic_obstacles_map_3d with deterministic text features
    potential = ["chair", "wall", "wall above the door", "table", "window", "floor", "stairs", "other"]
    obstacle_names = ["wall", "chair", "table", "window", "stairs", "other"]
    table = {}

    def fake_text_feats(in_text, clip_model, clip_feat_dim, batch_size=64):
        r = np.zeros((len(in_text), clip_feat_dim), dtype=np.float32)
        for i, t in enumerate(in_text):
            if t not in table:
                v = np.random.default_rng(abs(hash(t)) % (2 ** 32)).standard_normal(clip_feat_dim).astype(np.float32)
                table[t] = v / np.linalg.norm(v)
            r[i] = table[t]
        return r

    # deterministic across runs: derive the vectors from a seeded generator in registration order instead of hash()
    reg = np.random.default_rng(808)
    for lm in potential:
        for t in cu.multiple_templates:
            v = reg.standard_normal(D).astype(np.float32)
            table[t.format(lm)] = v / np.linalg.norm(v)
    cu.get_text_feats = fake_text_feats
    iu.get_text_feats = fake_text_feats
    iu.get_lseg_score = iu.get_lseg_score          # the module's own duplicate (index_utils.py:64-108) is what :150 calls
    mp.generate_obstacle_map(0, 1.5)
    new_obs = iu.get_dynamic_obstacles_map_3d(None, mp.obstacles_cropped, potential, obstacle_names, feat, pos, mp.rmin, mp.cmin, D)
    out["dyn_potential"] = np.array(potential)
    out["dyn_obstacle_names"] = np.array(obstacle_names)
    out["dyn_mean_feats"] = np.stack([np.mean(np.stack([table[t.format(lm)] for t in cu.multiple_templates]), axis=0) for lm in potential])
    out["dyn_new_obstacles"] = new_obs
    sc = iu.get_lseg_score(None, potential, feat, D, use_multiple_templates=True, avg_mode=0)
    out["dyn_scores"] = sc
    out["dyn_predict"] = np.argmax(sc, axis=1).astype(np.int32)
    # get_lseg_score with avg_mode=1 (average of the per-template SCORES instead of the features, clip_utils.py:231-240)
    lms3 = potential[:3]
    out["avg1_landmarks"] = np.array(lms3 + ["other"])
    out["avg1_template_feats"] = np.stack([np.stack([table[t.format(lm)] for t in cu.multiple_templates]) for lm in lms3 + ["other"]])
    out["avg1_scores"] = cu.get_lseg_score(None, list(lms3), feat, D, use_multiple_templates=True, avg_mode=1)
    # VLMap.get_pos up to cv2.findContours (navigation_utils.get_segment_islands_pos is stubbed and records its input)
    vm = VLMap(cfg)
    vm.grid_feat, vm.grid_pos, vm.occupied_ids, vm.grid_rgb = feat, pos, occ, rgb
    vm.clip_model, vm.clip_feat_dim = None, D
    m["vlmap"].get_lseg_score = cu.get_lseg_score
    vm.generate_obstacle_map(0, 1.5)
    cats = potential[:-1]
    vm.init_categories(list(cats))
    seen = {}

    def fake_islands(segment_map, label_id, detect_internal_contours=False):
        seen["foreground"] = np.array(segment_map, copy=True)
        return [], [], [], None
    m["vlmap"].get_segment_islands_pos = fake_islands
    m["vlmap"].find_similar_category_id = lambda name, cats_: cats_.index(name)
    for name in ("wall", "table"):
        vm.get_pos(name)
        out[f"get_pos_{name}_foreground"] = seen["foreground"]
        out[f"get_pos_{name}_mask3d"] = vm.index_map(name, with_init_cat=True)
    out["get_pos_categories"] = np.array(cats)
    out["get_pos_scores_mat"] = vm.scores_mat
    np.savez_compressed(OUT / "g8_map2d.npz", **out)
    print("G8 written: N =", N, "obstacle cells", int((~out["obstacles_default"]).sum()), "dyn free", int(new_obs.sum()))


def main():
    OUT.mkdir(parents=True, exist_ok=True)
    if "--only-g8" in sys.argv:
        gen_g8(import_reference(), np.random.default_rng(88))
        return
    if "--only-g7" in sys.argv:
        gen_g7_wide(import_reference(), np.random.default_rng(77))
        return
    if "--only-g9" in sys.argv:
        gen_g9_config1(import_reference(), np.random.default_rng(99))
        return
    m = import_reference()
    gen_g1(m, np.random.default_rng(11))
    gen_g2(m, np.random.default_rng(22))
    gen_g3(m, np.random.default_rng(33))
    gen_g4(m, np.random.default_rng(44))
    gen_g8(m, np.random.default_rng(88))
    gen_templates_hash(m)
    gen_g5_lseg_protocol()
    gen_g6_multi_floor(np.random.default_rng(66))
    if "--only-g7" in sys.argv or "--all" in sys.argv or not (OUT / "g7_similarity_wide.npz").exists():
        gen_g7_wide(import_reference(), np.random.default_rng(77))
    if "--only-g9" in sys.argv or "--all" in sys.argv or not (OUT / "g9_config1.npz").exists():
        gen_g9_config1(import_reference(), np.random.default_rng(99))
    os.system(f"ls -la {OUT}")




def gen_g7_wide(m, rng):
    """avlmaps/utils/clip_utils.py:196-242 get_lseg_score (real function) on query sets wider than one LDS-resident image
    (100 and 128 columns at D = 512) and on a 1536-column fused visual|audio feature map (BASELINE config 5 shape family)"""
    cu = m["clip_utils"]
    table = {}

    def fake_text_feats(in_text, clip_model, clip_feat_dim, batch_size=64):
        r = np.zeros((len(in_text), clip_feat_dim), dtype=np.float32)
        for i, t in enumerate(in_text):
            r[i] = table[(t, clip_feat_dim)]
        return r

    cu.get_text_feats = fake_text_feats
    out = {}
    for tag, N, D, nq, blocks in (("d512_q100", 320, 512, 99, False), ("d512_q128", 320, 512, 127, False),
                                  ("d1536_q128", 192, 1536, 127, True)):
        feat = rng.standard_normal((N, D)).astype(np.float32)
        feat *= (rng.uniform(0.2, 14.2857, (N, 1)) / np.linalg.norm(feat, axis=1, keepdims=True)).astype(np.float32)
        lms = [f"{tag}_{i}" for i in range(nq)]
        for qi, lm in enumerate(lms + ["other"]):
            for t in cu.multiple_templates:
                v = rng.standard_normal(D).astype(np.float32)
                if blocks:                       # text queries live in the visual block, "audio" queries in the audio block
                    v[512:] = 0 if qi % 2 == 0 else v[512:]
                    v[:512] = v[:512] if qi % 2 == 0 else 0
                table[(t.format(lm), D)] = v / np.linalg.norm(v)
        sc = cu.get_lseg_score(None, list(lms), feat, D, use_multiple_templates=True, add_other=True)
        tf = np.stack([np.stack([table[(t.format(lm), D)] for t in cu.multiple_templates]) for lm in lms + ["other"]])
        out[f"{tag}_feat"] = feat
        out[f"{tag}_mean_feats"] = np.mean(tf.astype(np.float32), axis=1)
        out[f"{tag}_scores"] = sc
        out[f"{tag}_argmax"] = np.argmax(sc, axis=1).astype(np.int32)
    np.savez_compressed(OUT / "g7_similarity_wide.npz", **out)
    print("G7 written", {k: v.shape for k, v in out.items() if k.endswith("scores")})


def config1_inputs():
    """BASELINE config 1 / SURVEY 8(d): 50 000 x 512 standard-normal map from seed 0 (the tests regenerate it from the seed,
    only the reference's outputs are stored)"""
    return np.random.default_rng(0).standard_normal((50_000, 512)).astype(np.float32)


def gen_g9_config1(m, rng):
    """BASELINE config 1 AT ITS STATED SIZE through the reference itself: avlmaps/utils/clip_utils.py:196-242 get_lseg_score
    and avlmaps/map/vlmap.py:104-125 VLMap.index_map(with_init_cat=False) on the 50 000 x 512 map, one landmark (+ "other")."""
    cu = m["clip_utils"]
    VLMap = m["vlmap"].VLMap
    D = 512
    feat = config1_inputs()
    table = {}
    for lm in ("sofa", "other"):
        for t in list(cu.multiple_templates) + ["{}"]:
            v = rng.standard_normal(D).astype(np.float32)
            table[t.format(lm)] = v / np.linalg.norm(v)

    def fake_text_feats(in_text, clip_model, clip_feat_dim, batch_size=64):
        return np.stack([table[t] for t in in_text]).astype(np.float32)

    cu.get_text_feats = fake_text_feats
    m["vlmap"].get_lseg_score = cu.get_lseg_score
    out = {"feat_seed": 0, "feat_shape": np.array(feat.shape), "feat_crc_rows": feat[::997].sum(axis=1)}
    tf = np.stack([np.stack([table[t.format(lm)] for t in cu.multiple_templates]) for lm in ("sofa", "other")])
    out["template_feats"] = tf.astype(np.float32)
    out["mean_feats"] = np.mean(tf.astype(np.float32), axis=1)
    out["scores"] = cu.get_lseg_score(None, ["sofa"], feat, D, use_multiple_templates=True, add_other=True)
    out["single_feats"] = np.stack([table["sofa"], table["other"]]).astype(np.float32)
    out["single_scores"] = cu.get_lseg_score(None, ["sofa"], feat, D, use_multiple_templates=False, add_other=True)
    cfg = make_map_config(1000, 0.05, 1.5, [540, 0, 540, 0, 540, 360, 0, 0, 1], 100)
    vm = VLMap(cfg)
    vm.grid_feat, vm.clip_model, vm.clip_feat_dim = feat, None, D
    out["index_map_mask"] = vm.index_map("sofa", with_init_cat=False)
    assert out["scores"].shape == (50_000, 2) and out["scores"].dtype == np.float32
    np.savez_compressed(OUT / "g9_config1.npz", **out)
    print("G9 written", out["scores"].shape, "mask true", int(out["index_map_mask"].sum()))


def gen_templates_hash(m):
    """hash (not text) of the reference's prompt templates, clip_utils.py:10-74"""
    import hashlib
    import json
    ref = m["clip_utils"].multiple_templates
    h = hashlib.sha256("\n".join(ref).encode()).hexdigest()
    json.dump({"n_templates": len(ref), "sha256_of_newline_joined": h,
               "source": "avlmaps/utils/clip_utils.py:10-74 (hash only)"}, open(OUT / "templates.json", "w"), indent=1)



# --------------------------------------------------------------------------------------
class FakeLSeg:
    """deterministic stand-in for LSegEncNet: per-pixel features that depend on the pixel value AND on the position inside
    the crop, so that the sliding-window offsets / overlap averaging / padding of get_lseg_feat are all observable"""
    out_c = 6

    def __call__(self, x, labels):
        import torch
        b, c, h, w = x.shape
        yy = torch.linspace(0, 1, h).view(1, 1, h, 1).expand(b, 1, h, w)
        xx = torch.linspace(0, 1, w).view(1, 1, 1, w).expand(b, 1, h, w)
        f = torch.cat([x, x.mean(1, keepdim=True) * yy, xx * yy, torch.sin(3 * x[:, :1]) + xx], dim=1)
        logits = torch.cat([x[:, :1] * 0 + float(i) for i in range(len(labels))], dim=1)
        return f, logits


def gen_g5_lseg_protocol():
    """avlmaps/utils/lseg_utils.py:20-119 get_lseg_feat (real function, fake model, CPU)"""
    import importlib
    import torch
    for name in ["avlmaps.lseg.additional_utils.models", "avlmaps.utils.lseg_utils"]:
        sys.modules.pop(name, None)
    lu = importlib.import_module("avlmaps.utils.lseg_utils")
    rng = np.random.default_rng(55)

    def tfm(img):
        t = torch.from_numpy(np.ascontiguousarray(img)).permute(2, 0, 1).float().div(255.0)
        return (t - 0.5) / 0.5

    out = {}
    # (base_size <= crop_size is not covered: the reference itself raises UnboundLocalError on that branch, lseg_utils.py:103)
    cases = {"pad_short": (72, 108, 48, 52), "grid_2x3": (80, 100, 48, 100), "tall": (110, 60, 40, 90)}
    for name, (H, W, crop, base) in cases.items():
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        f = lu.get_lseg_feat(FakeLSeg(), img, ["example"], tfm, "cpu", crop, base, [0.5] * 3, [0.5] * 3)
        out[f"{name}_img"] = img
        out[f"{name}_cfg"] = np.array([crop, base])
        out[f"{name}_feat"] = f
    np.savez_compressed(OUT / "g5_lseg_protocol.npz", **out)
    print("G5 written", {k: v.shape for k, v in out.items() if k.endswith("_feat")})



# --------------------------------------------------------------------------------------
class _FakePCD:
    """minimal stand-in for open3d.geometry.PointCloud (points container with +=)"""

    def __init__(self):
        self.points = np.zeros((0, 3))

    def __iadd__(self, other):
        self.points = np.concatenate([np.asarray(self.points).reshape(-1, 3), np.asarray(other.points).reshape(-1, 3)], 0)
        return self


def gen_g6_multi_floor(rng):
    """avlmaps/map/vlmap_builder_multi_floor.py:60-199 VLMapBuilderMultiFloor.create_global_map (real loop, fake I/O)"""
    import importlib
    vbm = importlib.import_module("avlmaps.map.vlmap_builder_multi_floor")
    H, W, Hf, Wf, D, nfr = 24, 32, 11, 15, 8, 6
    calib = [W / 2, 0, W / 2, 0, W / 2, H / 2, 0, 0, 1]
    cfg = make_map_config(gs=1000, cs=0.1, cam_h=1.5, calib=calib, rate=1)
    cfg["skip_frame"] = 1
    cfg.pose_info["building_init_height"] = 0.0
    yy, xx = np.meshgrid(np.linspace(-1, 1, H), np.linspace(-1, 1, W), indexing="ij")
    depths, rgbs, feats, poses = [], [], [], []
    from scipy.spatial.transform import Rotation as R
    for i in range(nfr):
        d = 2.0 + 0.8 * np.sin(2 * xx + 0.4 * i) * np.cos(yy) + 0.3 * yy
        d[rng.random(d.shape) < 0.05] = 0.0
        depths.append(np.round(d * 1000).astype(np.uint16))
        rgbs.append(rng.integers(0, 256, (H, W, 3), dtype=np.uint8))
        f = rng.standard_normal((1, D, Hf, Wf)).astype(np.float32)
        feats.append((f / np.linalg.norm(f, axis=1, keepdims=True) * 14.2857).astype(np.float16).astype(np.float32))
        T = np.eye(4)
        T[:3, :3] = R.from_euler("yxz", [0.3 * i, 0.05 * i, 0.02 * i]).as_matrix()
        T[:3, 3] = [0.3 * i, 0.1 + 0.6 * (i // 3), -0.2 * i]          # second "floor" after 3 frames
        poses.append(T)
    tmp = Path(tempfile.mkdtemp(prefix="avl_golden_mf_"))
    pose_paths = []
    for i, T in enumerate(poses):
        p = tmp / f"{i:06d}.txt"
        np.savetxt(p, T)
        pose_paths.append(p)
    poses_rt = np.stack([np.loadtxt(p).reshape(4, 4) for p in pose_paths])
    rgb_paths = [tmp / f"{i:06d}.png" for i in range(nfr)]
    depth_paths = [tmp / f"{i:06d}_d.png" for i in range(nfr)]
    builder = vbm.VLMapBuilderMultiFloor(tmp, cfg, pose_paths, rgb_paths, depth_paths, None, None)
    captured, samples = {}, []
    counter = {"i": 0}

    def fake_lseg(*a, **k):
        f = feats[counter["i"]]
        counter["i"] += 1
        return f

    def fake_init_lseg(self):
        self.device, self.clip_feat_dim = "cpu", D
        return None, None, 480, 520, [0.5] * 3, [0.5] * 3

    def fake_save(self, grid_feat, grid_pos, weight, grid_rgb, occupied_ids, mapped_iter_set, max_id):
        captured.update(grid_feat=np.array(grid_feat[:max_id]), grid_pos=np.array(grid_pos[:max_id]),
                        weight=np.array(weight[:max_id]), grid_rgb=np.array(grid_rgb[:max_id]),
                        occupied_ids=np.array(occupied_ids), max_id=max_id, pcd_min=np.array(self.pcd_min),
                        pcd_max=np.array(self.pcd_max), grid_size=np.array(self.grid_size),
                        mapped_iter_list=np.array(sorted(mapped_iter_set), dtype=np.int32))

    orig_shuffle = np.random.shuffle

    def rec_shuffle(x):
        orig_shuffle(x)
        samples.append(np.array(x[::cfg.depth_sample_rate], dtype=np.int32))

    vbm.cv2.imread = lambda p: rgbs[int(Path(p).stem)][:, :, ::-1].copy()
    vbm.cv2.cvtColor = lambda bgr, code: bgr[:, :, ::-1].copy()
    vbm.load_depth_img = lambda p: depths[int(Path(p).stem.split("_")[0])]
    vbm.get_lseg_feat = fake_lseg
    vbm.VLMapBuilderMultiFloor._init_lseg = fake_init_lseg
    vbm.VLMapBuilderMultiFloor.save_3d_map = fake_save
    vbm.tqdm = lambda it, **k: _NoBar(it)
    vbm.o3d.geometry.PointCloud = _FakePCD
    vbm.o3d.utility.Vector3dVector = lambda a: np.asarray(a)
    np.random.seed(777)
    np.random.shuffle = rec_shuffle
    try:
        builder.create_global_map()
    finally:
        np.random.shuffle = orig_shuffle
    occ = captured["occupied_ids"]
    nz = np.argwhere(occ != -1).astype(np.int32)
    out = dict(cs=cfg.cell_size, rate=cfg.depth_sample_rate, calib=np.array(calib, dtype=np.float64), depths_u16=np.stack(depths),
               rgbs=np.stack(rgbs), feats=np.concatenate(feats, 0), poses=poses_rt,
               samples_pass1=np.stack(samples[:nfr]), samples_pass2=np.stack(samples[nfr:]),
               pcd_min=captured["pcd_min"], pcd_max=captured["pcd_max"], grid_size=captured["grid_size"],
               grid_feat=captured["grid_feat"], grid_pos=captured["grid_pos"], weight=captured["weight"],
               grid_rgb=captured["grid_rgb"], occ_shape=np.array(occ.shape), occ_nz=nz,
               occ_nz_vals=occ[nz[:, 0], nz[:, 1], nz[:, 2]].astype(np.int32), max_id=captured["max_id"])
    np.savez_compressed(OUT / "g6_multi_floor.npz", **out)
    print("G6 written: voxels", captured["max_id"], "grid", captured["grid_size"], "occ", occ.shape, "weight", captured["weight"].dtype)


if __name__ == "__main__":
    main()
