"""CPU oracle (test infrastructure, NOT product code) for the AVLMaps map-build / index hot path.

Python face of oracle/avl_oracle.c (sequential, bit-faithful restatement of the reference loop) plus
NumPy restatements of the host-side pieces.  Parity is pinned by tests/test_oracle_golden.py against
fixtures produced by executing the upstream reference (tools/gen_golden.py).

Reference lines restated here (paths relative to the upstream repo root):
  avlmaps/utils/mapping_utils.py:18-26     cvt_pose_vec2tf
  avlmaps/map/map.py:54-68                 Map._setup_transforms
  avlmaps/map/vlmap_builder.py:64-76,106-108,133   pose chain -> pc_transform
  avlmaps/utils/mapping_utils.py:591-596   get_sim_cam_mat
  avlmaps/map/vlmap_builder.py:266-281     _backproject_depth sampling order
  avlmaps/utils/clip_utils.py:196-242      get_lseg_score (template mean, raw dot product)
  avlmaps/map/vlmap.py:123-124             argmax mask
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np
from scipy.spatial.transform import Rotation as R

_HERE = Path(__file__).resolve().parent
_LIB = None


def build(force: bool = False) -> Path:
    so = _HERE / "libavl_oracle.so"
    src = _HERE / "avl_oracle.c"
    if force or not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
        subprocess.check_call(["make", "-C", str(_HERE), "-s", "-B"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(str(build()))
        dp, fp, ip, up = (C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_uint8))
        llp = C.POINTER(C.c_longlong)
        L.avlo_base_pos2grid_id_3d.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, llp]
        L.avlo_project_point.argtypes = [dp, dp, llp, llp, dp]
        L.avlo_depth2pc_pixel.argtypes = [fp, C.c_int, dp, C.c_int, C.c_double, C.c_double, dp]
        L.avlo_depth2pc_pixel.restype = C.c_int
        L.avlo_transform_point.argtypes = [dp, dp, dp]
        L.avlo_map_create.argtypes = [C.c_int, C.c_double, C.c_int, C.c_int]
        L.avlo_map_create.restype = C.c_void_p
        L.avlo_map_create_grid.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_int]
        L.avlo_map_create_grid.restype = C.c_void_p
        L.avlo_integrate_frame_global.argtypes = [C.c_void_p, dp, C.c_int, C.c_int, dp, dp, dp, dp, ip, C.c_int, fp, C.c_int,
                                                  C.c_int, up, C.c_double, C.c_double, dp]
        L.avlo_integrate_frame_global.restype = C.c_longlong
        L.avlo_points_bbox.argtypes = [dp, C.c_int, dp, dp, ip, C.c_int, C.c_double, C.c_double, dp]
        L.avlo_map_destroy.argtypes = [C.c_void_p]
        L.avlo_integrate_frame.argtypes = [C.c_void_p, fp, C.c_int, C.c_int, dp, dp, dp, dp, ip, C.c_int, fp,
                                           C.c_int, C.c_int, up, C.c_double, C.c_double]
        L.avlo_integrate_frame.restype = C.c_longlong
        L.avlo_map_size.argtypes = [C.c_void_p]
        L.avlo_map_size.restype = C.c_longlong
        L.avlo_map_grown.argtypes = [C.c_void_p]
        L.avlo_map_grown.restype = C.c_int
        L.avlo_map_export.argtypes = [C.c_void_p, fp, ip, dp, dp, ip]
        L.avlo_sim_scores.argtypes = [fp, C.c_longlong, C.c_int, fp, C.c_int, fp, ip]
        L.avlo_heatmap_from_mask.argtypes = [ip, up, C.c_longlong, C.c_double, C.c_double, fp]
        _LIB = L
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


# ------------------------------------------------------------------ host-side geometry (float64)
def cvt_pose_vec2tf(v):
    """mapping_utils.py:18-26: (px,py,pz,qx,qy,qz,qw) -> 4x4."""
    tf = np.eye(4)
    v = np.asarray(v, dtype=np.float64)
    tf[:3, 3] = v[:3]
    tf[:3, :3] = R.from_quat(v[3:]).as_matrix()
    return tf


def setup_transforms(base2cam_rot, camera_height, fwd, left, up):
    """map.py:54-68."""
    b2c = np.eye(4)
    b2c[:3, :3] = np.array([base2cam_rot]).reshape((3, 3))
    b2c[1, 3] = camera_height
    bt = np.eye(4)
    bt[0, :3] = fwd
    bt[1, :3] = left
    bt[2, :3] = up
    return b2c, bt


def pc_transforms(poses, base_transform, base2cam_tf):
    """vlmap_builder.py:64-76 (init frame), :106-108 (per-frame tf), :133 (pc_transform)."""
    inv_bt = np.linalg.inv(base_transform)
    init_base_tf = base_transform @ cvt_pose_vec2tf(poses[0]) @ inv_bt
    inv_init = np.linalg.inv(init_base_tf)
    out = []
    for pv in poses:
        base_pose = base_transform @ cvt_pose_vec2tf(pv) @ np.linalg.inv(base_transform)
        tf = inv_init @ base_pose
        out.append(tf @ base_transform @ base2cam_tf)
    return np.stack(out)


def get_sim_cam_mat(h, w):
    """mapping_utils.py:591-596."""
    m = np.eye(3)
    m[0, 0] = m[1, 1] = w / 2.0
    m[0, 2] = w / 2.0
    m[1, 2] = h / 2.0
    return m


# ------------------------------------------------------------------ point-level functions (C)
def base_pos2grid_id_3d(gs, cs, x, y, z):
    out = (C.c_longlong * 3)()
    lib().avlo_base_pos2grid_id_3d(int(gs), float(cs), float(x), float(y), float(z), out)
    return [out[0], out[1], out[2]]


def project_point(K, p):
    K = np.ascontiguousarray(K, dtype=np.float64)
    p = np.ascontiguousarray(p, dtype=np.float64)
    px, py, pz = C.c_longlong(), C.c_longlong(), C.c_double()
    lib().avlo_project_point(_p(K, C.c_double), _p(p, C.c_double), C.byref(px), C.byref(py), C.byref(pz))
    return px.value, py.value, pz.value


def depth2pc_pixels(depth, Kinv, pix, min_depth, max_depth):
    depth = np.ascontiguousarray(depth, dtype=np.float32)
    Kinv = np.ascontiguousarray(Kinv, dtype=np.float64)
    pc = np.zeros((len(pix), 3))
    mask = np.zeros(len(pix), dtype=bool)
    tmp = (C.c_double * 3)()
    for i, s in enumerate(pix):
        mask[i] = bool(lib().avlo_depth2pc_pixel(_p(depth, C.c_float), depth.shape[1], _p(Kinv, C.c_double), int(s),
                                                 min_depth, max_depth, tmp))
        pc[i] = tmp[:]
    return pc, mask


def transform_points(T, pts):
    T = np.ascontiguousarray(T, dtype=np.float64)
    out = np.zeros_like(pts, dtype=np.float64)
    tmp = (C.c_double * 3)()
    for i, p in enumerate(np.ascontiguousarray(pts, dtype=np.float64)):
        lib().avlo_transform_point(_p(T, C.c_double), _p(p, C.c_double), tmp)
        out[i] = tmp[:]
    return out


# ------------------------------------------------------------------ sequential builder
class OracleMap:
    """Sequential reference-order map builder (vlmap_builder.py:102-183)."""

    def __init__(self, gs, cs, camera_height, D):
        self.gs, self.cs, self.D = int(gs), float(cs), int(D)
        self.vh = int(camera_height / cs)                       # vlmap_builder.py:201
        self._h = lib().avlo_map_create(self.gs, self.cs, self.vh, self.D)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().avlo_map_destroy(self._h)
            self._h = None

    def integrate(self, depth, calib, pc_transform, sample_idx, feat_chw, rgb, min_depth=0.1, max_depth=6.0, calib_inv=None):
        depth = np.ascontiguousarray(depth, dtype=np.float32)
        H, W = depth.shape
        K = np.ascontiguousarray(np.asarray(calib, dtype=np.float64).reshape(3, 3))
        # mapping_utils.py:237; tests may inject an exact camera-frame point through the inverse (tests/test_geometry_gpu.py)
        Kinv = np.ascontiguousarray(np.linalg.inv(K) if calib_inv is None else np.asarray(calib_inv, dtype=np.float64).reshape(3, 3))
        feat = np.ascontiguousarray(feat_chw, dtype=np.float32)
        if feat.ndim == 4:
            feat = feat[0]
        Dd, Hf, Wf = feat.shape
        assert Dd == self.D
        Kf = np.ascontiguousarray(get_sim_cam_mat(Hf, Wf))      # vlmap_builder.py:126
        T = np.ascontiguousarray(pc_transform, dtype=np.float64)
        idx = np.ascontiguousarray(sample_idx, dtype=np.int32)
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        n = lib().avlo_integrate_frame(self._h, _p(depth, C.c_float), H, W, _p(Kinv, C.c_double), _p(K, C.c_double),
                                       _p(Kf, C.c_double), _p(T, C.c_double), _p(idx, C.c_int32), len(idx),
                                       _p(feat, C.c_float), Hf, Wf, _p(rgb, C.c_uint8), min_depth, max_depth)
        if n < 0:
            raise IndexError("rgb index out of bounds (the reference would raise here too)")
        return n

    def export(self):
        n = lib().avlo_map_size(self._h)
        gf = np.zeros((n, self.D), dtype=np.float32)
        gp = np.zeros((n, 3), dtype=np.int32)
        w = np.zeros(n, dtype=np.float64)
        rgb = np.zeros((n, 3), dtype=np.float64)
        occ = np.zeros((self.gs, self.gs, self.vh), dtype=np.int32)
        lib().avlo_map_export(self._h, _p(gf, C.c_float), _p(gp, C.c_int32), _p(w, C.c_double), _p(rgb, C.c_double),
                              _p(occ, C.c_int32))
        grown = lib().avlo_map_grown(self._h)
        # dtypes the reference ends with (vlmap_builder.py:205-206 / :303-310)
        weight = w if grown else w.astype(np.float32)
        grid_rgb = rgb.astype(np.float32) if grown else rgb.astype(np.uint8)
        return dict(grid_feat=gf, grid_pos=gp, weight=weight, grid_rgb=grid_rgb, occupied_ids=occ, grown=grown)


HABITAT2CAM_ROT = np.diag([1.0, -1.0, -1.0, 1.0])     # vlmap_builder_multi_floor.py:77-79


def points_bbox(minmax, depth_m, calib, transform, sample_idx, min_depth=0.1, max_depth=100.0):
    """fold one frame into minmax (6,) float64 in place -- pass 1 of create_global_map (:97-118)"""
    d = np.ascontiguousarray(depth_m, dtype=np.float64)
    Kinv = np.ascontiguousarray(np.linalg.inv(np.asarray(calib, dtype=np.float64).reshape(3, 3)))
    T = np.ascontiguousarray(transform, dtype=np.float64)
    idx = np.ascontiguousarray(sample_idx, dtype=np.int32)
    lib().avlo_points_bbox(_p(d, C.c_double), d.shape[1], _p(Kinv, C.c_double), _p(T, C.c_double), _p(idx, C.c_int32), len(idx),
                           min_depth, max_depth, _p(minmax, C.c_double))
    return minmax


class OracleGlobalMap(OracleMap):
    """Sequential multi-floor builder (vlmap_builder_multi_floor.py:120-199) on a (n0, n1, n2) grid."""

    def __init__(self, pcd_min, pcd_max, cs, D):
        self.pcd_min = np.ascontiguousarray(pcd_min, dtype=np.float64)
        self.grid_size = np.ceil((np.asarray(pcd_max) - self.pcd_min) / cs + 1).astype(int)      # :222 (x, y, z)
        self.n0, self.gs, self.vh = int(self.grid_size[0]), int(self.grid_size[2]), int(self.grid_size[1])
        self.cs, self.D = float(cs), int(D)
        self._h = lib().avlo_map_create_grid(self.n0, self.gs, self.vh, self.cs, self.D)

    def integrate(self, depth_m, calib, transform, sample_idx, feat_chw, rgb, min_depth=0.1, max_depth=100.0):
        d = np.ascontiguousarray(depth_m, dtype=np.float64)
        H, W = d.shape
        K = np.ascontiguousarray(np.asarray(calib, dtype=np.float64).reshape(3, 3))
        Kinv = np.ascontiguousarray(np.linalg.inv(K))
        feat = np.ascontiguousarray(feat_chw, dtype=np.float32)
        if feat.ndim == 4:
            feat = feat[0]
        _, Hf, Wf = feat.shape
        Kf = np.ascontiguousarray(get_sim_cam_mat(Hf, Wf))
        T = np.ascontiguousarray(transform, dtype=np.float64)
        idx = np.ascontiguousarray(sample_idx, dtype=np.int32)
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        n = lib().avlo_integrate_frame_global(self._h, _p(d, C.c_double), H, W, _p(Kinv, C.c_double), _p(K, C.c_double),
                                              _p(Kf, C.c_double), _p(T, C.c_double), _p(idx, C.c_int32), len(idx),
                                              _p(feat, C.c_float), Hf, Wf, _p(rgb, C.c_uint8), min_depth, max_depth,
                                              _p(self.pcd_min, C.c_double))
        if n < 0:
            raise IndexError("rgb index out of bounds (the reference would raise here too)")
        return n

    def export(self):
        gs = self.gs
        self.gs = self.n0          # OracleMap.export allocates (gs, gs, vh); the global grid is (n0, n1, n2)
        try:
            n = lib().avlo_map_size(self._h)
            gf = np.zeros((n, self.D), dtype=np.float32)
            gp = np.zeros((n, 3), dtype=np.int32)
            w = np.zeros(n, dtype=np.float64)
            rgb = np.zeros((n, 3), dtype=np.float64)
            occ = np.zeros((self.n0, gs, self.vh), dtype=np.int32)
            lib().avlo_map_export(self._h, _p(gf, C.c_float), _p(gp, C.c_int32), _p(w, C.c_double), _p(rgb, C.c_double),
                                  _p(occ, C.c_int32))
        finally:
            self.gs = gs
        grown = lib().avlo_map_grown(self._h)
        return dict(grid_feat=gf, grid_pos=gp, weight=w if grown else w.astype(np.float32),
                    grid_rgb=rgb.astype(np.float32) if grown else rgb.astype(np.uint8), occupied_ids=occ, grown=grown)


def sample_indices(rng_state: np.random.RandomState, n_pix: int, rate: int):
    """vlmap_builder.py:275-277 with an explicit RandomState instead of the global one."""
    m = np.arange(n_pix)
    rng_state.shuffle(m)
    return m[::rate].astype(np.int32)


# ------------------------------------------------------------------ index path
def template_mean(template_feats):
    """clip_utils.py:223-225: (Q*T, D) -> reshape (Q, T, D) -> float32 mean over templates (not re-normalised)."""
    return np.mean(np.asarray(template_feats, dtype=np.float32), axis=1)


def sim_scores(map_feats, text_feats):
    """clip_utils.py:227-229: raw dot product, float32 BLAS sgemm."""
    map_feats = np.asarray(map_feats, dtype=np.float32).reshape((-1, map_feats.shape[-1]))
    return map_feats @ np.asarray(text_feats, dtype=np.float32).T


def argmax_mask(scores, cat_id=0):
    """vlmap.py:123-124 (np.argmax: first maximum wins)."""
    ids = np.argmax(scores, axis=1)
    return ids == cat_id, ids


def sim_scores_scalar(map_feats, text_feats, want_scores=True):
    """single-core scalar port (float64 accumulate) -- order-independent cross-check / cpu_baseline 'port'."""
    a = np.ascontiguousarray(map_feats, dtype=np.float32)
    q = np.ascontiguousarray(text_feats, dtype=np.float32)
    N, D = a.shape
    Q = q.shape[0]
    sc = np.zeros((N, Q), dtype=np.float32) if want_scores else None
    am = np.zeros(N, dtype=np.int32)
    lib().avlo_sim_scores(_p(a, C.c_float), N, D, _p(q, C.c_float), Q,
                          _p(sc, C.c_float) if want_scores else None, _p(am, C.c_int32))
    return sc, am


def heatmap_from_mask(grid_pos, mask, cell_size=0.05, decay_rate=0.01):
    """visualize_utils.py:29-49."""
    pos = np.ascontiguousarray(grid_pos, dtype=np.int32)
    mk = np.ascontiguousarray(mask, dtype=np.uint8)
    heat = np.zeros(len(pos), dtype=np.float32)
    lib().avlo_heatmap_from_mask(_p(pos, C.c_int32), _p(mk, C.c_uint8), len(pos), cell_size, decay_rate,
                                 _p(heat, C.c_float))
    return heat


# ---------------------------------------------------------------------------------------------------------------------
# top-down 2-D products (restated sequentially / with NumPy; pinned against tests/golden/g8_map2d.npz)
def pool_3d_label_to_2d(mask_3d, grid_pos, gs):
    """avlmaps/utils/visualize_utils.py:77-83: mask_2d[row, col] = mask_3d[i] or mask_2d[row, col] over all voxels"""
    mask_2d = np.zeros((gs, gs), dtype=bool)
    for i in np.flatnonzero(np.asarray(mask_3d, dtype=bool)):
        mask_2d[grid_pos[i, 0], grid_pos[i, 1]] = True
    return mask_2d


def obstacle_map(occupied_ids, cs, h_min=0, h_max=1.5):
    """avlmaps/map/map.py:79-95: True = free; a column is an obstacle if a voxel id > 0 lies strictly inside the height band"""
    heights = np.arange(0, occupied_ids.shape[-1]) * cs
    height_mask = np.logical_and(heights > h_min, heights < h_max)
    return np.sum(occupied_ids[..., height_mask] > 0, axis=2) == 0


def crop_bounds(obstacle_map_):
    """avlmaps/map/map.py:97-104: (rmin, rmax, cmin, cmax) of the obstacle cells"""
    x, y = np.where(obstacle_map_ == 0)
    return int(x.min()), int(x.max()), int(y.min()), int(y.max())


def rgb_topdown(grid_pos, grid_rgb, gs):
    """avlmaps/map/map.py:106-113: sequential loop, the last voxel written to a (row, col) wins"""
    out = np.zeros((gs, gs, 3))
    for rgb, pos in zip(grid_rgb, grid_pos):
        out[pos[0], pos[1], :] = np.asarray(rgb).flatten()
    return out.astype(np.uint8)


def dynamic_obstacles(predict, potential, obstacle_names, grid_pos, rmin, cmin, obstacles_cropped):
    """avlmaps/utils/index_utils.py:162-177 (after the argmax): True = free"""
    obs_inds = [i for name in obstacle_names for i, po in enumerate(potential) if name == po]
    pts_mask = np.zeros_like(predict, dtype=bool)
    for k in obs_inds:
        pts_mask |= predict == k
    new_obstacles = np.zeros_like(obstacles_cropped, dtype=bool)
    pts = grid_pos[pts_mask]
    new_obstacles[pts[:, 0] - rmin, pts[:, 1] - cmin] = 1
    return np.logical_not(np.logical_and(new_obstacles, obstacles_cropped == 0))
