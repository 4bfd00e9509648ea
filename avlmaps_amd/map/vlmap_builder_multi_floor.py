"""VLMapBuilderMultiFloor with the reference's interface (avlmaps/map/vlmap_builder_multi_floor.py:34-393): a map in the
GLOBAL frame from per-frame 4x4 camera poses, built in two passes on the MI355X.

  pass 1 (:97-118)  bounding box of all transformed sampled points          -> avl_points_bbox per frame
  pass 2 (:124-199) voxel index np.round((p - pcd_min) / cs), feature fusion -> avl_builder_integrate_frame_global

Both passes draw their own pixel sample from the global NumPy RNG, in the reference's order (all of pass 1, then all of
pass 2), so a seeded run samples the same pixels.  Deviation, on purpose: a pass-2 point outside the pass-1 box makes the
reference wrap a negative index or raise IndexError; here it is dropped.  The reference's resume path is broken upstream
(_init_map unpacks 8 of the 9 values load_3d_map returns) and is not mirrored.
"""
from __future__ import annotations

import os
from pathlib import Path
from typing import Callable, List, Optional

import numpy as np

from .. import ops
from ..utils.mapping_utils import load_rgb_png, read_map_datasets, write_map_datasets
from .map import cfg_get
from .vlmap_builder import VLMapBuilder

HABITAT2CAM_ROT = np.diag([1.0, -1.0, -1.0, 1.0])        # vlmap_builder_multi_floor.py:77-79


def load_depth_img(depth_filepath) -> np.ndarray:
    """uint16 millimetre depth PNG, unchanged (mapping_utils.py:93-94: cv2.imread(..., IMREAD_UNCHANGED))."""
    from PIL import Image
    with Image.open(depth_filepath) as im:
        return np.asarray(im).astype(np.uint16)


class VLMapBuilderMultiFloor:
    def __init__(self, data_dir: Path, map_config, pose_paths: List[Path], rgb_paths: List[Path], depth_paths: List[Path],
                 base2cam_tf: np.ndarray, base_transform: np.ndarray, feat_extractor: Optional[Callable] = None):
        self.data_dir = Path(data_dir)
        self.pose_paths = pose_paths
        self.rgb_paths = rgb_paths
        self.depth_paths = depth_paths
        self.map_config = map_config
        self.base2cam_tf = base2cam_tf
        self.base_transform = base_transform
        self.feat_extractor = feat_extractor
        self.capacity = None
        self.exact_rgb = True
        self.min_depth, self.max_depth, self.sigma_sq = 0.1, 100, 0.6     # :106, :160

    # frame sources (overridable for in-memory data)
    def load_frame(self, frame_i: int):
        return load_rgb_png(self.rgb_paths[frame_i]), load_depth_img(self.depth_paths[frame_i])

    def load_pose(self, frame_i: int) -> np.ndarray:
        return np.loadtxt(self.pose_paths[frame_i]).reshape((4, 4))

    _features_hwc = VLMapBuilder._features_hwc
    _init_lseg = VLMapBuilder._init_lseg
    sample_pixels = staticmethod(VLMapBuilder.sample_pixels)

    def create_global_map(self):
        """Reference: vlmap_builder_multi_floor.py:60-199."""
        cs = cfg_get(self.map_config, "cell_size")
        depth_sample_rate = cfg_get(self.map_config, "depth_sample_rate")
        skip_frame = cfg_get(self.map_config, "skip_frame")
        calib_mat = np.array(list(cfg_get(self.map_config, "cam_calib_mat")), dtype=np.float64).reshape((3, 3))
        calib_inv = np.linalg.inv(calib_mat)
        self.camera_pose_tfs = [self.load_pose(i) for i in range(len(self.pose_paths))]
        self.init_cam_tf = self.camera_pose_tfs[0]
        self.inv_init_cam_tf = np.linalg.inv(self.init_cam_tf)
        self.habitat2cam_rot_tf = HABITAT2CAM_ROT.copy()
        self.map_save_dir = self.data_dir / "vlmap_multi_floor"
        os.makedirs(self.map_save_dir, exist_ok=True)
        self.map_save_path = self.map_save_dir / "vlmaps_multi_floor.h5df"
        self._init_lseg()
        frames = [i for i in range(min(len(self.rgb_paths), len(self.depth_paths), len(self.camera_pose_tfs)))
                  if i % skip_frame == 0]

        # pass 1: global bounding box
        minmax = np.array([np.inf] * 3 + [-np.inf] * 3)
        for frame_i in frames:
            _, depth = self.load_frame(frame_i)
            samples = self.sample_pixels(depth.shape[0] * depth.shape[1], depth_sample_rate)
            ops.points_bbox(minmax, depth, calib_mat, self.camera_pose_tfs[frame_i] @ self.habitat2cam_rot_tf, samples,
                            depth_div=1000.0, min_depth=self.min_depth, max_depth=self.max_depth)
        self.pcd_min, self.pcd_max = minmax[:3].copy(), minmax[3:].copy()
        self.grid_size = np.ceil((self.pcd_max - self.pcd_min) / cs + 1).astype(int)       # (x, y, z), :222

        # pass 2: fusion
        n0, n1, n2 = int(self.grid_size[0]), int(self.grid_size[2]), int(self.grid_size[1])
        acc, mapped_iter_set = None, set()
        for frame_i in frames:
            rgb, depth = self.load_frame(frame_i)
            feat = self._features_hwc(rgb)
            if acc is None:
                self.clip_feat_dim = int(feat.shape[2])
                cap = self.capacity or max(n0 * n1, 1 << 16)
                acc = ops.VoxelAccumulator(n1, cs, n2, self.clip_feat_dim, capacity=cap, n_rows=n0)
                if self.exact_rgb:
                    npix = depth.shape[0] * depth.shape[1]
                    acc.enable_replay_log(len(frames) * ((npix + depth_sample_rate - 1) // depth_sample_rate))
            samples = self.sample_pixels(depth.shape[0] * depth.shape[1], depth_sample_rate)
            acc.integrate_frame_global(depth, calib_mat, self.camera_pose_tfs[frame_i] @ self.habitat2cam_rot_tf, samples, feat,
                                       rgb, frame_idx=frame_i, pcd_min=self.pcd_min, depth_div=1000.0, calib_inv=calib_inv,
                                       min_depth=self.min_depth, max_depth=self.max_depth, sigma_sq=self.sigma_sq)
            mapped_iter_set.add(frame_i)
        if acc is None:
            raise RuntimeError("no frames to map")
        self.save_3d_map(acc.finalize(), mapped_iter_set)

    def create_mobile_base_map(self):
        return NotImplementedError        # :201-207 upstream

    def create_camera_map(self):
        return NotImplementedError        # :209-215 upstream

    def save_3d_map(self, arrays, mapped_iter_set) -> None:
        """Reference: vlmap_builder_multi_floor.py:368-393 (the six datasets + pcd_min, pcd_max, cs)."""
        self.last_map = arrays
        data = dict(mapped_iter_list=np.array(sorted(mapped_iter_set), dtype=np.int32), grid_feat=arrays["grid_feat"],
                    grid_pos=arrays["grid_pos"], weight=arrays["weight"], occupied_ids=arrays["occupied_ids"],
                    grid_rgb=arrays["grid_rgb"], pcd_min=self.pcd_min, pcd_max=self.pcd_max,
                    cs=np.array(cfg_get(self.map_config, "cell_size")))
        write_map_datasets(self.map_save_path, {k: np.asarray(v) for k, v in data.items()})

    @staticmethod
    def load_3d_map(map_path):
        """-> (mapped_iter_list, grid_feat, grid_pos, weight, occupied_ids, grid_rgb, pcd_min, pcd_max, cs). Reference: :244-256."""
        d = read_map_datasets(Path(map_path))
        return (d["mapped_iter_list"].tolist(), d["grid_feat"], d["grid_pos"], d["weight"], d["occupied_ids"], d["grid_rgb"],
                d["pcd_min"], d["pcd_max"], float(d["cs"]))
