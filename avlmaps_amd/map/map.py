"""Map base class with the reference's attribute surface (avlmaps/map/map.py:18-129).

Consumers upstream (AVLMap, the Habitat dataloader and robot) read these attributes directly, so they keep
the reference's names, dtypes and host-NumPy type: grid_feat, grid_pos, weight, occupied_ids, grid_rgb,
mapped_iter_list, gs, cs, base2cam_tf, base_transform, obstacles_map, obstacles_cropped, rmin/rmax/cmin/cmax.
"""
from __future__ import annotations

from pathlib import Path
from typing import List, Union

import numpy as np


def cfg_get(cfg, key):
    """attribute-or-item access: works for omegaconf.DictConfig, dicts and plain namespaces"""
    if isinstance(cfg, dict):
        return cfg[key]
    try:
        return getattr(cfg, key)
    except AttributeError:
        return cfg[key]


class Map:
    def __init__(self, map_config, data_dir: str = ""):
        self.map_config = map_config
        self.gs = cfg_get(map_config, "grid_size")
        self.cs = cfg_get(map_config, "cell_size")
        self.mapped_iter_list = None
        self.grid_feat = None
        self.grid_pos = None
        self.weight = None
        self.occupied_ids = None
        self.grid_rgb = None
        self.obstacles_map = None
        self.obstacles_cropped = None
        self._setup_transforms()
        if data_dir:
            self._setup_paths(data_dir)

    def _setup_paths(self, data_dir: Union[Path, str]) -> None:
        """Dataset layout of dataset/README.md:76-93.  Reference: map.py:41-52."""
        self.data_dir = Path(data_dir)
        self.rgb_dir = self.data_dir / "rgb"
        self.depth_dir = self.data_dir / "depth"
        self.semantic_dir = self.data_dir / "semantic"
        self.pose_path = self.data_dir / "poses.txt"
        self.rgb_paths = sorted(self.rgb_dir.glob("*.png"))
        self.depth_paths = sorted(self.depth_dir.glob("*.npy"))
        self.semantic_paths = sorted(self.semantic_dir.glob("*.npy"))

    def _setup_transforms(self):
        """base->camera transform and the (forward, left, up) re-basing matrix.  Reference: map.py:54-68."""
        pose_info = cfg_get(self.map_config, "pose_info")
        self.base2cam_tf = np.eye(4)
        self.base2cam_tf[:3, :3] = np.array([list(cfg_get(pose_info, "base2cam_rot"))]).reshape((3, 3))
        self.base2cam_tf[1, 3] = cfg_get(pose_info, "camera_height")
        self.base_transform = np.eye(4)
        self.base_transform[0, :3] = list(cfg_get(pose_info, "base_forward_axis"))
        self.base_transform[1, :3] = list(cfg_get(pose_info, "base_left_axis"))
        self.base_transform[2, :3] = list(cfg_get(pose_info, "base_up_axis"))
        return self.base2cam_tf, self.base_transform

    def create_map(self, data_dir):
        return NotImplementedError

    def load_map(self, map_dir: str):
        return NotImplementedError

    def index_map(self, language_desc: str, with_init_cat: bool = True):
        return NotImplementedError

    def generate_obstacle_map(self, h_min: float = 0, h_max: float = 1.5) -> np.ndarray:
        """(gs, gs) bool, True = free.  Reference: map.py:79-95 (note `> 0`: voxel id 0 does not count upstream)."""
        assert self.occupied_ids is not None, "map not loaded"
        from .. import ops
        self.obstacles_map = ops.obstacle_map(self.occupied_ids, self.cs, h_min, h_max)      # avl_obstacle_map
        self.generate_cropped_obstacle_map(self.obstacles_map)
        return self.obstacles_map

    def generate_cropped_obstacle_map(self, obstacle_map: np.ndarray) -> np.ndarray:
        """Reference: map.py:97-104."""
        x_indices, y_indices = np.where(obstacle_map == 0)
        self.rmin, self.rmax = np.min(x_indices), np.max(x_indices)
        self.cmin, self.cmax = np.min(y_indices), np.max(y_indices)
        self.obstacles_cropped = obstacle_map[self.rmin:self.rmax + 1, self.cmin:self.cmax + 1]
        return self.obstacles_cropped

    def generate_rgb_topdown_map(self) -> np.ndarray:
        """Last voxel written per (row, col) wins, like the reference's sequential loop (map.py:106-113)."""
        assert self.grid_rgb is not None, "map not loaded"
        assert self.grid_pos is not None
        from .. import ops
        return ops.rgb_topdown(self.grid_pos, self.grid_rgb, self.gs)                         # avl_rgb_topdown

    def init_categories(self, categories: List[str]) -> np.ndarray:
        return NotImplementedError

    @staticmethod
    def _dilate_map(binary_map: np.ndarray, dilate_iter: int = 0, gaussian_sigma: float = 1.0):
        """2x bilinear upsample -> gaussian -> threshold -> dilation -> downsample.  Reference: map.py:169-181.
        Navigator-side image processing (not on the accelerated path); needs OpenCV exactly like upstream."""
        import cv2
        from scipy.ndimage import binary_dilation, gaussian_filter
        h, w = binary_map.shape
        m = cv2.resize(binary_map.astype(float), (w * 2, h * 2))
        m = gaussian_filter(m.astype(float), sigma=gaussian_sigma, truncate=3)
        m = (m > 0.5).astype(np.uint8)
        m = binary_dilation(m, structure=np.ones((3, 3)), iterations=dilate_iter * 2)
        return cv2.resize(m.astype(float), (w, h))

    @staticmethod
    def create(map_config) -> "Map":
        """Reference: map.py:120-129."""
        from .vlmap import VLMap
        if cfg_get(map_config, "map_type") == "vlmap":
            return VLMap(map_config)
        if cfg_get(map_config, "map_type") == "vlmap_openmap":
            from .vlmap_multi_floor import VLMapMultiFloor
            return VLMapMultiFloor(map_config)
        raise NotImplementedError(f"map_type {cfg_get(map_config, 'map_type')!r}: only 'vlmap' and 'vlmap_openmap' are on the accelerated path")
