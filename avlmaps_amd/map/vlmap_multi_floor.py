"""VLMapMultiFloor with the reference's interface (avlmaps/map/vlmap_multi_floor.py:27-206): global-frame map built by
VLMapBuilderMultiFloor; indexing is inherited from VLMap (same similarity + argmax kernels)."""
from __future__ import annotations

from pathlib import Path
from typing import Union

from .vlmap import VLMap
from .vlmap_builder_multi_floor import VLMapBuilderMultiFloor


class VLMapMultiFloor(VLMap):
    def _setup_paths(self, data_dir: Union[Path, str]) -> None:
        """rgb/*.png, depth/*.png (uint16 mm), pose/*.txt (4x4 per frame).  Reference: vlmap_multi_floor.py:33-45."""
        self.data_dir = Path(data_dir)
        self.rgb_dir = self.data_dir / "rgb"
        self.depth_dir = self.data_dir / "depth"
        self.pose_dir = self.data_dir / "pose"
        self.rgb_paths = sorted(self.rgb_dir.glob("*.png"))
        self.depth_paths = sorted(self.depth_dir.glob("*.png"))
        self.pose_paths = sorted(self.pose_dir.glob("*.txt"))

    def create_map(self, data_dir: Union[Path, str], feat_extractor=None) -> None:
        """Reference: vlmap_multi_floor.py:47-64."""
        print("Creating map for scene at: ", data_dir)
        self._setup_paths(data_dir)
        self.map_builder = VLMapBuilderMultiFloor(self.data_dir, self.map_config, self.pose_paths, self.rgb_paths,
                                                  self.depth_paths, self.base2cam_tf, self.base_transform,
                                                  feat_extractor=feat_extractor)
        self.map_builder.create_global_map()

    def load_map(self, data_dir: str) -> bool:
        """Reference: vlmap_multi_floor.py:66-84."""
        self._setup_paths(data_dir)
        self.map_save_path = Path(data_dir) / "vlmap_multi_floor" / "vlmaps_multi_floor.h5df"
        if not (self.map_save_path.exists() or self.map_save_path.with_name(self.map_save_path.name + ".npz").exists()):
            print("Loading VLMap failed because the file doesn't exist.")
            return False
        (self.mapped_iter_list, self.grid_feat, self.grid_pos, self.weight, self.occupied_ids, self.grid_rgb, self.pcd_min,
         self.pcd_max, self.cs) = VLMapBuilderMultiFloor.load_3d_map(self.map_save_path)
        self._dev_feat = None
        return True
