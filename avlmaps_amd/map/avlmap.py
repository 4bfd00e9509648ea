"""AVLMap facade restricted to the accelerated path (avlmaps/map/avlmap.py:18-76): the VLMap sub-map,
create_map / load_map / index_object.  Sound, area and image indexing are separate upstream subsystems
(AudioCLIP, CLIP ViT-L/14 sparse map, HLoc) that do not touch the voxel hot path."""
from __future__ import annotations

from typing import List

import numpy as np

from ..utils.visualize_utils import get_heatmap_from_mask_3d
from .map import cfg_get
from .vlmap import VLMap


class AVLMap:
    def __init__(self, config, data_dir: str = ""):
        self.config = config
        self.vlmap = VLMap(cfg_get(config, "map_config"), data_dir=data_dir)

    def create_map(self, data_dir, feat_extractor=None) -> bool:
        self.vlmap.create_map(data_dir, feat_extractor=feat_extractor)
        return True

    def load_map(self, data_dir: str) -> bool:
        return self.vlmap.load_map(data_dir)

    def index_object(self, object_name: str, init_categories: List[str] = None, decay_rate: float = 0.1) -> np.ndarray:
        """(N,) float32 heat.  Reference: avlmap.py:67-76."""
        cs = cfg_get(cfg_get(self.config, "params"), "cs")
        if init_categories is not None:
            self.vlmap.init_categories(init_categories[1:-1])
            mask = self.vlmap.index_map(object_name, with_init_cat=True)
            return get_heatmap_from_mask_3d(self.vlmap.grid_pos, mask, cell_size=cs, decay_rate=decay_rate)
        # query -> argmax -> mask -> heat without leaving the GPU: only the (N,) float32 heat comes back
        from .. import ops
        vm = self.vlmap
        q = vm._text_feats([object_name])            # cached per string: the text tower costs more than the kernels below
        feat = vm._device_feat()
        if vm._rows != (0, len(vm.grid_feat)):          # voxel rows sharded over ranks: gather the argmax, heat on the full map
            mask = vm._score(q, want_scores=False)[1] == 0
            if not mask.any():
                raise ValueError("attempt to get argmin of an empty sequence")
            return vm.heatmap_from_mask(mask.astype(np.uint8), cs, decay_rate).numpy()
        _, am, _ = ops.sim_scores(feat, q, want_scores=False, want_argmax=True, precision=vm._sim_precision)
        mask = ops.mask_from_argmax(am, 0)
        heat = vm.heatmap_from_mask(mask, cs, decay_rate)
        if vm.grid_pos.shape[0] and ops.argmax_f32(heat)[1] < 1.0:      # a target voxel has heat exactly 1
            raise ValueError("attempt to get argmin of an empty sequence")   # what np.argmin raises upstream (no voxel matched)
        return heat.numpy()

    def index_sound(self, *a, **k):
        raise NotImplementedError("sound indexing (AudioCLIP segment map) is outside the accelerated voxel path")

    def index_area(self, *a, **k):
        raise NotImplementedError("area indexing (sparse CLIP ViT-L/14 frame map) is outside the accelerated voxel path")

    def index_image(self, *a, **k):
        raise NotImplementedError("image localisation (HLoc) is outside the accelerated voxel path")
