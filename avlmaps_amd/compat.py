"""Switch an importable upstream `avlmaps` package onto the HIP path without editing it.

    import avlmaps_amd.compat as compat
    compat.install()            # patches avlmaps.* in place; compat.uninstall() restores the originals

What is replaced (every module of the upstream package that holds a reference to one of these objects is re-pointed,
so `from avlmaps.utils.clip_utils import get_lseg_score` in avlmaps/map/vlmap.py:24 is covered too):

  avlmaps.utils.clip_utils.get_lseg_score          (clip_utils.py:196-242)   -> voxel x query similarity kernels
  avlmaps.utils.index_utils.get_lseg_score         (index_utils.py:64-108, verbatim duplicate)
  avlmaps.utils.index_utils.get_dynamic_obstacles_map_3d (index_utils.py:138-184)
  avlmaps.utils.visualize_utils.get_heatmap_from_mask_3d (visualize_utils.py:29-49) -> heat kernels
  avlmaps.utils.visualize_utils.pool_3d_label_to_2d      (visualize_utils.py:77-83)
  avlmaps.map.vlmap_builder.VLMapBuilder.create_mobile_base_map (vlmap_builder.py:54-185) -> builder kernels
  avlmaps.map.map.Map.generate_obstacle_map / generate_rgb_topdown_map (map.py:79-95, :106-113) -> top-down scatter kernels
  avlmaps.robot.habitat_lang_robot.HabitatLanguageRobot.get_vl_distribution_map_3d (habitat_lang_robot.py:242-265) -> heat kernels
      (only if that module is already imported -- it needs habitat_sim -- and only for a robot whose map holds grid_pos: the
      navigator's own decay loop never calls get_heatmap_from_mask_3d; upstream it indexes a Python list -- global_pc comes
      from grid_id2base_pos_3d_batch, mapping_utils.py:369-376 -- so the adapter is for the day that is fixed)

The upstream objects keep their classes, attributes and files (vlmaps.h5df): VLMap / AVLMap / the Habitat navigator run
unchanged on top.  CLIP text encoding and LSeg stay the upstream PyTorch models.
"""
from __future__ import annotations

import importlib
import sys
from typing import Dict, List, Tuple

_INSTALLED: Dict[str, List[Tuple[object, str, object]]] = {}


def _repoint(upstream: str, old, new, log):
    """replace every module-level reference to `old` inside the upstream package"""
    for name, mod in list(sys.modules.items()):
        if mod is None or not (name == upstream or name.startswith(upstream + ".")):
            continue
        try:
            items = list(vars(mod).items())
        except TypeError:
            continue
        for attr, val in items:
            if val is old:
                setattr(mod, attr, new)
                log.append((mod, attr, old))


def _builder_adapter():
    def create_mobile_base_map(self):
        """avlmaps.map.vlmap_builder.VLMapBuilder.create_mobile_base_map on the HIP builder (same inputs, same output file)"""
        from .map.vlmap_builder import VLMapBuilder as HipBuilder
        hb = HipBuilder(self.data_dir, self.map_config, self.pose_path, self.rgb_paths, self.depth_paths, self.base2cam_tf,
                        self.base_transform, feat_extractor=getattr(self, "feat_extractor", None))
        hb.create_mobile_base_map()
        self.map_save_dir, self.map_save_path = hb.map_save_dir, hb.map_save_path
        return None
    return create_mobile_base_map


def _map_adapters():
    def generate_obstacle_map(self, h_min: float = 0, h_max: float = 1.5):
        """avlmaps.map.map.Map.generate_obstacle_map on avl_obstacle_map (same attributes set)"""
        from . import ops
        assert self.occupied_ids is not None, "map not loaded"
        self.obstacles_map = ops.obstacle_map(self.occupied_ids, self.cs, h_min, h_max)
        self.generate_cropped_obstacle_map(self.obstacles_map)
        return self.obstacles_map

    def generate_rgb_topdown_map(self):
        """avlmaps.map.map.Map.generate_rgb_topdown_map on avl_rgb_topdown (the last voxel of a column wins)"""
        from . import ops
        assert self.grid_rgb is not None, "map not loaded"
        assert self.grid_pos is not None
        return ops.rgb_topdown(self.grid_pos, self.grid_rgb, self.gs)
    return dict(generate_obstacle_map=generate_obstacle_map, generate_rgb_topdown_map=generate_rgb_topdown_map)


def _navigator_adapter(module):
    def get_vl_distribution_map_3d(self, name: str, decay_rate: float = 0.1):
        """avlmaps.robot.habitat_lang_robot.HabitatLanguageRobot.get_vl_distribution_map_3d on the heat kernels: per voxel
        clip(1 - (distance to the nearest voxel of category `name` in CELLS) * decay_rate, 0, 1), 1 on the category's own voxels
        (upstream divides metre distances of global_pc = grid_pos * cs by cs: distances in cells, habitat_lang_robot.py:252-256)"""
        import numpy as np
        from . import ops
        predict = np.argmax(self.map.scores_mat, axis=1)
        i = module.find_similar_category_id(name, self.map.categories)
        sim = predict == i
        pos = np.ascontiguousarray(self.map.grid_pos, dtype=np.int32)
        heat = ops.heatmap_from_mask(pos, sim, 1.0, decay_rate, reuse_plan=pos is self.map.grid_pos).astype(np.float32)
        cfg = getattr(self, "config", None)
        try:
            if cfg is not None and cfg["nav"]["vis"]:
                self._vis_dist_map_3d(heat[:, None], name=name)
        except (KeyError, TypeError):
            pass
        return heat
    return get_vl_distribution_map_3d


def install(upstream: str = "avlmaps") -> Dict[str, int]:
    """Patch the upstream package (must be importable).  Returns {patched name: number of references re-pointed}."""
    if upstream in _INSTALLED:
        return {}
    from .utils import clip_utils as my_cu
    from .utils import index_utils as my_iu
    from .utils import visualize_utils as my_vu
    log: List[Tuple[object, str, object]] = []
    counts: Dict[str, int] = {}

    def swap(modname, attr, new):
        try:
            mod = importlib.import_module(f"{upstream}.{modname}")
        except Exception:
            return
        old = getattr(mod, attr, None)
        if old is None or old is new:
            return
        n0 = len(log)
        _repoint(upstream, old, new, log)
        if getattr(mod, attr) is not new:          # not a plain module attribute any more (e.g. wrapped): set it directly
            setattr(mod, attr, new)
            log.append((mod, attr, old))
        counts[f"{modname}.{attr}"] = len(log) - n0

    swap("utils.clip_utils", "get_lseg_score", my_cu.get_lseg_score)
    swap("utils.index_utils", "get_lseg_score", my_cu.get_lseg_score)
    swap("utils.index_utils", "get_dynamic_obstacles_map_3d", my_iu.get_dynamic_obstacles_map_3d)
    swap("utils.visualize_utils", "get_heatmap_from_mask_3d", my_vu.get_heatmap_from_mask_3d)
    swap("utils.visualize_utils", "pool_3d_label_to_2d", my_vu.pool_3d_label_to_2d)
    try:
        vb = importlib.import_module(f"{upstream}.map.vlmap_builder")
        cls = vb.VLMapBuilder
        old = cls.__dict__.get("create_mobile_base_map")
        if old is not None:
            setattr(cls, "create_mobile_base_map", _builder_adapter())
            log.append((cls, "create_mobile_base_map", old))
            counts["map.vlmap_builder.VLMapBuilder.create_mobile_base_map"] = 1
    except Exception:
        pass
    try:
        mp = importlib.import_module(f"{upstream}.map.map")
        for name, fn in _map_adapters().items():
            old = mp.Map.__dict__.get(name)
            if old is not None:
                setattr(mp.Map, name, fn)
                log.append((mp.Map, name, old))
                counts[f"map.map.Map.{name}"] = 1
    except Exception:
        pass
    nav = sys.modules.get(f"{upstream}.robot.habitat_lang_robot")       # never imported from here: it pulls in habitat_sim
    cls = getattr(nav, "HabitatLanguageRobot", None) if nav is not None else None
    if cls is not None and "get_vl_distribution_map_3d" in cls.__dict__ and hasattr(nav, "find_similar_category_id"):
        old = cls.__dict__["get_vl_distribution_map_3d"]
        setattr(cls, "get_vl_distribution_map_3d", _navigator_adapter(nav))
        log.append((cls, "get_vl_distribution_map_3d", old))
        counts["robot.habitat_lang_robot.HabitatLanguageRobot.get_vl_distribution_map_3d"] = 1
    _INSTALLED[upstream] = log
    return counts


def uninstall(upstream: str = "avlmaps") -> None:
    for owner, attr, old in reversed(_INSTALLED.pop(upstream, [])):
        setattr(owner, attr, old)
