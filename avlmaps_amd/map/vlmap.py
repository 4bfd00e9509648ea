"""VLMap with the reference's interface (avlmaps/map/vlmap.py:27-187), indexing on the MI355X.

grid_feat stays a host NumPy attribute (the navigator reads it) and is mirrored ONCE into HBM; every
index_map / init_categories call then streams the device copy through the similarity kernel."""
from __future__ import annotations

from pathlib import Path
from typing import List, Union

import numpy as np

from ..utils.clip_utils import get_lseg_score, landmark_text_feats
from ..utils.mapping_utils import load_3d_map, map_file_exists
from .map import Map, cfg_get

CLIP_FEAT_DIM = {"RN50": 1024, "RN101": 512, "RN50x4": 640, "RN50x16": 768, "RN50x64": 1024, "ViT-B/32": 512,
                 "ViT-B/16": 512, "ViT-L/14": 768}


_UPLOADS = []       # map-upload threads still running: joined at interpreter exit so that the process never tears HIP down mid-copy


def _join_uploads():
    for th in list(_UPLOADS):
        th.join(timeout=30)


import atexit  # noqa: E402
atexit.register(_join_uploads)


class VLMap(Map):
    def __init__(self, map_config, data_dir: str = ""):
        super().__init__(map_config, data_dir=data_dir)
        self.scores_mat = None
        self.categories = None
        self._dev_feat = None
        self._dev_feat_src = None
        import threading
        self._dev_lock = threading.RLock()
        self.prefetch_device = True       # load_map starts the one-off upload + conversion of grid_feat (1.1 s at 2 M voxels: 4 GB over
                                          # PCIe from pageable memory) on a host thread, so that it overlaps with whatever the
                                          # caller does next (upstream: loading CLIP, seconds) instead of sitting in the first query
        self.compact_map = True           # the resident copy is the 3-byte form (ops.prepare_map(compact=True): fp16 hi + one
                                          # byte of residual in units of ulp(hi)/256, per-row scale): a query pass reads a quarter less
                                          # HBM (0.70 -> 0.61 ms at 2 M voxels x 64 queries), the copy is a quarter smaller, and the
                                          # scores stay float32-class (max error 2.3e-6 against 1.4e-6 for the 4-byte form; the path's
                                          # contract is 1e-4).  False: the 4-byte split-fp16 form
        self.shard_index_rows = True      # with torch.distributed initialised (one process per GPU) every rank keeps and scores
                                          # only its block of voxel rows; the per-voxel results are all-gathered (parallel.gather_rows)

    # ------------------------------------------------------------------ build / load
    def create_map(self, data_dir: Union[Path, str], feat_extractor=None) -> None:
        """Reference: vlmap.py:33-48."""
        from .vlmap_builder import VLMapBuilder
        print("Creating map for scene at: ", data_dir)
        self._setup_paths(data_dir)
        self.map_builder = VLMapBuilder(self.data_dir, self.map_config, self.pose_path, self.rgb_paths, self.depth_paths,
                                        self.base2cam_tf, self.base_transform, feat_extractor=feat_extractor)
        pose_type = cfg_get(cfg_get(self.map_config, "pose_info"), "pose_type")
        if pose_type == "mobile_base":
            self.map_builder.create_mobile_base_map()
        elif pose_type == "camera":
            self.map_builder.create_camera_map()

    def load_map(self, data_dir: str) -> bool:
        """Reference: vlmap.py:50-65 (prints and returns False when the file is missing)."""
        self._setup_paths(data_dir)
        self.map_save_path = Path(data_dir) / "vlmap" / "vlmaps.h5df"
        if not map_file_exists(self.map_save_path):
            print("Loading VLMap failed because the file doesn't exist.")
            return False
        (self.mapped_iter_list, self.grid_feat, self.grid_pos, self.weight, self.occupied_ids,
         self.grid_rgb) = load_3d_map(self.map_save_path)[:6]
        self._dev_feat = None
        # straight after a multi-GPU build in this process the merged map already lies row-sharded in the ranks' HBM
        # (VLMapBuilder.map_shard): take this rank's block as the resident copy instead of uploading it again from the file
        shard = getattr(getattr(self, "map_builder", None), "map_shard", None)
        if shard is not None and self.adopt_device_shard(shard):
            self.map_builder.map_shard = None            # the compact copy replaces the float32 block
            return True
        self._start_device_prefetch()
        return True

    def _start_device_prefetch(self) -> None:
        if not self.prefetch_device or self.grid_feat is None or len(self.grid_feat) == 0:
            return
        import threading
        from .. import _lib
        # the device the CALLING thread works on (HIP keeps the current device per thread): what it selected through
        # _lib.set_device or torch; asking HIP itself would start the runtime right here (~0.8 s), so if nothing is known yet the
        # upload goes to device 0 and _device_feat() checks at query time that the resident copy lives on the querying
        # thread's device (re-uploading once if the caller picked another GPU in between)
        dev = _lib.current_device(query_runtime=False)
        dev = 0 if dev is None else dev

        def work():
            try:
                _lib.load()
                _lib.require_gpu()                                         # first HIP call of the process: runtime start-up
                _lib.set_device(dev)
                self._device_feat()
            except Exception:             # no GPU / no library / upload failed: the query path repeats it on the calling thread
                pass                      # and reports the error there
        th = threading.Thread(target=work, name="avl-map-upload", daemon=True)
        th.start()
        _UPLOADS.append(th)
        del _UPLOADS[:-8]

    def _init_clip(self, clip_version="ViT-B/32"):
        """Reference: vlmap.py:67-90."""
        if hasattr(self, "clip_model"):
            print("clip model is already initialized")
            return
        import torch
        import clip
        self.device = "cuda" if torch.cuda.is_available() else "cpu"
        self.clip_version = clip_version
        self.clip_feat_dim = CLIP_FEAT_DIM[self.clip_version]
        print("Loading CLIP model...")
        self.clip_model, self.preprocess = clip.load(self.clip_version)
        self.clip_model.to(self.device).eval()

    # ------------------------------------------------------------------ index
    def _device_feat(self):
        """grid_feat mirrored into HBM once (re-uploaded only if the host array object changes).  The private device copy
        is converted to the split-fp16 layout the matrix-core kernel consumes directly, every row with its own power-of-two
        scale (avl_sim_prepare_map: same bytes, no per-query conversion work, and voxels observed once from far away --
        rows of magnitude 1e-6 ... 1e-15, vlmap_builder.py:166-168 -- score as accurately as any other row);
        self._sim_precision tells the kernels which form it has."""
        from .. import ops, parallel
        from ..device import DeviceArray
        with self._dev_lock:
            return self._device_feat_locked(ops, parallel, DeviceArray)

    def _device_feat_locked(self, ops, parallel, DeviceArray):
        from .. import _lib
        cur = _lib.current_device()
        if self._dev_feat is not None and getattr(self, "_dev_feat_device", cur) != cur:
            # the resident copy was uploaded before the caller selected this GPU (load_map's upload thread had nothing to go
            # by): pointers of another device would fault or crawl through peer access -- upload again, here
            self._dev_feat = None
        if self._dev_feat is None or self._dev_feat_src is not self.grid_feat:
            self._dev_feat_device = cur
            rank, ws = parallel.rank_world()
            self._rows = (0, len(self.grid_feat))
            if ws > 1 and self.shard_index_rows:
                self._rows = parallel.shard_rows(len(self.grid_feat), rank, ws)
            dev = DeviceArray.from_numpy(np.ascontiguousarray(self.grid_feat[self._rows[0]:self._rows[1]], dtype=np.float32))
            self._dev_feat_src = self.grid_feat
            self._sim_precision = "auto"
            if dev.shape[1] % 64 == 0 and dev.shape[0] > 0:
                # the compact form is read by every matrix-core kernel (resident, K-swap, streamed, column-block); its residual plane
                # is laid out in lines of 128 columns, other widths keep the 4-byte copy
                if self.compact_map and dev.shape[1] % 128 == 0:
                    raw = dev
                    dev = ops.prepare_map(raw, compact=True)
                    raw.free()                                        # the float32 device copy is not needed any more
                else:
                    dev = ops.prepare_map(dev, scaled=True)
                self._sim_precision = "prepared"
            self._dev_feat = dev
        return self._dev_feat

    def _device_pos(self):
        """grid_pos mirrored into HBM once (int32, re-uploaded only if the host array object changes)"""
        from ..device import DeviceArray
        if getattr(self, "_dev_pos", None) is None or self._dev_pos_src is not self.grid_pos:
            self._dev_pos = DeviceArray.from_numpy(np.ascontiguousarray(self.grid_pos, dtype=np.int32))
            self._dev_pos_src = self.grid_pos
        return self._dev_pos

    def _heat_plan(self):
        """ops.HeatPlan of this map's voxel positions (cell order + grid buffers), rebuilt only when the positions change; None
        for a map whose bounding box is too large for one (callers then use the stateless ops.heatmap_from_mask)."""
        from .. import ops
        pos = self._device_pos()
        if getattr(self, "_heat_plan_src", None) is not pos:
            old = getattr(self, "_heat_plan_obj", None)
            if old is not None:
                old.close()
            self._heat_plan_obj = ops.HeatPlan.for_positions(pos) if len(self.grid_pos) else None
            self._heat_plan_src = pos
        return self._heat_plan_obj

    def heatmap_from_mask(self, mask, cell_size, decay_rate):
        """visualize_utils.py:29-49 for this map's voxels: mask (N,) host/device -> device heat (planned when possible)"""
        from .. import ops
        plan = self._heat_plan()
        if plan is not None:
            return plan(mask, cell_size, decay_rate)
        return ops.heatmap_from_mask(self._device_pos(), mask, cell_size, decay_rate)

    def adopt_device_shard(self, shard) -> bool:
        """Take this rank's block of a freshly merged map (VLMapBuilder.map_shard: device tensors of
        parallel.merge_accumulator_sharded) as the resident copy the index kernels read -- no 4 GB host round trip after a
        multi-GPU build.  The host attributes (load_map) must describe the same map; returns False (and changes nothing) if the
        block is not the one shard_index_rows would upload."""
        from .. import _lib, ops, parallel
        if shard is None or self.grid_feat is None or int(shard["M"]) != len(self.grid_feat) or not self.shard_index_rows:
            return False
        rank, ws = parallel.rank_world()
        if tuple(shard["rows"]) != parallel.shard_rows(len(self.grid_feat), rank, ws):
            return False
        feat = shard["grid_feat"]
        D = int(feat.shape[1])
        with self._dev_lock:
            self._rows = tuple(shard["rows"])
            self._sim_precision = "auto"
            dev = feat
            if D % 64 == 0 and feat.shape[0] > 0:
                if self.compact_map and D % 128 == 0:
                    dev = ops.prepare_map(feat, compact=True)
                else:
                    dev = ops.prepare_map(feat.clone(), scaled=True)
                self._sim_precision = "prepared"
            self._dev_feat, self._dev_feat_src, self._dev_feat_device = dev, self.grid_feat, _lib.current_device()
        return True

    def _text_feats(self, names, use_multiple_templates=True, add_other=True):
        """landmark_text_feats with a per-instance cache keyed by the strings: the CLIP text tower costs ~1.6 ms per call, twice
        the similarity kernel, and a navigator asks for the same landmarks again and again"""
        key = (tuple(names), bool(use_multiple_templates), bool(add_other), id(self.clip_model), int(self.clip_feat_dim))
        cache = self.__dict__.setdefault("_text_cache", {})
        q = cache.get(key)
        if q is None:
            q, _ = landmark_text_feats(self.clip_model, list(names), self.clip_feat_dim, use_multiple_templates=use_multiple_templates,
                                       add_other=add_other)
            if len(cache) >= 256:
                cache.clear()
            cache[key] = q
        return q

    def _score_device(self, q):
        """row argmax of this rank's block as a DEVICE array (nothing crosses PCIe)"""
        from .. import ops
        feat = self._device_feat()
        _, am, _ = ops.sim_scores(feat, q, want_scores=False, want_argmax=True, precision=self._sim_precision)
        return am

    def _score(self, q, want_scores: bool):
        """(scores (N, Q) | None, argmax (N,)) as host arrays for ALL voxels: this rank's row block through the similarity
        kernel, the other ranks' blocks through one all_gather of results when the rows are sharded"""
        from .. import ops, parallel
        feat = self._device_feat()
        sc, am, _ = ops.sim_scores(feat, q, want_scores=want_scores, want_argmax=True, precision=self._sim_precision)
        from ..utils.clip_utils import _to_numpy
        to_np = lambda x: None if x is None else _to_numpy(x)        # DeviceArray or (adopted shard) torch tensor
        sc, am = to_np(sc), to_np(am)
        n = len(self.grid_feat)
        if self._rows != (0, n):
            am = parallel.gather_rows(am, n)
            sc = parallel.gather_rows(sc, n) if sc is not None else None
        return sc, am

    def index_queries(self, queries, want_scores: bool = False):
        """Row argmax (N,) int32 -- and, on request, scores (N, Q) float32 -- of the map against an explicit query matrix (Q, D):
        the entry for query sets that do not come from the CLIP text tower.  On a fused visual | audio map (BASELINE config 5:
        D = 512 + 1024, every query non-zero in one modality block) the non-zero column window of every query is detected and
        each group of queries is scored against its own columns of the compact resident copy (ops.sim_scores ->
        avl_sim_scores_blocks); same semantics as scores = grid_feat @ queries.T; np.argmax(scores, axis=1)
        (clip_utils.py:227-229, vlmap.py:123)."""
        q = np.ascontiguousarray(queries, dtype=np.float32)
        if q.ndim != 2 or self.grid_feat is None or q.shape[1] != self.grid_feat.shape[1]:
            raise ValueError(f"queries must be (Q, {None if self.grid_feat is None else self.grid_feat.shape[1]}), got {q.shape}")
        sc, am = self._score(q, want_scores=want_scores)
        return (am, sc) if want_scores else am

    def init_categories(self, categories: List[str]) -> np.ndarray:
        """scores_mat (N, Q) float32 cached on the instance.  Reference: vlmap.py:92-102."""
        self.categories = categories
        q = self._text_feats(self.categories)
        # one launch gives scores_mat AND its row argmax (same values, first maximum wins like np.argmax), so index_map
        # does not have to rescan the (N, Q) host matrix per query as upstream does (vlmap.py:123)
        self.scores_mat, self._argmax = self._score(q, want_scores=True)
        self._argmax_src = self.scores_mat
        return self.scores_mat

    def index_map(self, language_desc: str, with_init_cat: bool = True):
        """bool mask (N,): argmax over the query columns == category.  Reference: vlmap.py:104-125."""
        from .. import ops
        if with_init_cat and self.scores_mat is not None and self.categories is not None:
            from ..utils.index_utils import find_similar_category_id
            cat_id = find_similar_category_id(language_desc, self.categories)
            if getattr(self, "_argmax_src", None) is self.scores_mat:
                return self._argmax == cat_id
            return np.argmax(self.scores_mat, axis=1) == cat_id      # scores_mat was replaced from outside
        if with_init_cat:
            raise Exception(
                "Categories are not preloaded. Call init_categories(categories: List[str]) to initialize categories.")
        # fused path: scores and argmax never leave the GPU; the mask is compared and bit-packed there, 1 bit per voxel comes back
        q = self._text_feats([language_desc])
        self._device_feat()
        if self._rows == (0, len(self.grid_feat)):
            am = self._score_device(q)
            if isinstance(am, np.ndarray):
                return am == 0
            return ops.mask_bool_from_argmax(am, 0)
        _, am = self._score(q, want_scores=False)          # rows sharded over ranks: the argmax blocks are all-gathered
        return am == 0

    def customize_obstacle_map(self, potential_obstacle_names: List[str], obstacle_names: List[str], vis: bool = False):
        """Reference: vlmap.py:127-156.  The class scoring runs on the GPU; the 2-D smoothing (Map._dilate_map,
        map.py:169-181: cv2 resize + gaussian + dilation) is navigator-side image processing and needs OpenCV."""
        from ..utils.index_utils import get_dynamic_obstacles_map_3d
        if self.obstacles_cropped is None and self.obstacles_map is None:
            self.generate_obstacle_map()
        if not hasattr(self, "clip_model"):
            print("init_clip in customize obstacle map")
            self._init_clip()
        potential = list(cfg_get(self.map_config, "potential_obstacle_names"))
        feat, predict = self._device_feat(), None
        if self._rows != (0, len(self.grid_feat)):      # rows sharded over ranks: score here, hand the full argmax down
            q, _ = landmark_text_feats(self.clip_model, potential, self.clip_feat_dim, use_multiple_templates=True, add_other=True)
            predict = self._score(q, want_scores=False)[1]
        self.obstacles_new_cropped = get_dynamic_obstacles_map_3d(
            self.clip_model, self.obstacles_cropped, potential, list(cfg_get(self.map_config, "obstacle_names")), feat,
            self.grid_pos, self.rmin, self.cmin, self.clip_feat_dim, vis=vis, precision=self._sim_precision, predict=predict)
        self.obstacles_new_cropped = Map._dilate_map(self.obstacles_new_cropped == 0, cfg_get(self.map_config, "dilate_iter"),
                                                     cfg_get(self.map_config, "gaussian_sigma"))
        self.obstacles_new_cropped = self.obstacles_new_cropped == 0

    def get_pos(self, name: str):
        """Contours, centres and bounding boxes of a category on the full map.  Reference: vlmap.py:158-187.  The per-voxel
        work (argmax mask, top-down pooling) runs on the GPU; the 2-D morphology is the same SciPy calls as upstream; the
        island extraction is utils/navigation_utils.get_segment_islands_pos (OpenCV when installed)."""
        from scipy.ndimage import binary_closing, binary_dilation, gaussian_filter
        from ..utils.navigation_utils import get_segment_islands_pos
        from ..utils.visualize_utils import pool_3d_label_to_2d
        assert self.categories
        pc_mask = self.index_map(name, with_init_cat=True)
        mask_2d = pool_3d_label_to_2d(pc_mask, self._device_pos(), self.gs)
        mask_2d = mask_2d[self.rmin:self.rmax + 1, self.cmin:self.cmax + 1]
        foreground = binary_closing(mask_2d, iterations=3)
        foreground = gaussian_filter(foreground.astype(float), sigma=0.8, truncate=3)
        foreground = foreground > 0.5
        foreground = binary_dilation(foreground)
        self._last_foreground = foreground
        contours, centers, bbox_list, _ = get_segment_islands_pos(foreground, 1)
        for i in range(len(contours)):          # whole-map positions (upstream adds rmin to both row bounds: kept)
            centers[i][0] += self.rmin
            centers[i][1] += self.cmin
            bbox_list[i][0] += self.rmin
            bbox_list[i][1] += self.rmin
            bbox_list[i][2] += self.cmin
            bbox_list[i][3] += self.cmin
            contours[i][:, 0] += self.rmin
            contours[i][:, 1] += self.cmin
        return contours, centers, bbox_list
