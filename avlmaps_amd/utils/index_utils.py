"""Category matching helper with the reference's name (avlmaps/utils/index_utils.py:8-32).

Upstream asks an OpenAI model to pick the closest category; that network call is outside the hot path.  This mirror resolves
exact / case-insensitive / substring matches locally and hands everything else to a pluggable matcher -- the navigator's LLM
hook goes there (set_category_matcher, or the `matcher=` argument); without one an unmatched name raises."""
from __future__ import annotations

from typing import Callable, List, Optional

_MATCHER: Optional[Callable[[str, List[str]], int]] = None


def set_category_matcher(matcher: Optional[Callable[[str, List[str]], int]]) -> None:
    """matcher(class_name, classes_list) -> index into classes_list, used when the local rules find no unique match (what
    upstream's find_similar_category_id asks its LLM, index_utils.py:8-32).  None removes it."""
    global _MATCHER
    _MATCHER = matcher


def find_similar_category_id(class_name: str, classes_list: List[str], matcher: Optional[Callable[[str, List[str]], int]] = None) -> int:
    if class_name in classes_list:
        return classes_list.index(class_name)
    low = [c.lower() for c in classes_list]
    name = class_name.lower().strip()
    if name in low:
        return low.index(name)
    hits = [i for i, c in enumerate(low) if name in c or c in name]
    if len(hits) == 1:
        return hits[0]
    fn = matcher or _MATCHER
    if fn is not None:
        i = int(fn(class_name, list(classes_list)))
        if not 0 <= i < len(classes_list):
            raise ValueError(f"the category matcher returned {i} for {class_name!r}: not an index into {len(classes_list)} categories")
        return i
    raise KeyError(f"{class_name!r} does not match one of the initialised categories {classes_list}; upstream delegates this to an "
                   "LLM (index_utils.py:8-32): plug one in with utils.index_utils.set_category_matcher")


def get_dynamic_obstacles_map_3d(clip_model, obstacles_cropped, potential_obstacle_classes, obstacle_classes, grid_feat,
                                 grid_pos, rmin, cmin, clip_feat_dim, use_multiple_templates=True, avg_mode=0, vis=False,
                                 precision="auto", predict=None):
    """Cropped top-down map, True = free, after keeping only voxels whose best class is one of `obstacle_classes`.
    Reference: avlmaps/utils/index_utils.py:138-184 -- the same matmul + argmax as index_map (:153-161), here the fused
    similarity kernel (scores are never materialised), then one scatter kernel fed by the device-resident argmax.
    `grid_feat` may be a host array or a device-resident (N, D) array."""
    import numpy as np
    from .. import ops
    from .clip_utils import landmark_text_feats, _to_numpy
    if predict is not None:         # the caller already has the class argmax of every voxel (row-sharded VLMap)
        predict = np.ascontiguousarray(predict, dtype=np.int32)
    elif avg_mode != 0:
        from .clip_utils import get_lseg_score
        scores = get_lseg_score(clip_model, list(potential_obstacle_classes), grid_feat, clip_feat_dim,
                                use_multiple_templates=use_multiple_templates, avg_mode=avg_mode, precision=precision)
        predict = np.argmax(scores, axis=1).astype(np.int32)
    else:
        q, _ = landmark_text_feats(clip_model, list(potential_obstacle_classes), clip_feat_dim, use_multiple_templates, True)
        if isinstance(grid_feat, np.ndarray):
            grid_feat = np.ascontiguousarray(grid_feat, dtype=np.float32)
        _, predict, _ = ops.sim_scores(grid_feat, q, want_scores=False, want_argmax=True, precision=precision)   # stays in HBM
    obs_inds = [i for obs_name in obstacle_classes for i, po in enumerate(potential_obstacle_classes) if obs_name == po]
    print("obs_inds: ", obs_inds)
    n_classes = len(potential_obstacle_classes) + (0 if potential_obstacle_classes[-1] == "other" else 1)
    return ops.obstacle_scatter(grid_pos, predict, obs_inds, n_classes, obstacles_cropped, rmin, cmin)          # avl_obstacle_scatter
