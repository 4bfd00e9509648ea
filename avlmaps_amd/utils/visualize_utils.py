"""Reference interface of avlmaps/utils/visualize_utils.py for the pieces on the hot path."""
from __future__ import annotations

import numpy as np


def get_heatmap_from_mask_3d(pc: np.ndarray, mask: np.ndarray, cell_size: float = 0.05, decay_rate: float = 0.01) -> np.ndarray:
    """Nearest-target distance-decay heat per voxel, (N,) float32.  Reference: visualize_utils.py:29-49
    (an O(N_other * N_target) Python loop upstream; here the windowed / brute-force HIP kernels)."""
    from .. import ops
    mask = np.asarray(mask)
    if mask.sum() == 0:
        raise ValueError("attempt to get argmin of an empty sequence")   # what np.argmin raises upstream
    # the same grid_pos array comes back on every query of a map: its device copy and cell order are kept between calls --
    # only when the array reaches the library as it is (a converted temporary is a new object every time: stateless path)
    pos = np.ascontiguousarray(pc, dtype=np.int32)
    return ops.heatmap_from_mask(pos, mask.astype(np.uint8), cell_size, decay_rate, reuse_plan=pos is pc)


def pool_3d_label_to_2d(mask_3d: np.ndarray, grid_pos: np.ndarray, gs: int) -> np.ndarray:
    """Top-down OR-pooling of a per-voxel mask, (gs, gs) bool.  Reference: visualize_utils.py:77-83 (a Python loop over all
    voxels upstream; here one scatter kernel, avl_pool_label_2d).  grid_pos / mask_3d may be device-resident."""
    from .. import ops
    return ops.pool_label_2d(mask_3d, grid_pos, int(gs))
