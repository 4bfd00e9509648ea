"""Island extraction with the reference's name and return convention (avlmaps/utils/navigation_utils.py:10-36), used by
VLMap.get_pos.  Upstream calls cv2.findContours(RETR_EXTERNAL, CHAIN_APPROX_SIMPLE); with OpenCV installed this module
does exactly that.  Without it (OpenCV is not part of the ROCm image) the 8-connected components come from
scipy.ndimage.label: bounding boxes and centres -- which upstream derives from the contour extents, i.e. from the
component extents -- are identical; the contour itself is the component's full outer boundary (Moore tracing) rather than
OpenCV's corner-compressed polyline, and components are listed bottom-up like OpenCV lists them."""
from __future__ import annotations

import numpy as np


def _trace_boundary(comp: np.ndarray, start):
    """outer boundary of an 8-connected component (bool image), Moore-neighbour tracing from its top-left pixel"""
    H, W = comp.shape
    nb = [(0, -1), (-1, -1), (-1, 0), (-1, 1), (0, 1), (1, 1), (1, 0), (1, -1)]   # clockwise starting west
    pts = [start]
    cur, back = start, 0
    if comp.sum() == 1:
        return np.array(pts)
    for _ in range(4 * comp.size + 8):
        found = False
        for k in range(8):
            d = (back + k) % 8
            r, c = cur[0] + nb[d][0], cur[1] + nb[d][1]
            if 0 <= r < H and 0 <= c < W and comp[r, c]:
                back = (d + 5) % 8          # restart the scan just past the pixel we came from
                cur = (r, c)
                found = True
                break
        if not found or cur == start:
            break
        pts.append(cur)
    return np.array(pts)


def get_segment_islands_pos(segment_map, label_id, detect_internal_contours=False):
    """-> (contours [(n_i, 2) arrays of (row, col)], centers [[row, col]], bbox_list [[rmin, rmax, cmin, cmax]], hierarchy)"""
    mask = (np.asarray(segment_map) == label_id).astype(np.uint8)
    try:
        import cv2
    except Exception:
        cv2 = None
    hierarchy = None
    if cv2 is not None:
        mode = cv2.RETR_TREE if detect_internal_contours else cv2.RETR_EXTERNAL
        contours, hierarchy = cv2.findContours(mask, mode, cv2.CHAIN_APPROX_SIMPLE)
        contours_list = [np.stack([c.reshape((-1, 2))[:, 1], c.reshape((-1, 2))[:, 0]], axis=1) for c in contours]
    else:
        if detect_internal_contours:
            raise NotImplementedError("internal contours (cv2.RETR_TREE) need OpenCV")
        from scipy import ndimage
        lab, n = ndimage.label(mask, structure=np.ones((3, 3), dtype=int))
        contours_list = []
        for k in range(n, 0, -1):                      # raster order of the first pixel, reversed (bottom-up)
            comp = lab == k
            r0 = int(np.argmax(comp.any(axis=1)))
            c0 = int(np.argmax(comp[r0]))
            contours_list.append(_trace_boundary(comp, (r0, c0)))
    centers_list, bbox_list = [], []
    for c in contours_list:
        xmin, xmax, ymin, ymax = np.min(c[:, 0]), np.max(c[:, 0]), np.min(c[:, 1]), np.max(c[:, 1])
        bbox_list.append([xmin, xmax, ymin, ymax])
        centers_list.append([(xmin + xmax) / 2, (ymin + ymax) / 2])
    return contours_list, centers_list, bbox_list, hierarchy
