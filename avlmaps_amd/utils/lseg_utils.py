"""Pixel-feature extraction with the interface of the reference's avlmaps/utils/lseg_utils.py, kept ON THE DEVICE.

The reference's protocol (lseg_utils.py:20-119): resize the long side to `base_size` (bilinear, align_corners), pad to
`crop_size`, slide crop_size windows with stride int(crop_size * 2/3) -- each window padded to crop_size and sent through the
model on its own --, add the window outputs into a canvas, divide by the overlap count, crop back, copy (1, D, Hf, Wf) to the host.

Here the same numbers come out of a different data flow:
  * WindowPlan is the geometry alone (resized size, canvas, window origins);
  * the resized image is written once into a canvas that is already large enough for every window, so the windows are plain
    views (unfold) -- one batch (G, 3, crop, crop), ONE call of the model for all windows of a frame;
  * the library's avl_lseg_merge_windows sums the overlapping windows in the reference's order, divides by the count and writes
    the result CHANNELS-LAST (Hf, Wf, D) float32 in one pass -- exactly what the builder kernel gathers from -- instead of G
    read-modify-write passes over a canvas, a division pass, a permute copy and the reference's 369 MB device-to-host copy
    per frame (lseg_utils.py:101-102).
The model itself (LSegEncNet) is the upstream PyTorch module running on PyTorch-ROCm.
"""
from __future__ import annotations

import math
from typing import List, NamedTuple, Tuple

import numpy as np


class WindowPlan(NamedTuple):
    """geometry of the sliding-window evaluation of one (h, w) image (lseg_utils.py:36-60, :78-84)"""
    height: int                      # size of the resized image = size of the feature map
    width: int
    canvas: Tuple[int, int]          # padded canvas that holds every crop_size window completely
    origins: List[Tuple[int, int]]   # (h0, w0) of the windows, reference loop order (rows of windows first)
    crop: int

    @staticmethod
    def make(h: int, w: int, crop_size: int, base_size: int) -> "WindowPlan":
        if h > w:
            height, width = base_size, int(1.0 * w * base_size / h + 0.5)
        else:
            width, height = base_size, int(1.0 * h * base_size / w + 0.5)
        if base_size <= crop_size:                                   # one window holds the whole resized image
            return WindowPlan(height, width, (crop_size, crop_size), [(0, 0)], crop_size)
        stride = int(crop_size * (2.0 / 3.0))
        ph, pw = max(height, crop_size), max(width, crop_size)       # the reference pads the short side up to one window
        hg = int(math.ceil(1.0 * (ph - crop_size) / stride)) + 1
        wg = int(math.ceil(1.0 * (pw - crop_size) / stride)) + 1
        origins = [(i * stride, j * stride) for i in range(hg) for j in range(wg)]
        # the last windows of a row / column reach past (ph, pw): upstream cuts them and pads the cut crop back to crop_size
        # with the same value -- a canvas of this size holds them whole, with the same pixels
        return WindowPlan(height, width, ((hg - 1) * stride + crop_size, (wg - 1) * stride + crop_size), origins, crop_size)


def default_transform(image: np.ndarray):
    """ToTensor + Normalize(0.5, 0.5) of the reference (vlmap_builder.py:257-262) without torchvision"""
    import torch
    t = torch.from_numpy(np.array(image, copy=True)).permute(2, 0, 1).float().div_(255.0)
    return (t - 0.5) / 0.5


def window_batch(image, plan: WindowPlan, norm_mean, norm_std):
    """(1, 3, h, w) normalised image on the device -> (G, 3, crop, crop): the windows the reference feeds the model one by one.
    Pixels outside the resized image hold the normalised value of a black pixel (additional_utils/models.py:145-156)."""
    import torch
    import torch.nn.functional as F
    cur = F.interpolate(image, (plan.height, plan.width), mode="bilinear", align_corners=True)
    pad = (-torch.tensor(norm_mean, dtype=cur.dtype) / torch.tensor(norm_std, dtype=cur.dtype)).to(cur.device)
    canvas = pad.view(1, -1, 1, 1).expand(1, cur.shape[1], *plan.canvas).contiguous()
    canvas[:, :, :plan.height, :plan.width] = cur
    if len(plan.origins) == 1:
        return canvas
    stride = int(plan.crop * (2.0 / 3.0))
    win = canvas.unfold(2, plan.crop, stride).unfold(3, plan.crop, stride)          # (1, 3, hg, wg, crop, crop): views
    return win.permute(0, 2, 3, 1, 4, 5).reshape(-1, canvas.shape[1], plan.crop, plan.crop)


def merge_windows(outputs, plan: WindowPlan):
    """(G, D, crop, crop) window outputs (float32 / float16, device) -> (Hf, Wf, D) float32 channels-last: overlaps summed in
    window order and divided by their count (avl_lseg_merge_windows; lseg_utils.py:85-99)"""
    import torch
    from .. import _lib
    from ..device import torch_stream_ptr
    G, D = int(outputs.shape[0]), int(outputs.shape[1])
    if G > 64:
        raise ValueError(f"{G} sliding windows per frame: avl_lseg_merge_windows takes at most 64 (a 480-pixel window grid over an image "
                         "whose long side is resized to 520 has 1-2); use a larger crop_size")
    if outputs.dtype not in (torch.float32, torch.float16):
        outputs = outputs.float()
    outputs = outputs.contiguous()
    origin = np.ascontiguousarray(plan.origins, dtype=np.int32)
    out = torch.empty((plan.height, plan.width, D), dtype=torch.float32, device=outputs.device)
    _lib.check(_lib.load().avl_lseg_merge_windows(outputs.data_ptr(), int(outputs.dtype == torch.float16), G, D, plan.crop, origin.ctypes.data,
                                                  plan.height, plan.width, out.data_ptr(), torch_stream_ptr()), "avl_lseg_merge_windows")
    return out


def get_lseg_feat(model, image: np.ndarray, labels, transform, device, crop_size=480, base_size=520,
                  norm_mean=(0.5, 0.5, 0.5), norm_std=(0.5, 0.5, 0.5), vis=False, channels_last=True, window_batch_size=None):
    """image (H, W, 3) uint8 -> pixel embeddings on `device`: (Hf, Wf, D) float32 if channels_last (default), else the
    reference layout (1, D, Hf, Wf).  Reference: lseg_utils.py:20-119 (the logits / visualisation branch is not part of the
    map-building path).  window_batch_size: windows per model call; default: all windows of the frame in one call up to 8 (the
    reference's shapes have 1-2 windows; activation memory grows with the batch, so larger grids go in chunks of 8 -- the merge
    arithmetic is the same bits either way, the model's own kernels may round a batched call differently from single calls)."""
    import torch
    img = (transform or default_transform)(image).unsqueeze(0).to(device)
    plan = WindowPlan.make(int(img.shape[2]), int(img.shape[3]), int(crop_size), int(base_size))
    with torch.no_grad():
        batch = window_batch(img, plan, norm_mean, norm_std)
        step = int(window_batch_size or min(int(batch.shape[0]), 8))
        outs = [model(batch[i:i + step], labels)[0] for i in range(0, batch.shape[0], step)]
        outputs = outs[0] if len(outs) == 1 else torch.cat(outs, dim=0)
        feat = merge_windows(outputs, plan)
    if channels_last:
        return feat
    return feat.permute(2, 0, 1).unsqueeze(0).contiguous()
