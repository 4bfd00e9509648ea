"""Pixel-feature extraction with the interface of the reference's avlmaps/utils/lseg_utils.py, kept ON THE DEVICE.

get_lseg_feat mirrors the reference's evaluation protocol (lseg_utils.py:20-119): resize the long side to `base_size`
(bilinear, align_corners), pad to `crop_size`, slide crop_size windows with stride int(crop_size * 2/3), average the
overlaps, crop back.  Differences, all about data movement only:
  * the result stays on the GPU and is returned CHANNELS-LAST (Hf, Wf, D) -- exactly what the builder kernel gathers from;
    the reference copies a (1, D, Hf, Wf) array to the host every frame (369 MB at 1080x720, lseg_utils.py:101-102);
  * the unused logits accumulation and the visualisation branch are dropped (labels are still passed to the model).
The model itself (LSegEncNet) is the upstream PyTorch module running on PyTorch-ROCm.
"""
from __future__ import annotations

import math

import numpy as np


def resize_image(img, h, w):
    import torch.nn.functional as F
    return F.interpolate(img, (h, w), mode="bilinear", align_corners=True)


def pad_image(img, mean, std, crop_size):
    """pad bottom/right up to crop_size with the normalised value of a black pixel (additional_utils/models.py:145-156)"""
    import torch
    import torch.nn.functional as F
    b, c, h, w = img.shape
    padh = crop_size - h if h < crop_size else 0
    padw = crop_size - w if w < crop_size else 0
    if padh == 0 and padw == 0:
        return img
    pad_values = -np.array(mean) / np.array(std)
    return torch.stack([F.pad(img[:, i], (0, padw, 0, padh), value=float(pad_values[i])) for i in range(c)], dim=1)


def crop_image(img, h0, h1, w0, w1):
    return img[:, :, h0:h1, w0:w1]


def default_transform(image: np.ndarray):
    """ToTensor + Normalize(0.5, 0.5) of the reference (vlmap_builder.py:257-262) without torchvision"""
    import torch
    t = torch.from_numpy(np.array(image, copy=True)).permute(2, 0, 1).float().div_(255.0)
    return (t - 0.5) / 0.5


def get_lseg_feat(model, image: np.ndarray, labels, transform, device, crop_size=480, base_size=520,
                  norm_mean=(0.5, 0.5, 0.5), norm_std=(0.5, 0.5, 0.5), vis=False, channels_last=True):
    """image (H, W, 3) uint8 -> pixel embeddings on `device`: (Hf, Wf, D) float32 if channels_last (default), else the
    reference layout (1, D, Hf, Wf).  Reference: lseg_utils.py:20-119."""
    import torch
    image = (transform or default_transform)(image).unsqueeze(0).to(device)
    batch, _, h, w = image.shape
    stride = int(crop_size * (2.0 / 3.0))
    long_size = base_size
    if h > w:
        height = long_size
        width = int(1.0 * w * long_size / h + 0.5)
        short_size = width
    else:
        width = long_size
        height = int(1.0 * h * long_size / w + 0.5)
        short_size = height
    cur_img = resize_image(image, height, width)
    with torch.no_grad():
        if long_size <= crop_size:
            pad_img = pad_image(cur_img, norm_mean, norm_std, crop_size)
            outputs, _ = model(pad_img, labels)
            outputs = crop_image(outputs, 0, height, 0, width)
        else:
            pad_img = pad_image(cur_img, norm_mean, norm_std, crop_size) if short_size < crop_size else cur_img
            _, _, ph, pw = pad_img.shape
            assert ph >= height and pw >= width
            h_grids = int(math.ceil(1.0 * (ph - crop_size) / stride)) + 1
            w_grids = int(math.ceil(1.0 * (pw - crop_size) / stride)) + 1
            outputs = torch.zeros((batch, model.out_c, ph, pw), dtype=image.dtype, device=device)
            count_norm = torch.zeros((batch, 1, ph, pw), dtype=image.dtype, device=device)
            for idh in range(h_grids):
                for idw in range(w_grids):
                    h0, w0 = idh * stride, idw * stride
                    h1, w1 = min(h0 + crop_size, ph), min(w0 + crop_size, pw)
                    pad_crop_img = pad_image(crop_image(pad_img, h0, h1, w0, w1), norm_mean, norm_std, crop_size)
                    output, _ = model(pad_crop_img, labels)
                    outputs[:, :, h0:h1, w0:w1] += crop_image(output, 0, h1 - h0, 0, w1 - w0).to(outputs.dtype)
                    count_norm[:, :, h0:h1, w0:w1] += 1
            assert (count_norm == 0).sum() == 0
            outputs = (outputs / count_norm)[:, :, :height, :width]
    if channels_last:
        return outputs[0].permute(1, 2, 0).float().contiguous()
    return outputs.float()
