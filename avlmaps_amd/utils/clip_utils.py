"""Text-query scoring with the interface of the reference's avlmaps/utils/clip_utils.py.

get_lseg_score keeps the reference's semantics (clip_utils.py:196-242): optional "other" column, 63 prompt
templates averaged WITHOUT re-normalisation, raw dot product -- but the (N x D) . (D x Q) contraction runs in
the HIP similarity kernel instead of NumPy/OpenBLAS.  CLIP itself stays on PyTorch-ROCm.
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np


def _build_templates() -> List[str]:
    """The 63 prompt templates of clip_utils.py:10-74 (CLIP's prompt-engineering set as used by VLMaps),
    same order, including the upstream 'picture of of' typo (the text encoder sees exactly these strings)."""
    t = ["There is {} in the scene.", "There is the {} in the scene.", "a photo of {} in the scene.",
         "a photo of the {} in the scene.", "a photo of one {} in the scene."]
    t += [f"I took a picture of of {x}{{}}." for x in ("", "my ", "the ")]
    t += [f"a photo of {x}{{}}." for x in ("", "my ", "the ", "one ", "many ")]
    for adj in ("good", "bad"):
        t += [f"a {adj} photo of {{}}.", f"a {adj} photo of the {{}}."]
    for adj in ("nice", "cool", "weird", "small", "large", "clean", "dirty"):
        t += [f"a photo of a {adj} {{}}.", f"a photo of the {adj} {{}}."]
    for adj in ("bright", "dark"):
        t += [f"a {adj} photo of {{}}.", f"a {adj} photo of the {{}}."]
    t += ["a photo of a hard to see {}.", "a photo of the hard to see {}."]
    for adj in ("low resolution", "cropped", "close-up", "jpeg corrupted", "blurry", "pixelated"):
        t += [f"a {adj} photo of {{}}.", f"a {adj} photo of the {{}}."]
    t += ["a black and white photo of the {}.", "a black and white photo of {}."]
    for noun in ("plastic", "toy", "plushie", "cartoon"):
        t += [f"a {noun} {{}}.", f"the {noun} {{}}."]
    t += ["an embroidered {}.", "the embroidered {}.", "a painting of the {}.", "a painting of a {}."]
    assert len(t) == 63
    return t


multiple_templates = _build_templates()


def get_text_feats(in_text: Sequence[str], clip_model, clip_feat_dim: int, batch_size: int = 64) -> np.ndarray:
    """CLIP text embeddings, each row L2-normalised, float32 (len(in_text), clip_feat_dim).
    Reference: clip_utils.py:133-149.  `clip_model` is an OpenAI-CLIP style model (encode_text); the tokenizer is
    clip.tokenize, or clip_model.tokenize when the model object carries its own."""
    import torch
    tokenize = getattr(clip_model, "tokenize", None)
    if tokenize is None:
        import clip  # OpenAI CLIP (pip git+https://github.com/openai/CLIP.git), as upstream
        tokenize = clip.tokenize
    device = "cuda" if torch.cuda.is_available() else "cpu"
    tokens = tokenize(list(in_text)).to(device)
    feats = np.zeros((len(in_text), clip_feat_dim), dtype=np.float32)
    i = 0
    while i < len(tokens):
        bs = min(len(in_text) - i, batch_size)
        with torch.no_grad():
            b = clip_model.encode_text(tokens[i:i + bs]).float()
        b /= b.norm(dim=-1, keepdim=True)
        feats[i:i + bs] = b.cpu().numpy().astype(np.float32)
        i += bs
    return feats


def get_text_feats_multiple_templates(in_text, clip_model, clip_feat_dim, batch_size=64) -> np.ndarray:
    """Reference: clip_utils.py:152-160."""
    prompts = [t.format(lm) for lm in in_text for t in multiple_templates]
    f = get_text_feats(prompts, clip_model, clip_feat_dim)
    return np.mean(f.reshape((-1, len(multiple_templates), f.shape[-1])), axis=1)


def landmark_text_feats(clip_model, landmarks: list, clip_feat_dim: int, use_multiple_templates=False, add_other=True):
    """(Q, D) query matrix + the landmark list actually scored (clip_utils.py:213-225)."""
    landmarks_other = landmarks
    if add_other and landmarks_other[-1] != "other":
        landmarks_other = landmarks + ["other"]
    if use_multiple_templates:
        prompts = [t.format(lm) for lm in landmarks_other for t in multiple_templates]
        f = get_text_feats(prompts, clip_model, clip_feat_dim)
        f = np.mean(f.reshape((-1, len(multiple_templates), f.shape[-1])), axis=1)
    else:
        f = get_text_feats(landmarks_other, clip_model, clip_feat_dim)
    return np.ascontiguousarray(f, dtype=np.float32), landmarks_other


def get_lseg_score(clip_model, landmarks: list, lseg_map, clip_feat_dim: int, use_multiple_templates: bool = False,
                   avg_mode: int = 0, add_other=True, precision: str = "auto"):
    """scores (N, Q) float32 = map_feats @ text_feats.T.  Reference: clip_utils.py:196-242.

    lseg_map: (h, w, D) / (N, D) numpy array, or a device-resident (N, D) array (torch CUDA tensor /
    avlmaps_amd.device.DeviceArray) to skip the upload.  avg_mode=1 (average scores instead of features)
    scores every template separately and averages on the host, as upstream."""
    from .. import ops
    landmarks_other = landmarks
    if add_other and landmarks_other[-1] != "other":
        landmarks_other = landmarks + ["other"]
    if isinstance(lseg_map, np.ndarray):
        lseg_map = np.ascontiguousarray(lseg_map.reshape((-1, lseg_map.shape[-1])), dtype=np.float32)
    if use_multiple_templates and avg_mode == 1:
        prompts = [t.format(lm) for lm in landmarks_other for t in multiple_templates]
        f = get_text_feats(prompts, clip_model, clip_feat_dim)
        sc, _, _ = ops.sim_scores(lseg_map, f, want_argmax=False, precision=precision)
        sc = _to_numpy(sc)
        return np.mean(sc.reshape((-1, len(landmarks_other), len(multiple_templates))), axis=2)
    f, _ = landmark_text_feats(clip_model, landmarks, clip_feat_dim, use_multiple_templates, add_other)
    sc, _, _ = ops.sim_scores(lseg_map, f, want_argmax=False, precision=precision)
    return _to_numpy(sc)


def _to_numpy(x):
    if isinstance(x, np.ndarray):
        return x
    if hasattr(x, "numpy") and not hasattr(x, "cpu"):
        return x.numpy()
    return x.cpu().numpy()
