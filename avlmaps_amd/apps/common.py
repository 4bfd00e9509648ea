"""Config + stand-in models shared by the CLI apps.

The reference drives its apps with hydra (config/map_creation_cfg.yaml, config/map_indexing_cfg.yaml); here a plain YAML
file (or the built-in defaults below, = config/map_config/vlmaps.yaml + config/params/default.yaml) is enough."""
from __future__ import annotations

import hashlib
from pathlib import Path

import numpy as np

DEFAULTS = {
    "params": {"gs": 1000, "cs": 0.05, "camera_height": 1.5},
    "map_config": {
        "map_type": "vlmap",
        "pose_info": {"pose_type": "mobile_base", "camera_height": 1.5, "base2cam_rot": [1, 0, 0, 0, -1, 0, 0, 0, -1],
                      "base_forward_axis": [0, 0, -1], "base_left_axis": [-1, 0, 0], "base_up_axis": [0, 1, 0]},
        "cam_calib_mat": [540, 0, 540, 0, 540, 360, 0, 0, 1], "grid_size": 1000, "cell_size": 0.05,
        "depth_sample_rate": 100, "dilate_iter": 3, "gaussian_sigma": 1.0,
        "potential_obstacle_names": ["chair", "wall", "wall above the door", "table", "window", "floor", "stairs", "other"],
        "obstacle_names": ["wall", "chair", "table", "window", "stairs", "other"],
    },
}


class Cfg(dict):
    """attribute + item access, nested (stand-in for omegaconf.DictConfig)"""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


def to_cfg(d):
    return Cfg({k: to_cfg(v) for k, v in d.items()}) if isinstance(d, dict) else d


def load_config(path=None, overrides=None):
    import copy
    cfg = copy.deepcopy(DEFAULTS)
    if path:
        import yaml
        user = yaml.safe_load(Path(path).read_text()) or {}
        for k, v in user.items():
            if isinstance(v, dict) and isinstance(cfg.get(k), dict):
                _deep_update(cfg[k], v)
            else:
                cfg[k] = v
    for k, v in (overrides or {}).items():
        node = cfg
        parts = k.split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = v
    cfg["map_config"]["grid_size"] = cfg["map_config"].get("grid_size", cfg["params"]["gs"])
    cfg["map_config"]["cell_size"] = cfg["map_config"].get("cell_size", cfg["params"]["cs"])
    return to_cfg(cfg)


def _deep_update(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _deep_update(dst[k], v)
        else:
            dst[k] = v


class HashFeatureExtractor:
    """Stand-in for LSeg when no checkpoint is available (demo / smoke runs): a fixed random projection of a small
    colour + position code to D channels, normalised to the LSeg logit scale, computed on the GPU, channels-last."""

    def __init__(self, D=512, scale=8, seed=0):
        import torch
        self.D, self.scale = D, scale
        g = torch.Generator(device="cuda").manual_seed(seed)
        self.proj = torch.randn((5, D), device="cuda", generator=g)

    def __call__(self, rgb):
        import torch
        t = torch.from_numpy(np.array(rgb, copy=True)).cuda().float().div_(255.0)
        H, W, _ = t.shape
        t = t[:: self.scale // 4 or 1, :: self.scale // 4 or 1][: max(1, H // 2), : max(1, W // 2)]
        h, w, _ = t.shape
        yy = torch.linspace(0, 1, h, device="cuda").view(h, 1, 1).expand(h, w, 1)
        xx = torch.linspace(0, 1, w, device="cuda").view(1, w, 1).expand(h, w, 1)
        f = torch.cat([t, yy, xx], dim=2) @ self.proj
        f = f / f.norm(dim=2, keepdim=True) * 14.2857
        return f.half().float().contiguous()


class HashClip:
    """Stand-in for CLIP's text tower (demo / smoke runs): deterministic unit vectors derived from the prompt text."""

    def __init__(self, D=512):
        self.D = D
        self._texts = []

    def tokenize(self, texts):
        import torch
        base = len(self._texts)
        self._texts.extend(texts)
        return torch.arange(base, base + len(texts), dtype=torch.int64)

    def encode_text(self, ids):
        import torch
        rows = []
        for i in ids.cpu().tolist():
            seed = int.from_bytes(hashlib.sha256(self._texts[i].encode()).digest()[:8], "little")
            rows.append(np.random.default_rng(seed).standard_normal(self.D).astype(np.float32))
        return torch.from_numpy(np.stack(rows)).to(ids.device)
