"""Import the upstream reference's hot-path modules in THIS container.

Test/fixture infrastructure only.  The reference (/root/reference) is plain
Python whose hot path depends only on numpy/scipy; its heavy, absent
dependencies (h5py, cv2, clip, omegaconf, open3d, ...) are replaced by
MagicMock so that the *real* reference functions
  avlmaps.utils.mapping_utils / clip_utils / visualize_utils / index_utils
  avlmaps.map.vlmap_builder.VLMapBuilder, avlmaps.map.vlmap.VLMap, avlmaps.map.map.Map
can be executed on synthetic inputs to produce golden vectors.
Nothing from the reference is copied: it is imported from where it lies.
"""
import os
import sys
import types
from unittest.mock import MagicMock

REF_ROOT = os.environ.get("AVLMAPS_REFERENCE", "/root/reference")

_STUBS = [
    "h5py", "cv2", "torchvision", "torchvision.transforms", "clip", "omegaconf",
    "hydra", "open3d", "gdown", "shapely", "shapely.geometry", "timm", "encoding",
    "openai", "pyvisgraph", "habitat_sim", "matplotlib.patches",
]


def import_reference():
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError(f"reference not found at {REF_ROOT}")
    for name in _STUBS:
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = MagicMock(name=name)
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    # bypass avlmaps/map/__init__.py (pulls hloc / AudioCLIP)
    import avlmaps  # noqa: F401
    pkg = types.ModuleType("avlmaps.map")
    pkg.__path__ = [os.path.join(REF_ROOT, "avlmaps", "map")]
    sys.modules["avlmaps.map"] = pkg
    for name in ["avlmaps.lseg.modules.models.lseg_net", "avlmaps.lseg.additional_utils.models",
                 "avlmaps.utils.lseg_utils", "avlmaps.utils.navigation_utils"]:
        sys.modules[name] = MagicMock(name=name)
    import importlib
    mods = {}
    for name in ["avlmaps.utils.mapping_utils", "avlmaps.utils.clip_utils",
                 "avlmaps.utils.visualize_utils", "avlmaps.utils.index_utils",
                 "avlmaps.map.vlmap_builder", "avlmaps.map.map", "avlmaps.map.vlmap"]:
        mods[name.split(".")[-1]] = importlib.import_module(name)
    return mods
