"""avlmaps_amd.compat: in-place switch of an upstream-shaped package onto the HIP path (mechanics only, no GPU)."""
import importlib
import os
import sys
import textwrap

import pytest


def _fake_upstream(tmp_path, name):
    root = tmp_path / name
    for sub in ("", "utils", "map"):
        (root / sub).mkdir(parents=True, exist_ok=True)
        (root / sub / "__init__.py").write_text("")
    (root / "utils" / "clip_utils.py").write_text("def get_lseg_score(*a, **k):\n    return 'upstream-score'\n")
    (root / "utils" / "index_utils.py").write_text(
        "def get_lseg_score(*a, **k):\n    return 'upstream-score-dup'\n"
        "def get_dynamic_obstacles_map_3d(*a, **k):\n    return 'upstream-obst'\n")
    (root / "utils" / "visualize_utils.py").write_text(
        "def get_heatmap_from_mask_3d(*a, **k):\n    return 'upstream-heat'\n"
        "def pool_3d_label_to_2d(*a, **k):\n    return 'upstream-pool'\n")
    (root / "map" / "vlmap_builder.py").write_text(textwrap.dedent("""
        class VLMapBuilder:
            def __init__(self):
                self.data_dir = 'd'
            def create_mobile_base_map(self):
                return 'upstream-build'
    """))
    # modules that import the names, like avlmaps/map/vlmap.py:24 and avlmaps/map/avlmap.py:15 do
    (root / "map" / "vlmap.py").write_text(f"from {name}.utils.clip_utils import get_lseg_score\n"
                                           f"from {name}.utils.index_utils import get_dynamic_obstacles_map_3d\n")
    (root / "map" / "avlmap.py").write_text(f"from {name}.utils.visualize_utils import get_heatmap_from_mask_3d\n")
    (root / "robot").mkdir(exist_ok=True)
    (root / "robot" / "__init__.py").write_text("")
    (root / "robot" / "habitat_lang_robot.py").write_text(textwrap.dedent("""
        def find_similar_category_id(name, categories):
            return categories.index(name)
        class HabitatLanguageRobot:
            def get_vl_distribution_map_3d(self, name, decay_rate=0.1):
                return 'upstream-decay-loop'
    """))
    sys.path.insert(0, str(tmp_path))
    for m in ("map.vlmap", "map.avlmap", "map.vlmap_builder", "robot.habitat_lang_robot"):
        importlib.import_module(f"{name}.{m}")


def test_install_repoints_every_reference_and_uninstall_restores(tmp_path):
    from avlmaps_amd import compat
    from avlmaps_amd.utils import clip_utils, index_utils, visualize_utils
    name = "fake_avlmaps_pkg"
    _fake_upstream(tmp_path, name)
    try:
        counts = compat.install(name)
        vl = sys.modules[f"{name}.map.vlmap"]
        av = sys.modules[f"{name}.map.avlmap"]
        cu = sys.modules[f"{name}.utils.clip_utils"]
        iu = sys.modules[f"{name}.utils.index_utils"]
        vu = sys.modules[f"{name}.utils.visualize_utils"]
        vb = sys.modules[f"{name}.map.vlmap_builder"]
        assert cu.get_lseg_score is clip_utils.get_lseg_score and vl.get_lseg_score is clip_utils.get_lseg_score
        assert iu.get_lseg_score is clip_utils.get_lseg_score
        assert vl.get_dynamic_obstacles_map_3d is index_utils.get_dynamic_obstacles_map_3d
        assert av.get_heatmap_from_mask_3d is visualize_utils.get_heatmap_from_mask_3d
        assert vu.pool_3d_label_to_2d is visualize_utils.pool_3d_label_to_2d
        assert vb.VLMapBuilder.create_mobile_base_map.__doc__.startswith("avlmaps.map.vlmap_builder.VLMapBuilder.create_mobile_base_map")
        assert counts["utils.clip_utils.get_lseg_score"] == 2 and compat.install(name) == {}
        # the navigator's inline decay loop (habitat_lang_robot.py:242-265) is re-pointed when its module is loaded
        rb = sys.modules[f"{name}.robot.habitat_lang_robot"]
        assert counts["robot.habitat_lang_robot.HabitatLanguageRobot.get_vl_distribution_map_3d"] == 1
        assert "heat kernels" in rb.HabitatLanguageRobot.get_vl_distribution_map_3d.__doc__
        compat.uninstall(name)
        assert rb.HabitatLanguageRobot().get_vl_distribution_map_3d("x") == "upstream-decay-loop"
        assert cu.get_lseg_score() == "upstream-score" and vl.get_lseg_score() == "upstream-score"
        assert iu.get_lseg_score() == "upstream-score-dup" and av.get_heatmap_from_mask_3d() == "upstream-heat"
        assert vb.VLMapBuilder().create_mobile_base_map() == "upstream-build"
    finally:
        compat.uninstall(name)
        sys.path.remove(str(tmp_path))
        for m in [m for m in sys.modules if m == name or m.startswith(name + ".")]:
            del sys.modules[m]


@pytest.mark.skipif(not os.path.isdir(os.environ.get("AVLMAPS_REFERENCE", "/root/reference")),
                    reason="the upstream checkout is only present in the build container")
def test_install_on_the_real_upstream_package():
    """the real avlmaps modules (imported with their heavy dependencies mocked): the names the patch targets exist there
    with the call signatures the replacements accept"""
    import inspect
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))
    from ref_import import import_reference
    from avlmaps_amd import compat
    from avlmaps_amd.utils import clip_utils
    m = import_reference()
    ref_sig = inspect.signature(m["clip_utils"].get_lseg_score)
    try:
        counts = compat.install("avlmaps")
        assert m["clip_utils"].get_lseg_score is clip_utils.get_lseg_score
        assert m["vlmap"].get_lseg_score is clip_utils.get_lseg_score                 # vlmap.py:24 imported the name
        assert counts["utils.clip_utils.get_lseg_score"] >= 2
        assert "map.vlmap_builder.VLMapBuilder.create_mobile_base_map" in counts
        assert "utils.visualize_utils.get_heatmap_from_mask_3d" in counts
        mine = inspect.signature(clip_utils.get_lseg_score)
        assert list(mine.parameters)[:len(ref_sig.parameters)] == list(ref_sig.parameters)   # same positional order
        for k, p in ref_sig.parameters.items():
            assert mine.parameters[k].default == p.default, k
    finally:
        compat.uninstall("avlmaps")
    assert m["clip_utils"].get_lseg_score is not clip_utils.get_lseg_score
    # leave no mocked third-party module (h5py, cv2, ...) or upstream module behind for the other tests
    from unittest.mock import MagicMock
    for name in [n for n, mod in sys.modules.items() if isinstance(mod, MagicMock) or n == "avlmaps" or n.startswith("avlmaps.")]:
        del sys.modules[name]
