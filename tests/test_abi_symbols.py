"""CPU-side checks of the C-ABI library: it builds, loads, and exports every symbol the header declares."""
import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def lib_path():
    from avlmaps_amd.build import build
    return build()


def _declared():
    text = (ROOT / "include" / "avlmaps_hip.h").read_text()
    return sorted(set(re.findall(r"AVL_API\s+[\w\s\*]+?\b(avl_\w+)\s*\(", text)))


def test_header_declares_symbols():
    names = _declared()
    assert "avl_sim_scores" in names and "avl_builder_integrate_frame" in names and len(names) >= 30


def test_library_exports_every_declared_symbol(lib_path):
    lib = ctypes.CDLL(str(lib_path))
    missing = [n for n in _declared() if not hasattr(lib, n)]
    assert not missing, missing


def test_ctypes_table_matches_header(lib_path):
    from avlmaps_amd import _lib
    assert sorted(_lib.EXPORTED_SYMBOLS) == _declared()
    _lib.load()


def test_error_string_and_version(lib_path):
    from avlmaps_amd import _lib
    lib = _lib.load()
    assert lib.avl_version() >= 100
    # argument validation happens before any device work, so it is testable without a GPU
    rc = lib.avl_sim_workspace_bytes(512, 64, None)
    assert rc != 0 and b"null" in lib.avl_last_error()


def test_no_cpu_fallback_without_gpu(lib_path):
    from avlmaps_amd import _lib, ops
    import numpy as np
    if _lib.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(_lib.AvlError):
        ops.sim_scores(np.zeros((4, 512), np.float32), np.zeros((2, 512), np.float32))


def test_product_never_imports_oracle():
    for p in (ROOT / "avlmaps_amd").rglob("*.py"):
        src = p.read_text()
        assert "import oracle" not in src and "from oracle" not in src, p


def test_header_is_plain_c_and_the_c_caller_links(tmp_path):
    """include/avlmaps_hip.h compiles as C (gcc, -std=c99 -pedantic) and examples/c_caller.c links against the library:
    the boundary really is a C ABI, usable without Python or torch"""
    import shutil
    import subprocess
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    gcc = shutil.which("gcc")
    if gcc is None:
        import pytest
        pytest.skip("no gcc")
    probe = tmp_path / "probe.c"
    probe.write_text('#include "avlmaps_hip.h"\nint main(void) { return avl_version() < 0; }\n')
    r = subprocess.run([gcc, "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", str(root / "include"), "-c", str(probe), "-o",
                        str(tmp_path / "probe.o")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lib = root / "avlmaps_amd" / "lib"
    r = subprocess.run([gcc, "-O1", "-I", str(root / "include"), str(root / "examples" / "c_caller.c"), "-L", str(lib), "-lavlmaps_hip",
                        f"-Wl,-rpath,{lib}", "-lm", "-o", str(tmp_path / "c_caller")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
