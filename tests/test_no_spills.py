"""No kernel of the library may spill registers to scratch (the column-block variants of the similarity kernel
spilled 17-37 VGPRs unnoticed).  Compiles every .hip source to gfx950 assembly with build.py's flags and reads the
per-kernel metadata (.vgpr_spill_count / .sgpr_spill_count / .private_segment_fixed_size).  CPU-only: hipcc cross-compiles."""
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tools"))
sys.path.insert(0, str(ROOT))

from avlmaps_amd import build as B  # noqa: E402
from kernel_regs import kernel_regs  # noqa: E402


@pytest.fixture(scope="module")
def reports():
    srcs = [B.CSRC / s for s in B.SOURCES]
    with ThreadPoolExecutor(max_workers=len(srcs)) as ex:
        return dict(zip([s.name for s in srcs], ex.map(kernel_regs, srcs)))


def test_no_kernel_spills(reports):
    bad = [(src, r["name"], r["spill"], r["sgpr_spill"]) for src, rows in reports.items() for r in rows if r["spill"]]
    assert not bad, f"kernels with register spills: {bad}"
    # scratch that is not a spill (a dynamically indexed private array) is tolerated in the builder's wide-row kernels and in
    # rocPRIM's sorts, never in the similarity kernels
    bad = [(r["name"], r["scratch"]) for r in reports["avl_sim.hip"] if r["scratch"]]
    assert not bad, f"similarity kernels with scratch: {bad}"


def test_similarity_variants_are_all_there_and_fit(reports):
    """every variant the dispatcher of avl_sim.hip can select (configs 2 and 5: resident / streamed / tile-blocked, raw /
    prepared / compact, dense / column-block) exists and stays inside the 256-register budget of 8 waves per CU"""
    rows = reports["avl_sim.hip"]
    names = [r["mangled"] for r in rows]
    for frag in ("sim_split_f16_kernel", "sim_kswap_f16_kernel", "sim_stream_f16_kernel", "sim_stream_tb_f16_kernel", "sim_mfma_f32_kernel",
                 "sim_fixup_rows_kernel", "sim_prepare_map24_kernel"):
        assert any(frag in n for n in names), frag
    assert sum("sim_split_f16_kernel" in n for n in names) >= 40
    assert sum("sim_kswap_f16_kernel" in n for n in names) == 24      # (two query tiles x {8, 6, run-time steps} + one tile, run-time steps) x {raw, prepared, compact} x {dense, column block}
    for r in rows:
        assert r["vgpr"] + r["agpr"] <= 256, r
