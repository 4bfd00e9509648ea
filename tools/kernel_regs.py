"""Register / spill report of every kernel in a .hip source, from the gfx950 assembly's metadata.

    python tools/kernel_regs.py avlmaps_amd/csrc/avl_sim.hip [-DAVL_X ...]

Used by tests/test_no_spills.py: no kernel the dispatcher can select may spill."""
from __future__ import annotations

import re
import shutil
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def _demangle(names):
    filt = shutil.which("c++filt") or shutil.which("llvm-cxxfilt")
    if not filt:
        return list(names)
    out = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return [re.sub(r"\(.*", "", o).replace("void ", "").replace("avl::", "") for o in out]


def kernel_regs(src: Path, extra_flags=()):
    from avlmaps_amd import build as b
    flags = [*b.COMMON, *b.SOURCES.get(Path(src).name, []), *extra_flags]
    with tempfile.TemporaryDirectory() as td:
        out = Path(td) / "k.s"
        cmd = [b._hipcc(), *flags, "-S", "--cuda-device-only", "-o", str(out), str(src)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stderr)
        text = out.read_text()
    rows = []
    for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", text, re.S):
        blk = m.group(0)
        g = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", blk).group(1))
        rows.append(dict(mangled=re.search(r"\.name:\s+(\S+)", blk).group(1), vgpr=g("vgpr_count"), agpr=g("agpr_count"),
                         sgpr=g("sgpr_count"), spill=g("vgpr_spill_count"), sgpr_spill=g("sgpr_spill_count"),
                         scratch=g("private_segment_fixed_size"), lds=g("group_segment_fixed_size")))
    for row, name in zip(rows, _demangle([r["mangled"] for r in rows])):
        row["name"] = name
    return rows


if __name__ == "__main__":
    src = Path(sys.argv[1])
    for r in kernel_regs(src, sys.argv[2:]):
        print(f"{r['name']:78s} vgpr={r['vgpr']:4d} agpr={r['agpr']:3d} spill={r['spill']:3d} scratch={r['scratch']:4d}")
