"""Builds libavlmaps_hip.so (gfx950) in-tree with hipcc.  `python -m avlmaps_amd.build [--force]`."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIBDIR = PKG / "lib"
LIB = LIBDIR / "libavlmaps_hip.so"
ARCH = "gfx950"

# per-source extra flags; the builder's index math must not be contracted into FMAs implicitly
SOURCES = {
    "avl_api.hip": [],
    "avl_sim.hip": [],
    "avl_builder.hip": ["-ffp-contract=off"],
    "avl_heat.hip": [],
    "avl_map2d.hip": [],
    "avl_lseg.hip": [],
    "avl_merge.hip": [],
    "avl_merge2.hip": [],
}
COMMON = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fvisibility=hidden", "-Wall",
          "-Wno-unused-function", "-munsafe-fp-atomics"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and Path(c).exists():
            return c
    raise RuntimeError("hipcc not found (need ROCm with gfx950 support)")


def _stale(out: Path, deps) -> bool:
    if not out.exists():
        return True
    t = out.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


LAST_BUILD = dict(compiled=[], reused=[], linked=False)      # what the last build() call actually did (checked by the driver's log)


def build(force: bool = False, verbose: bool = False) -> Path:
    force = force or os.environ.get("AVLMAPS_FORCE_BUILD") == "1"
    hipcc = _hipcc()
    LIBDIR.mkdir(exist_ok=True)
    objdir = PKG / "build"
    objdir.mkdir(exist_ok=True)
    headers = list(CSRC.glob("*.h")) + [PKG.parent / "include" / "avlmaps_hip.h"]
    srcs = [s for s in SOURCES if (CSRC / s).exists()]
    jobs = []
    for s in srcs:
        src, obj = CSRC / s, objdir / (Path(s).stem + ".o")
        if force or _stale(obj, [src] + headers):
            jobs.append((s, [hipcc, *COMMON, *SOURCES[s], "-c", str(src), "-o", str(obj)]))

    def run(job):
        name, cmd = job
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {name}:\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr)
        return name

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    LAST_BUILD.update(compiled=[j[0] for j in jobs], reused=[s for s in srcs if s not in {j[0] for j in jobs}], linked=False)
    objs = [objdir / (Path(s).stem + ".o") for s in srcs]
    if force or jobs or _stale(LIB, objs):
        LAST_BUILD["linked"] = True
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", str(LIB), *map(str, objs)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    p = build(force="--force" in sys.argv, verbose=True)
    print("built", p)
