#!/usr/bin/env python3
"""Copy the judged summaries from gpurun_out/prof_<tag>/ into profiles/<tag>_* (trimmed to this library's kernels)
and refresh profiles/pmc_traffic.json (HBM bytes per launch of the dominant kernels, gfx950 FETCH_SIZE correction applied)."""
import csv
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src, dst = ROOT / "gpurun_out" / f"prof_{tag}", ROOT / "profiles"
dst.mkdir(exist_ok=True)
ours = lambda name: "avl" in name or "rocclr" in name or "rocprim" in name
for name in ("index_kernel_stats.csv", "build_kernel_stats.csv", "config5_kernel_stats.csv", "build_b64_kernel_stats.csv"):
    if not (src / name).exists():
        continue
    rows = list(csv.reader(open(src / name)))
    keep = [rows[0]] + [r for r in rows[1:] if ours(r[0])]
    other = [r for r in rows[1:] if not ours(r[0])]
    keep.append([f"(torch input-generation kernels: {len(other)} names summed)", sum(int(r[1]) for r in other),
                 sum(int(r[2]) for r in other), "", "", "", "", ""])
    csv.writer(open(dst / f"{tag}_{name}", "w")).writerows(keep)
pmc = {}
for name in ("pmc_index.json", "pmc_build.json"):
    d = {k: v for k, v in json.load(open(src / name)).items() if "avl" in k}
    json.dump(d, open(dst / f"{tag}_{name}", "w"), indent=1)
    pmc[name] = d
import shutil
if (src / "power_probe.txt").exists():
    shutil.copy(src / "power_probe.txt", dst / f"{tag}_power_probe.txt")
for name in ("index_bench.log", "build_bench.log", "bench_default.log", "config5_bench.log", "build_b64_bench.log", "build_config3.log"):
    if not (src / name).exists():
        continue
    lines = [l for l in open(src / name) if l.startswith("{")]
    open(dst / f"{tag}_{name.replace('.log', '.json')}", "w").writelines(lines)


def kb(d, key, counter, how="mean"):
    for k, v in d.items():
        if key in k and counter in v:
            return v[counter][how]
    return None


# FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports half of a wide coalesced read stream
# (/opt/skills/guides/MI355X_MICROARCH.md, HBM section) -> doubled.  WRITE_SIZE is uncalibrated (atomics inflate it).
traffic = {}
f, w = kb(pmc["pmc_index.json"], "sim_split_f16_kernel", "FETCH_SIZE"), kb(pmc["pmc_index.json"], "sim_split_f16_kernel", "WRITE_SIZE", "min")
if f is not None:
    traffic["index"] = dict(kernel="sim_split_f16_kernel", read_bytes=2 * f * 1024, write_bytes=(w or 0) * 1024,
                            total_bytes=2 * f * 1024 + (w or 0) * 1024, note="FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, KB -> B")
tot_r = tot_w = 0.0
for kern in ("bp_voxelize_kernel", "link_kernel", "fuse_kernel"):
    fr, wr = kb(pmc["pmc_build.json"], kern, "FETCH_SIZE"), kb(pmc["pmc_build.json"], kern, "WRITE_SIZE")
    if fr is not None:
        tot_r += 2 * fr * 1024
        tot_w += (wr or 0) * 1024
if tot_r:
    traffic["build"] = dict(kernels="bp_voxelize + link + fuse (one frame)", read_bytes=tot_r, write_bytes=tot_w,
                            total_bytes=tot_r + tot_w, note="FETCH_SIZE x2 is calibrated for wide coalesced reads only; gathers may differ")
json.dump(traffic, open(dst / "pmc_traffic.json", "w"), indent=1)
print(json.dumps(traffic, indent=1))
