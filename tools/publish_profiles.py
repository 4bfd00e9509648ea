#!/usr/bin/env python3
"""Copy the judged summaries from gpurun_out/prof_<tag>/ into profiles/<tag>_* (trimmed to this library's kernels) and
refresh profiles/pmc_traffic.json: one entry per (workload, shape) with the HBM bytes per step of the dominant kernels
(gfx950 FETCH_SIZE correction applied) and, where the SQ pass exists, the matrix-pipe busy fraction.  bench.py only reports
an entry whose shape matches its own run."""
import csv
import json
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
src, dst = ROOT / "gpurun_out" / f"prof_{tag}", ROOT / "profiles"
dst.mkdir(exist_ok=True)
ours = lambda name: "avl" in name or "rocclr" in name or "rocprim" in name or "ccl" in name.lower()
for f in sorted(src.glob("*_kernel_stats.csv")):
    rows = list(csv.reader(open(f)))
    keep = [rows[0]] + [r for r in rows[1:] if ours(r[0])]
    other = [r for r in rows[1:] if not ours(r[0])]
    keep.append([f"(torch input-generation / bookkeeping kernels: {len(other)} names summed)", sum(int(r[1]) for r in other),
                 sum(int(r[2]) for r in other), "", "", "", "", ""])
    csv.writer(open(dst / f"{tag}_{f.name}", "w")).writerows(keep)
pmc = {}
for f in sorted(src.glob("pmc_*.json")):
    d = {k: v for k, v in json.load(open(f)).items() if "avl" in k}
    json.dump(d, open(dst / f"{tag}_{f.name}", "w"), indent=1)
    pmc[f.stem] = d
if (src / "power_probe.txt").exists():
    shutil.copy(src / "power_probe.txt", dst / f"{tag}_power_probe.txt")
for f in sorted(src.glob("build_8ranks_one_gpu_*_summary.txt")) + sorted(src.glob("build_8ranks_one_gpu_*_trace.txt")) + sorted(src.glob("pipeline_probe.txt")):
    shutil.copy(f, dst / f"{tag}_{f.name}")
lines = {}
for f in sorted(src.glob("*.log")):
    if f.name.startswith("pmc_"):
        continue
    ls = [l for l in open(f, errors="replace") if l.startswith("{")]
    if ls:
        open(dst / f"{tag}_{f.stem}.json", "w").writelines(ls)
        lines[f.stem] = json.loads(ls[-1])
for name in ("rccl_two_ranks_one_gpu.log",):
    if (src / name).exists():
        keep = [l for l in open(src / name, errors="replace") if "Duplicate GPU" in l or "NCCL version" in l or "RCCL version" in l]
        open(dst / f"{tag}_{name.replace('.log', '.txt')}", "w").writelines(keep[:6])


def pretty(name):
    """_ZN3avl20sim_split_f16_kernelILi2ELi8ELb0ELb1ELb0EEEv... -> sim_split_f16_kernel<2,8,0,1,0>; plain names pass through"""
    import re
    m = re.match(r"_ZN3avl\d+([A-Za-z0-9_]+?)I((?:L[ib]\d+E)+)E", name)
    if m:
        return m.group(1) + "<" + ",".join(re.findall(r"L[ib](\d+)E", m.group(2))) + ">"
    return name.split("(")[0].replace("avl::", "").replace("void ", "")


def per_step(d, match, counter, how="mean"):
    """sum over the kernels of one step (each is launched once per step) of a counter's per-launch mean"""
    tot, names = 0.0, []
    for k, v in d.items():
        if any(m in k for m in match) and "prepare_map" not in k and counter in v:      # the one-off conversion is not part of a step
            tot += v[counter][how]
            names.append(pretty(k))
    return tot, names


def busy(d, match):
    out = {}
    for k, v in d.items():
        if any(m in k for m in match) and "SQ_VALU_MFMA_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v and v["GRBM_GUI_ACTIVE"]["mean"] > 0:
            short = "split" if "sim_split" in k else ("stream" if "sim_stream" in k else ("kswap" if "sim_kswap" in k else k[:40]))
            out[short] = v["SQ_VALU_MFMA_BUSY_CYCLES"]["mean"] / (1024.0 * v["GRBM_GUI_ACTIVE"]["mean"] / 8.0)
    return out


# FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports half of a wide coalesced read stream
# (/opt/skills/guides/MI355X_MICROARCH.md, HBM section) -> doubled.  WRITE_SIZE is uncalibrated (atomics inflate it).
entries = []
SIM = ("sim_split_f16_kernel", "sim_kswap_f16_kernel", "sim_stream_f16_kernel", "sim_stream_tb_f16_kernel", "sim_fixup_rows_kernel", "sim_gather_queries_kernel", "sim_prep_queries_kernel")
for stem, shape, line in (("pmc_index", dict(N=2000000, D=512, Q=64, resident="raw"), "index_bench"),
                          ("pmc_config5", dict(N=2000000, D=1536, Q=128, resident="raw"), "config5_bench"),
                          ("pmc_index_compact", dict(N=2000000, D=512, Q=64, resident="compact"), "index_compact_bench"),
                          ("pmc_config5_compact", dict(N=2000000, D=1536, Q=128, resident="compact"), "config5_compact_bench")):
    if stem not in pmc:
        continue
    f, names = per_step(pmc[stem], SIM, "FETCH_SIZE")
    w, _ = per_step(pmc[stem], SIM, "WRITE_SIZE")
    if f:
        b = busy(pmc[stem], ("sim_split_f16_kernel", "sim_kswap_f16_kernel", "sim_stream_f16_kernel", "sim_stream_tb_f16_kernel"))
        entries.append(dict(workload="index", shape=shape, kernel=" + ".join(sorted(set(names)))[:300],
                            read_bytes=2 * f * 1024, write_bytes=w * 1024, total_bytes=2 * f * 1024 + w * 1024,
                            mfma_busy_frac=(list(b.values())[0] if len(b) == 1 else (b or None)),
                            source=f"profiles/{tag}_{stem}.json", note="sum over the kernels of one step; FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, KB -> B"))
BUILD = ("voxelize_link_kernel", "pipe_kernel", "fuse_kernel", "fuse_generic_kernel")
for stem, fpl in (("pmc_build", 1), ("pmc_build_b16", 16)):
    if stem not in pmc:
        continue
    f, names = per_step(pmc[stem], BUILD, "FETCH_SIZE")
    w, _ = per_step(pmc[stem], BUILD, "WRITE_SIZE")
    if f:
        entries.append(dict(workload="build", shape=dict(frames_per_launch=fpl), kernel="voxelize_link + fuse (one launch pair)",
                            read_bytes=2 * f * 1024, write_bytes=w * 1024, total_bytes=2 * f * 1024 + w * 1024, mfma_busy_frac=None,
                            source=f"profiles/{tag}_{stem}.json",
                            note="FETCH_SIZE x2 is calibrated for wide coalesced reads only; gathers may differ; mean over the launches of the run"))
if entries:
    json.dump(dict(entries=entries), open(dst / "pmc_traffic.json", "w"), indent=1)
print(json.dumps(entries, indent=1))
for k, d in lines.items():
    print(k, d.get("metric"), d.get("value"), (d.get("roofline") or {}).get("frac"))
