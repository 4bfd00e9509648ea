#!/bin/bash
# round 6: replay log as 32-byte records -- builder / merge tests, build bench (per-frame kernels must not move), kernel table
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_s6; mkdir -p $O
timeout 1500 python -m pytest tests/test_builder_gpu.py tests/test_merge2_gpu.py tests/test_geometry_gpu.py -m gpu -x -q 2>&1 | tail -4
for mode in "--deferred-fuse" ""; do
timeout 600 python bench.py --workload build --steps 10000 --warmup 8 --no-cpu $mode > $O/b1.log 2> $O/b1.err
python - <<'PY'
import json
for l in open("gpurun_out/r06_s6/b1.log"):
    if l.startswith("{"):
        d = json.loads(l); e = d["extra"]; s = e["single_gpu_merge_path"]
        print("deferred", e["deferred_fuse"], "frames/s", round(e["frames_per_s"]), "us/frame fuse", round(1e3 * e["ms_per_frame_fuse"], 2), {k: s.get(k) for k in ("compute_s", "compute_total_s", "plain_finalize_s", "merge_cold_s")})
PY
done
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bd -- python bench.py --workload build --steps 10000 --warmup 8 --no-cpu --deferred-fuse > /dev/null 2>&1
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r06_s6/prof/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:14]:
        print(f'{r["Name"][:100]:100s} calls {r["Calls"]:>6s} avg_us {float(r["AverageNs"])/1e3:9.1f} total_ms {float(r["TotalDurationNs"])/1e6:9.2f}')
PY
