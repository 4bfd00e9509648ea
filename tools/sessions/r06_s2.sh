#!/bin/bash
# round 6: kernel table of the gather-plan merge, eight ranks in one process
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_s2; mkdir -p $O
timeout 600 python -m pytest tests/test_api_gpu.py -m gpu -x -q -k "vlmapbuilder or ring_of or recycling" 2>&1 | tail -5
export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o m2 -- python tools/probe_merge2.py 8 10000 3 > $O/probe.log 2>&1
tail -32 $O/probe.log
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r06_s2/prof/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    print(f)
    for r in rows[:45]:
        print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>6s} avg_us {float(r["AverageNs"])/1e3:9.1f} total_ms {float(r["TotalDurationNs"])/1e6:9.2f}')
PY
