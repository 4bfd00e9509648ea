#!/bin/bash
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -s KILL 120 python $R/tools/scratch/probe_deferred.py 2>&1 | tail -4
rm -rf /tmp/prof
timeout -s KILL 90 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o p -- python $R/bench.py --workload build --steps 3000 --no-cpu --deferred-fuse > /tmp/o.txt 2>&1
echo "rocprof rc=$?"; tail -1 /tmp/o.txt | cut -c1-200
find /tmp/prof -type f | head
python - <<PY
import csv,glob
f=glob.glob('/tmp/prof/**/*kernel_stats.csv', recursive=True)
for r in list(csv.DictReader(open(f[0])))[:5] if f else []:
    print(r['Name'][:50], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
PY
