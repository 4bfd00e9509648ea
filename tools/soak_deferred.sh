#!/bin/bash
# GPU box: bounded soak of the deferred-fuse build (3 000 frames per run = 3 000 pipe_kernel launches, each with samples polling
# for cells under creation), plain and under rocprofv3; every run under its own timeout, so a stall shows up as "bad", not as a hung box.
# SOAK_RUNS / SOAK_PROF_RUNS set the repetitions (200 / 20 were run for profiles/HISTORY.md 4.3).
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ok=0; bad=0
for i in $(seq 1 ${SOAK_RUNS:-60}); do
  timeout -s KILL 40 python $R/bench.py --workload build --steps 3000 --no-cpu --deferred-fuse > /tmp/o.txt 2>&1
  rc=$?
  if [ $rc -eq 0 ]; then ok=$((ok+1)); else bad=$((bad+1)); echo "plain run $i rc=$rc"; tail -3 /tmp/o.txt; fi
done
echo "plain: ok=$ok bad=$bad"
ok=0; bad=0
for i in $(seq 1 ${SOAK_PROF_RUNS:-10}); do
  rm -rf /tmp/prof
  timeout -s KILL 60 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o p -- python $R/bench.py --workload build --steps 3000 --no-cpu --deferred-fuse > /tmp/o.txt 2>&1
  rc=$?
  if [ $rc -eq 0 ]; then ok=$((ok+1)); else bad=$((bad+1)); echo "rocprof run $i rc=$rc"; tail -5 /tmp/o.txt; fi
done
echo "rocprof: ok=$ok bad=$bad"
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1); echo $f
python - <<PY
import csv,glob
f=glob.glob('/tmp/prof/**/*kernel_stats.csv', recursive=True)
if f:
    for r in list(csv.DictReader(open(f[0])))[:6]:
        print(r['Name'][:60], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
PY
