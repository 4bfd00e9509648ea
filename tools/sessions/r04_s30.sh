#!/bin/bash
# heat plan: tests, probe, rocprof split
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s30
for k in uniform clustered; do
  timeout 200 python tools/probe_heat.py $k 30 >> gpurun_out/s30/probe.txt 2>&1
done
HEAT_GEOMETRY=cube timeout 200 python tools/probe_heat.py uniform 30 >> gpurun_out/s30/probe.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for k in uniform clustered; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$k -o heat -- python $GRAFT_REPO_ROOT/tools/probe_heat.py $k 30 > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
for k in uniform clustered; do
  f=$(find /tmp/prof_$k -name "*kernel_stats.csv" | head -1)
  echo "== $k" >> gpurun_out/s30/stats.txt
  head -12 "$f" | cut -d, -f1-6 >> gpurun_out/s30/stats.txt
done
du -sh gpurun_out /tmp/prof_* > gpurun_out/s30/du.txt 2>&1
ls -la gpurun_out >> gpurun_out/s30/du.txt
