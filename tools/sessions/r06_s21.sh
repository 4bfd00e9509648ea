#!/bin/bash
# round 6: the first build of a cold process, eight separate processes: is there a seconds-long stall, and in which step?
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_s21; mkdir -p $O
for k in 1 2 3 4 5 6 7 8; do
PROBE_CASE=reference,0,100 timeout 300 python tools/probe_pipeline.py 300 > $O/cold_$k.log 2>&1
grep -h "frames/s\|slow_steps" $O/cold_$k.log | cut -c1-900
done
