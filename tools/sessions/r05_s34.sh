#!/bin/bash
# round 5: replay walk, entries in flight 2 / 4 / 8 (same box)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_s34; mkdir -p $O
for rep in 1 2; do
for v in "" ahead8 ahead2; do
  if [ -n "$v" ]; then export AVLMAPS_HIP_LIB=$PWD/variants/libavlmaps_hip_$v.so; else unset AVLMAPS_HIP_LIB; fi
  timeout 600 python bench.py --workload build --steps 10000 --no-cpu 2>/dev/null > $O/b.log
  grep '^{"metric"' $O/b.log | python -c "
import sys,json
j=json.loads(sys.stdin.read()); s=j['extra']['single_gpu_merge_path']; print('${v:-ahead4}', {k:round(1e3*v,2) for k,v in s['wall_s'].items()}, 'total', round(1e3*s['compute_total_s'],2), 'plain finalize', round(1e3*s['plain_finalize_s'],2))"
done; done
