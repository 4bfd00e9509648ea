#!/bin/bash
# round 5: where the single-GPU merge stand-in (M = 2.25 M) spends its device time: torch profiler table of one merge
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_s31; mkdir -p $O
AVLMAPS_MERGE_PROFILE=0 timeout 600 python bench.py --workload build --steps 10000 --no-cpu > $O/b.log 2> $O/prof.txt
grep '^{"metric"' $O/b.log | python -c "
import sys,json
j=json.loads(sys.stdin.read()); s=j['extra']['single_gpu_merge_path']; print({k:round(1e3*v,2) for k,v in s['wall_s'].items()}, 'total', round(1e3*s['compute_total_s'],2))"
grep -n "Self CUDA\|Name" $O/prof.txt | head
