#!/bin/bash
# round 5: torch-op profile of the whole single-GPU merge path (AVLMAPS_MERGE_PROFILE=0)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_s17; mkdir -p $O
AVLMAPS_MERGE_PROFILE=0 AVLMAPS_MERGE_TRACE=1 timeout 600 python bench.py --workload build --steps 10000 --no-cpu > $O/merge_profile.txt 2>&1
grep "merge trace" $O/merge_profile.txt | tail -8
