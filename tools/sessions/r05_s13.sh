#!/bin/bash
# round 5: where the directory plan's time goes at n = 2.25 M (torch op breakdown)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_s13; mkdir -p $O
timeout 300 python tools/probe_merge_plan.py 2250000 > $O/merge_plan_ops.txt 2>&1
head -60 $O/merge_plan_ops.txt
