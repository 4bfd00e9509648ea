#!/bin/bash
# round 5: full GPU suite, then the whole profile collection of the round (tools/collect_profiles.sh r05)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_r05
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/prof_r05/pytest_gpu.txt 2>&1; tail -3 gpurun_out/prof_r05/pytest_gpu.txt
bash tools/collect_profiles.sh r05 > gpurun_out/prof_r05/collect.log 2>&1
tail -5 gpurun_out/prof_r05/collect.log
