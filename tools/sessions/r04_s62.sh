#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s62
timeout 1500 python tools/ab_sim.py --reps 3 --shapes 2000000x512x64 --modes raw,compact stock kt2 kt4 > gpurun_out/s62/ab.txt 2>&1
