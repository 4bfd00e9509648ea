#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s49
timeout 1500 python tools/ab_sim.py --reps 3 --shapes 2000000x512x64 --modes compact stock planar planar2 > gpurun_out/s49/ab.txt 2>&1
