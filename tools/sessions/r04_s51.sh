#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s51
timeout 1500 python tools/ab_sim.py --reps 3 --shapes 2000000x512x64,2000000x512x65,2000000x1024x64,2000000x512x128 --modes compact stock oldlayout > gpurun_out/s51/ab.txt 2>&1
timeout 900 python tools/ab_sim.py --reps 3 --shapes 2000000x1536x128 --modes compactblocks stock oldlayout >> gpurun_out/s51/ab.txt 2>&1
