#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s56
timeout 1500 python tools/ab_sim.py --reps 3 --shapes 2000000x1536x128 --modes rawblocks,compactblocks stock consec > gpurun_out/s56/ab.txt 2>&1
timeout 900 python tools/ab_sim.py --reps 2 --shapes 2000000x1024x64 --modes raw stock consec >> gpurun_out/s56/ab.txt 2>&1
