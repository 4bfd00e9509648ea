#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s61
timeout 1200 python tools/ab_sim.py --reps 3 --shapes 2000000x1536x128 --modes rawblocks stock ring3 > gpurun_out/s61/ab.txt 2>&1
