#!/bin/bash
# round 6: config 5 on the raw map, column-block visual kernel with a three-buffer ring (VERDICT r5 #5 candidate) -- same-box A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_s10; mkdir -p $O
python tools/ab_sim.py --reps 3 --shapes 2000000x1536x128 --modes rawblocks stock ringqm3 2>&1 | tee $O/ab_ringqm.txt | tail -4
