#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s47
timeout 600 python -m pytest tests/test_sim_gpu.py -q -m gpu -x 2>&1 | tail -3 > gpurun_out/s47/tests.txt
timeout 1500 python tools/ab_sim.py --reps 2 --shapes 2000000x1024x32,2000000x768x16 --modes raw,compact env:AVL_SIM_KSWAP=0 stock q1t3 > gpurun_out/s47/ab.txt 2>&1
