#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s44
timeout 600 python -m pytest tests/test_sim_gpu.py -q -m gpu -x 2>&1 | tail -3 > gpurun_out/s44/tests.txt
timeout 1500 python tools/ab_sim.py --reps 3 --shapes 2000000x1024x64 --modes raw,compact env:AVL_SIM_KSWAP=0 stock syncswap > gpurun_out/s44/ab_d1024.txt 2>&1
timeout 1500 python tools/ab_sim.py --reps 3 --shapes 2000000x1536x128 --modes rawblocks,compactblocks env:AVL_SIM_KSWAP=0 stock syncswap > gpurun_out/s44/ab_config5.txt 2>&1
