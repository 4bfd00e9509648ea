#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s58
timeout 900 python -m pytest tests/test_sim_gpu.py -q -m gpu -x 2>&1 | tail -3 > gpurun_out/s58/tests.txt
timeout 900 python tools/ab_sim.py --reps 3 --shapes 2000000x768x64 --modes raw,prepared,compact stock env:AVL_SIM_KSWAP=0 > gpurun_out/s58/ab.txt 2>&1
