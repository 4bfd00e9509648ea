#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s55
timeout 600 python -m pytest tests/test_sim_gpu.py -q -m gpu -x 2>&1 | tail -3 > gpurun_out/s55/tests.txt
timeout 1500 python tools/ab_sim.py --reps 3 --shapes 2000000x512x64 --modes raw,prepared stock nocarry > gpurun_out/s55/ab.txt 2>&1
