#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s57
timeout 900 python -m pytest tests/test_sim_gpu.py tests/test_api_gpu.py -q -m gpu -x 2>&1 | tail -3 > gpurun_out/s57/tests.txt
timeout 900 python tools/ab_sim.py --reps 2 --shapes 2000000x1024x64,2000000x768x64 --modes raw,compact stock env:AVL_SIM_KSWAP=0 > gpurun_out/s57/ab.txt 2>&1
timeout 300 python tools/fuzz_parity.py 150 31337 > gpurun_out/s57/fuzz.txt 2>&1
