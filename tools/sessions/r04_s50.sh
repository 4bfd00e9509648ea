#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s50
timeout 1200 python -m pytest tests/test_sim_gpu.py -q -m gpu 2>&1 | tail -12 > gpurun_out/s50/tests.txt
timeout 900 python -m pytest tests/test_api_gpu.py -q -m gpu -k "compact or fused or index" 2>&1 | tail -5 >> gpurun_out/s50/tests.txt
timeout 1200 python tools/ab_sim.py --reps 2 --shapes 2000000x512x64,2000000x512x65,2000000x1024x64 --modes compact stock > gpurun_out/s50/ab.txt 2>&1
timeout 600 python tools/ab_sim.py --reps 2 --shapes 2000000x1536x128 --modes compactblocks stock >> gpurun_out/s50/ab.txt 2>&1
