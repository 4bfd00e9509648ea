#!/bin/bash
# closing sweeps on the final tree
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s63
timeout 1300 python tools/fuzz_parity.py 1200 777001 > gpurun_out/s63/fuzz.txt 2>&1
timeout 700 python tools/soak_pipeline.py 600 99 > gpurun_out/s63/soak.txt 2>&1
