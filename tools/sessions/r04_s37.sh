#!/bin/bash
# long randomised sweeps after the heat / sampler changes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s37
timeout 1000 python tools/fuzz_parity.py 780 2024 > gpurun_out/s37/fuzz.txt 2>&1
timeout 700 python tools/soak_pipeline.py 500 > gpurun_out/s37/soak.txt 2>&1
