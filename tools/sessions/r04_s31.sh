#!/bin/bash
# full GPU suite + default bench + short fuzz after the heat plan
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s31
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 > gpurun_out/s31/tests.txt
timeout 900 python bench.py > gpurun_out/s31/bench.json 2> gpurun_out/s31/bench.err
timeout 400 python tools/fuzz_parity.py 180 41 > gpurun_out/s31/fuzz.txt 2>&1
