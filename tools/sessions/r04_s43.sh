#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s43
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -4 > gpurun_out/s43/tests.txt
timeout 500 python tools/fuzz_parity.py 400 4242 > gpurun_out/s43/fuzz.txt 2>&1
