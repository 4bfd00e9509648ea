#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s52
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -4 > gpurun_out/s52/tests.txt
timeout 500 python tools/fuzz_parity.py 420 9001 > gpurun_out/s52/fuzz.txt 2>&1
timeout 600 python bench.py --resident compact --no-build-extra > gpurun_out/s52/index_compact_line.json 2> gpurun_out/s52/err.txt
