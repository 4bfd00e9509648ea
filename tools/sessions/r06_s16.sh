#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_s16; mkdir -p $O
timeout 900 python -m pytest tests/test_merge2_gpu.py -x -q > $O/t_merge2.log 2>&1; tail -5 $O/t_merge2.log
timeout 400 python tools/fuzz_merge2.py 200 11 > $O/fuzz.log 2>&1; tail -8 $O/fuzz.log | cut -c1-400
