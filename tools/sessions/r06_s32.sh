#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_s32; mkdir -p $O
timeout 900 python -m pytest tests/test_builder_gpu.py -x -q -k "heavy_collisions" > $O/t.log 2>&1; tail -5 $O/t.log
