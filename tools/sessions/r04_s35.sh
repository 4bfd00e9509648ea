#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s35
timeout 900 python -m pytest tests/test_builder_gpu.py -q -m gpu -k "heat" -x 2>&1 | tail -5 > gpurun_out/s35/tests.txt
timeout 600 python tools/probe_pipeline.py 2000 > gpurun_out/s35/pipeline2000.txt 2>&1
