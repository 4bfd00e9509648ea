#!/bin/bash
# split sampler: host micro-bench + whole pipeline (reference sampling) with and without workers
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s33
nproc > gpurun_out/s33/sampler.txt
timeout 300 python tools/probe_sampler.py >> gpurun_out/s33/sampler.txt 2>&1
PROBE_CASE=reference,0,100 timeout 300 python tools/probe_pipeline.py 600 > gpurun_out/s33/pipe_ref_workers.txt 2>&1
AVL_SAMPLER_WORKERS=0 PROBE_CASE=reference,0,100 timeout 300 python tools/probe_pipeline.py 600 > gpurun_out/s33/pipe_ref_noworkers.txt 2>&1
PROBE_CASE=reference,0,0 timeout 300 python tools/probe_pipeline.py 600 >> gpurun_out/s33/pipe_ref_workers.txt 2>&1
timeout 600 python -m pytest tests/test_api_gpu.py -q -m gpu -x 2>&1 | tail -3 > gpurun_out/s33/tests.txt
