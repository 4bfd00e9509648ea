#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s34
grep -m1 "model name" /proc/cpuinfo > gpurun_out/s34/shuffle.txt
for e in "X=1" "AVL_NO_AVX512=1" "AVL_NO_AVX512=1 AVL_NO_AVX2=1"; do echo "== $e" >> gpurun_out/s34/shuffle.txt; env $e timeout 200 python tools/probe_shuffle.py 200 >> gpurun_out/s34/shuffle.txt 2>&1; done
timeout 300 python tools/probe_sampler.py > gpurun_out/s34/sampler.txt 2>&1
timeout 300 python -m pytest tests/test_host_mirror.py -q -k "simd_form or sample_pixels" 2>&1 | tail -3 > gpurun_out/s34/tests.txt
PROBE_CASE=reference,0,100 timeout 300 python tools/probe_pipeline.py 600 > gpurun_out/s34/pipe.txt 2>&1
PROBE_CASE=reference,0,0 timeout 300 python tools/probe_pipeline.py 600 >> gpurun_out/s34/pipe.txt 2>&1
PROBE_CASE=reference,1,100 timeout 300 python tools/probe_pipeline.py 600 >> gpurun_out/s34/pipe.txt 2>&1
PROBE_CASE=uniform,1,100 timeout 300 python tools/probe_pipeline.py 600 >> gpurun_out/s34/pipe.txt 2>&1
