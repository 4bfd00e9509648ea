#!/bin/bash
# round 5: per-hop probe of the SHIPPED builder kernels (instrumented variant of the final source)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_s22; mkdir -p $O
AVLMAPS_HIP_LIB=$PWD/variants/libavlmaps_hip_probe.so timeout 300 python tools/probe_chain.py 1500 > $O/probe_chain.txt 2>&1
tail -4 $O/probe_chain.txt
timeout 300 python -m pytest tests/test_builder_gpu.py -m gpu -x -q 2>&1 | tail -2
