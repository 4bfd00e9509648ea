#!/bin/bash
# round 5: resident kernel on power-of-two row strides -- consecutive tiles per workgroup there (same-box A/B against the tree before, variants/prevsim)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_s23; mkdir -p $O
timeout 600 python -m pytest tests/test_sim_gpu.py -m gpu -x -q > $O/pytest_sim.txt 2>&1; tail -2 $O/pytest_sim.txt
for rep in 1 2; do
for lib in variants/libavlmaps_hip_prevsim.so avlmaps_amd/lib/libavlmaps_hip.so; do
 echo "== $lib" >> $O/stride.txt
 AVLMAPS_HIP_LIB=$PWD/$lib timeout 300 python tools/probe_stride.py 2>&1 | grep "row stride" >> $O/stride.txt
done; done
cat $O/stride.txt
