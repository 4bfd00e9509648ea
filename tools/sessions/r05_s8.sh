#!/bin/bash
# round 5: host enqueue time vs device time of the frame loop, previous kernels vs the three-stage pipeline (same box)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_s8; mkdir -p $O
for lib in variants/libavlmaps_hip_prev.so avlmaps_amd/lib/libavlmaps_hip.so; do
 echo "== $lib" >> $O/frame_loop.txt
 AVLMAPS_HIP_LIB=$PWD/$lib timeout 300 python tools/probe_frame_loop.py >> $O/frame_loop.txt 2>&1
done
cat $O/frame_loop.txt
