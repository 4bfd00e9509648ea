#!/bin/bash
# round 5: consecutive tiles per workgroup for EVERY strided view (spread1) / always (spread2) against 4 KiB multiples only (stock), same box
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_s24; mkdir -p $O
for rep in 1 2; do
for lib in avlmaps_amd/lib/libavlmaps_hip.so variants/libavlmaps_hip_spread1.so variants/libavlmaps_hip_spread2.so; do
 echo "== $lib" >> $O/stride.txt
 AVLMAPS_HIP_LIB=$PWD/$lib timeout 300 python tools/probe_stride.py 2>&1 | grep "row stride   512\|row stride  1536\|row stride   768" >> $O/stride.txt
done; done
cat $O/stride.txt
for lib in avlmaps_amd/lib/libavlmaps_hip.so variants/libavlmaps_hip_spread1.so; do
 AVLMAPS_HIP_LIB=$PWD/$lib timeout 300 python bench.py --feat-dim 1536 --queries 128 --steps 200 --no-pmc --no-cpu 2>/dev/null | grep '^{"metric"' | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print('$lib config5 ms', j['ms_per_step'], j['roofline']['frac'])"
done
