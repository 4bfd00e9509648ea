#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
for rep in 1 2; do
for lib in "" "variants/libavlmaps_hip_prevk3.so"; do
for f in "" "--deferred-fuse" "--build-batch 16" "--build-batch 64"; do
 AVLMAPS_HIP_LIB=${lib:-avlmaps_amd/lib/libavlmaps_hip.so} timeout -s KILL 120 python bench.py --workload build --steps 10000 --no-cpu $f 2>&1 | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read()); e = r['extra']
print('${lib:-new}', '$f', e['deferred_fuse'], round(r['value']), round(e['ms_per_frame_fuse']*1e3,2), 'us/frame', e['voxels_local'], round(r['roofline']['frac'],3))"
done; done; done
