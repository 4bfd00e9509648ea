#!/bin/bash
# GPU box: same-box A/B of the stock library against a variant (AB_VARIANT=variants/lib....so, tools/build_variant.py --src avl_builder.hip)
# on the build workload, alternating; box-to-box spread is several per cent, so only same-box pairs mean anything.
R=$GRAFT_REPO_ROOT
cd $R
for rep in 1 2; do
for lib in "" "${AB_VARIANT:-variants/libavlmaps_hip_prev.so}"; do
for f in "" "--deferred-fuse" "--build-batch 16" "--build-batch 64"; do
 AVLMAPS_HIP_LIB=${lib:-avlmaps_amd/lib/libavlmaps_hip.so} timeout -s KILL 120 python bench.py --workload build --steps 10000 --no-cpu $f 2>&1 | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read()); e = r['extra']
print('${lib:-new}', '$f', e['deferred_fuse'], round(r['value']), round(e['ms_per_frame_fuse']*1e3,2), 'us/frame', e['voxels_local'], round(r['roofline']['frac'],3))"
done; done; done
