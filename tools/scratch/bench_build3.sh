#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
timeout -s KILL 300 python -m pytest tests/test_builder_gpu.py -q -m gpu -x 2>&1 | tail -3
for f in "" "--deferred-fuse" "--build-batch 16" "--build-batch 64"; do
 timeout -s KILL 120 python bench.py --workload build --steps 6000 --no-cpu $f 2>&1 | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read()); e = r['extra']
print('$f', e['deferred_fuse'], round(r['value']), round(e['ms_per_frame_fuse']*1e3,2), 'us/frame', e['voxels_local'], round(r['roofline']['frac'],3))"
done
