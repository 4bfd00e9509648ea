#!/bin/bash
# round 6: same-box A/B of the replay-log record store (stock = only active samples; uncond = every sample; nologrec = no record at all)
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for lib in avlmaps_amd/lib/libavlmaps_hip.so variants/libavlmaps_hip_loguncond.so variants/libavlmaps_hip_nologrec.so; do
 AVLMAPS_HIP_LIB=$lib timeout -s KILL 120 python bench.py --workload build --steps 10000 --no-cpu --deferred-fuse 2>/dev/null | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read()); e = r['extra']
print('$lib', round(r['value']), round(e['ms_per_frame_fuse']*1e3,2), 'us/frame')"
done; done
