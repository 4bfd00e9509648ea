#!/bin/bash
# round 5: end-to-end frames/s of the builder, same box: previous kernels (variants/prev) vs the three-stage pipeline vs write-through result rows
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_s7; mkdir -p $O
for rep in 1 2; do
for lib in variants/libavlmaps_hip_prev.so avlmaps_amd/lib/libavlmaps_hip.so variants/libavlmaps_hip_sc1.so; do
for f in "--deferred-fuse" "" "--build-batch 16" "--build-batch 64"; do
 AVLMAPS_HIP_LIB=$PWD/$lib timeout -s KILL 120 python bench.py --workload build --steps 10000 --no-cpu $f > /tmp/o.txt 2>&1
 grep '^{"metric"' /tmp/o.txt | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print('$lib'.split('_')[-1], '[$f]', 'frames/s %.0f  us/frame %.2f' % (j['value'], 1e3*j['ms_per_step']), 'roofline frac', j.get('roofline',{}).get('frac'))
" >> $O/ab.txt 2>&1
done; done; done
cat $O/ab.txt
