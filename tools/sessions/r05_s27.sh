#!/bin/bash
# round 5: K3 over owners compacted by K2 (half the workgroups, all resident) -- same-box A/B against the tree before (variants/nocompact): tests, kernel averages, end to end
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_s27; mkdir -p $O
timeout 900 python -m pytest tests/test_builder_gpu.py tests/test_geometry_gpu.py -m gpu -x -q > $O/pytest_builder.txt 2>&1; tail -3 $O/pytest_builder.txt
export TMPDIR=/tmp
for rep in 1 2; do
for lib in variants/libavlmaps_hip_nocompact.so avlmaps_amd/lib/libavlmaps_hip.so; do
for f in "--deferred-fuse" "" "--build-batch 16" "--build-batch 64"; do
 AVLMAPS_HIP_LIB=$PWD/$lib timeout -s KILL 120 python bench.py --workload build --steps 10000 --no-cpu $f > /tmp/o.txt 2>&1
 grep '^{"metric"' /tmp/o.txt | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print('$lib'.split('_')[-1], '[$f]', 'frames/s %.0f  us/frame %.2f' % (j['value'], 1e3*j['ms_per_step']))
" >> $O/ab.txt 2>&1
 if [ $rep = 1 ]; then
 rm -rf /tmp/prof
 (cd /tmp; AVLMAPS_HIP_LIB=$GRAFT_REPO_ROOT/$lib timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o p -- python $GRAFT_REPO_ROOT/bench.py --workload build --steps 4000 --no-cpu $f > /tmp/o2.txt 2>&1)
 python - "$lib $f" <<PY >> $O/kernels.txt
import csv,glob,sys
f=glob.glob('/tmp/prof/**/*kernel_stats.csv', recursive=True)
out=[]
for r in csv.DictReader(open(f[0])):
    n=r['Name']
    for k in ('pipe_kernel','fuse_kernel','voxelize_link_kernel'):
        if k in n: out.append(f"{k}:{float(r['AverageNs'])/1e3:.2f}us x{r['Calls']}")
print(sys.argv[1], ' '.join(out))
PY
 fi
done; done; done
cat $O/ab.txt $O/kernels.txt
