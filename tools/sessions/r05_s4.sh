#!/bin/bash
# round 5: the three-stage builder pipeline -- second pass: is_new from S2, accumulator row requested with the table, member rows requested unconditionally
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_s4; mkdir -p $O
timeout 900 python -m pytest tests/test_builder_gpu.py tests/test_geometry_gpu.py -m gpu -x -q > $O/pytest_builder.txt 2>&1; tail -15 $O/pytest_builder.txt
AVLMAPS_HIP_LIB=$PWD/variants/libavlmaps_hip_probe.so timeout 300 python tools/probe_chain.py 1500 > $O/probe_chain.txt 2>&1
tail -3 $O/probe_chain.txt
export TMPDIR=/tmp; cd /tmp
for f in "" "--deferred-fuse" "--build-batch 16" "--build-batch 64"; do
 rm -rf /tmp/prof
 timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o p -- python $GRAFT_REPO_ROOT/bench.py --workload build --steps 4000 --no-cpu $f > /tmp/o.txt 2>&1
 tail -1 /tmp/o.txt | python -c "
import sys,json
try:
    j=json.loads(sys.stdin.read()); print('frames/s', j['value'], 'ms_per_step', j['ms_per_step'])
except Exception as e: print('no json', e)
" >> $GRAFT_REPO_ROOT/$O/kernels.txt
 python - "$f" <<PY >> $GRAFT_REPO_ROOT/$O/kernels.txt
import csv,glob,sys
f=glob.glob('/tmp/prof/**/*kernel_stats.csv', recursive=True)
out=[]
for r in csv.DictReader(open(f[0])):
    n=r['Name']
    if 'stage_kernel' in n: out.append(f"stage_kernel:{float(r['AverageNs'])/1e3:.2f}us x{r['Calls']}")
print(sys.argv[1], ' '.join(out))
PY
done
cat $GRAFT_REPO_ROOT/$O/kernels.txt
cd $GRAFT_REPO_ROOT
