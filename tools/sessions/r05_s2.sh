#!/bin/bash
# round 5: where a frame-by-frame builder launch spends its time (instrumented variant), next to the stock kernels' averages on the same box
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_s2; mkdir -p $O
AVLMAPS_HIP_LIB=$PWD/variants/libavlmaps_hip_probe.so timeout 300 python tools/probe_chain.py 1500 > $O/probe_chain.txt 2>&1
tail -5 $O/probe_chain.txt
export TMPDIR=/tmp; cd /tmp
for f in "" "--deferred-fuse"; do
 rm -rf /tmp/prof
 timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o p -- python $GRAFT_REPO_ROOT/bench.py --workload build --steps 4000 --no-cpu $f > /tmp/o.txt 2>&1
 python - "$f" <<PY >> $GRAFT_REPO_ROOT/$O/stock_kernels.txt
import csv,glob,sys
f=glob.glob('/tmp/prof/**/*kernel_stats.csv', recursive=True)
out=[]
for r in csv.DictReader(open(f[0])):
    n=r['Name']
    for k in ('pipe_kernel','fuse_kernel','voxelize_link_kernel'):
        if k in n: out.append(f"{k}:{float(r['AverageNs'])/1e3:.2f}us x{r['Calls']}")
print(sys.argv[1], ' '.join(out))
PY
done
cat $GRAFT_REPO_ROOT/$O/stock_kernels.txt
