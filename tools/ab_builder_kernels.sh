#!/bin/bash
# GPU box: same-box A/B of the stock library against a variant (AB_VARIANT=variants/lib....so, tools/build_variant.py --src avl_builder.hip)
# on the build workload, alternating; box-to-box spread is several per cent, so only same-box pairs mean anything.
R=$GRAFT_REPO_ROOT
cd $R; timeout -s KILL 300 python -m pytest tests/test_builder_gpu.py -q -m gpu -x 2>&1 | tail -3
cd /tmp; export TMPDIR=/tmp
for rep in 1 2; do
for lib in avlmaps_amd/lib/libavlmaps_hip.so ${AB_VARIANT:-variants/libavlmaps_hip_prev.so}; do
for f in "" "--deferred-fuse" "--build-batch 16" "--build-batch 64"; do
 rm -rf /tmp/prof
 AVLMAPS_HIP_LIB=$R/$lib timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o p -- python $R/bench.py --workload build --steps 4000 --no-cpu $f > /tmp/o.txt 2>&1
 python - "$lib" "$f" <<PY
import csv,glob,sys
f=glob.glob('/tmp/prof/**/*kernel_stats.csv', recursive=True)
out=[]
for r in csv.DictReader(open(f[0])):
    n=r['Name']
    for k in ('pipe_kernel','fuse_kernel','voxelize_link_kernel','bp_voxelize_kernel',' avl::link_kernel'):
        if k in n: out.append(f"{k}:{float(r['AverageNs'])/1e3:.2f}us x{r['Calls']}")
print(sys.argv[1].split('/')[-1], sys.argv[2], ' '.join(out))
PY
done; done; done
