#!/bin/bash
# round 5: K3 requests the accumulator row and the owner's feature row together with slot_key / head (-DAVL_K3_SPEC=1), same box
R=$GRAFT_REPO_ROOT
cd $R
AVLMAPS_HIP_LIB=$R/variants/libavlmaps_hip_spec.so timeout -s KILL 600 python -m pytest tests/test_builder_gpu.py tests/test_geometry_gpu.py -q -m gpu -x 2>&1 | tail -1
cd /tmp; export TMPDIR=/tmp
for rep in 1 2; do
for lib in avlmaps_amd/lib/libavlmaps_hip.so variants/libavlmaps_hip_spec.so; do
for f in "" "--deferred-fuse" "--build-batch 16" "--build-batch 64"; do
 rm -rf /tmp/prof
 AVLMAPS_HIP_LIB=$R/$lib timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o p -- python $R/bench.py --workload build --steps 4000 --no-cpu $f > /tmp/o.txt 2>&1
 python - "$lib" "$f" <<PY
import csv,glob,sys,json
f=glob.glob('/tmp/prof/**/*kernel_stats.csv', recursive=True)
out=[]
for r in csv.DictReader(open(f[0])):
    n=r['Name']
    for k in ('pipe_kernel','fuse_kernel','voxelize_link_kernel','voxelize_link_next_kernel'):
        if k in n: out.append(f"{k}:{float(r['AverageNs'])/1e3:.2f}us x{r['Calls']}")
print(sys.argv[1].split('/')[-1], sys.argv[2], ' '.join(out))
PY
done; done; done
