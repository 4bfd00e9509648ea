#!/bin/bash
# round 5: S3 wave interleaving; S3 waves per S2 workgroup (128 / 96 / 64); kernel arguments in device memory
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_s5; mkdir -p $O
timeout 900 python -m pytest tests/test_builder_gpu.py tests/test_geometry_gpu.py -m gpu -x -q > $O/pytest_builder.txt 2>&1; tail -3 $O/pytest_builder.txt
AVLMAPS_HIP_LIB=$PWD/variants/libavlmaps_hip_probe.so timeout 300 python tools/probe_chain.py 1500 > $O/probe_chain.txt 2>&1
AVLMAPS_FUSE_WAVES=96 AVLMAPS_HIP_LIB=$PWD/variants/libavlmaps_hip_probe.so timeout 300 python tools/probe_chain.py 1500 > $O/probe_chain_w96.txt 2>&1
HIP_FORCE_DEV_KERNARG=1 AVLMAPS_HIP_LIB=$PWD/variants/libavlmaps_hip_probe.so timeout 300 python tools/probe_chain.py 1500 > $O/probe_chain_devkernarg.txt 2>&1
export TMPDIR=/tmp; cd /tmp
run() {  # run <label> <bench flags...>
 local label=$1; shift
 rm -rf /tmp/prof
 timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o p -- python $GRAFT_REPO_ROOT/bench.py --workload build --steps 4000 --no-cpu "$@" > /tmp/o.txt 2>&1
 python - "$label" <<PY >> $GRAFT_REPO_ROOT/$O/kernels.txt
import csv,glob,sys
f=glob.glob('/tmp/prof/**/*kernel_stats.csv', recursive=True)
out=[]
for r in csv.DictReader(open(f[0])):
    n=r['Name']
    if 'stage_kernel' in n: out.append(f"stage_kernel:{float(r['AverageNs'])/1e3:.2f}us x{r['Calls']}")
print(sys.argv[1], ' '.join(out))
PY
}
run "deferred w128" --deferred-fuse
AVLMAPS_FUSE_WAVES=96 run "deferred w96" --deferred-fuse
AVLMAPS_FUSE_WAVES=64 run "deferred w64" --deferred-fuse
HIP_FORCE_DEV_KERNARG=1 run "deferred w128 devkernarg" --deferred-fuse
HIP_FORCE_DEV_KERNARG=1 AVLMAPS_FUSE_WAVES=96 run "deferred w96 devkernarg" --deferred-fuse
run "frame-at-once w128"
run "batch16 w128" --build-batch 16
run "batch64 w128" --build-batch 64
cat $GRAFT_REPO_ROOT/$O/kernels.txt
python -c "import os; print('HIP_FORCE_DEV_KERNARG in env:', os.environ.get('HIP_FORCE_DEV_KERNARG'))"
