#!/bin/bash
# round 5: finer trace of the owner-side fold with eight ranks on one GPU (exploration, radius 4); builder tests on the tree (D = 768 variant, C frame loop)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_s20; mkdir -p $O
timeout 600 python -m pytest tests/test_builder_gpu.py -m gpu -x -q > $O/pytest_builder.txt 2>&1; tail -2 $O/pytest_builder.txt
AVLMAPS_DIST_BACKEND=gloo AVLMAPS_MERGE_TRACE=1 AVLMAPS_SHARED_GPU_LOCK=/tmp/avl_gpu.lock timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 \
    --master-addr 127.0.0.1 --master-port 29713 bench.py --gpus 8 --workload build --steps 10000 --warmup 8 --no-cpu --trajectory spiral --spiral-radius 4 > $O/r8.log 2> $O/r8.err
grep "merge trace" $O/r8.err | grep " fold \| merge " | tail -16 | cut -c1-600
python tools/summarize_merge.py $O/r8.log 2>&1 | head -5 | cut -c1-500
for f in "--deferred-fuse" "" ; do timeout 120 python bench.py --workload build --steps 10000 --no-cpu $f 2>/dev/null | grep '^{"metric"' | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print('[$f]', 'frames/s %.0f  us/frame %.2f' % (j['value'], 1e3*j['ms_per_step']), j['extra'].get('host_loop'))"; done
