#!/bin/bash
# round 5: the plan's local steps as fused kernels (csrc/avl_merge.hip) -- equality with the tensor code, N-rank builds, plan timing, eight-rank rehearsal
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_s29; mkdir -p $O
timeout 900 python -m pytest tests/test_merge_kernels_gpu.py tests/test_api_gpu.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
for k in 1 0; do AVLMAPS_MERGE_KERNELS=$k timeout 300 python tools/probe_merge_plan.py 2250000 2>&1 | grep "^iter" | sed "s/^/kernels=$k /"; done
AVLMAPS_DIST_BACKEND=gloo AVLMAPS_MERGE_TRACE=1 AVLMAPS_SHARED_GPU_LOCK=/tmp/avl_gpu.lock timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 \
    --master-addr 127.0.0.1 --master-port 29713 bench.py --gpus 8 --workload build --steps 10000 --warmup 8 --no-cpu --trajectory spiral --spiral-radius 4 > $O/r8.log 2> $O/r8.err
python tools/summarize_merge.py $O/r8.log 2>&1 | head -4 | cut -c1-500
timeout 300 python bench.py --workload build --steps 10000 --no-cpu 2>/dev/null | grep '^{"metric"' | python -c "
import sys,json
j=json.loads(sys.stdin.read()); s=j['extra']['single_gpu_merge_path']; print({k:round(1e3*v,2) for k,v in s['wall_s'].items()}, 'total', round(1e3*s['compute_total_s'],2))"
