#!/bin/bash
# round 5: avl_rows_div_f32 in the fold -- N-rank tests, eight-rank rehearsal (exploration radius 4), D = 768 builder (three-chunk K3)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_s28; mkdir -p $O
timeout 900 python -m pytest tests/test_api_gpu.py tests/test_merge_kernels_gpu.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
AVLMAPS_DIST_BACKEND=gloo AVLMAPS_MERGE_TRACE=1 AVLMAPS_SHARED_GPU_LOCK=/tmp/avl_gpu.lock timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 \
    --master-addr 127.0.0.1 --master-port 29713 bench.py --gpus 8 --workload build --steps 10000 --warmup 8 --no-cpu --trajectory spiral --spiral-radius 4 > $O/r8.log 2> $O/r8.err
python tools/summarize_merge.py $O/r8.log 2>&1 | head -4 | cut -c1-500
grep "merge trace" $O/r8.err | grep " fold " | tail -3 | cut -c1-500
for d in 768 1024; do timeout 300 python tools/probe_build_width.py $d 1 2000 2>&1 | grep "D="; timeout 300 python tools/probe_build_width.py $d 0 2000 2>&1 | grep "D="; done
