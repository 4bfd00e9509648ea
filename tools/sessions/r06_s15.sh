#!/bin/bash
# round 6: chunked payload exchange -- merge tests, fuzz with random chunk sizes, three rehearsal launches (default chunk rows), one with small chunks
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_s15; mkdir -p $O; rm -f $O/rehearsal.json
timeout 900 python -m pytest tests/test_merge2_gpu.py -x -q > $O/t_merge2.log 2>&1; tail -5 $O/t_merge2.log
timeout 400 python tools/fuzz_merge2.py 240 11 > $O/fuzz.log 2>&1; tail -3 $O/fuzz.log
for k in 1 2 3; do
AVLMAPS_DIST_BACKEND=gloo AVLMAPS_SHARED_GPU_LOCK=/tmp/avl_gpu.lock timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 \
    --master-addr 127.0.0.1 --master-port 2974$k bench.py --gpus 8 --workload build --steps 10000 --warmup 8 --no-cpu --trajectory spiral --spiral-radius 4 > $O/r8_$k.log 2> $O/r8_$k.err
python tools/summarize_merge.py $O/r8_$k.log --json=$O/rehearsal.json 2>&1 | sed -n 1,6p | cut -c1-160
done
AVLMAPS_MERGE_CHUNK_ROWS=8192 AVLMAPS_DIST_BACKEND=gloo AVLMAPS_SHARED_GPU_LOCK=/tmp/avl_gpu.lock timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 \
    --master-addr 127.0.0.1 --master-port 29749 bench.py --gpus 8 --workload build --steps 10000 --warmup 8 --no-cpu --trajectory spiral --spiral-radius 4 > $O/r8_c8k.log 2> $O/r8_c8k.err
python tools/summarize_merge.py $O/r8_c8k.log 2>&1 | sed -n 1,6p | cut -c1-160
grep -o '"exchange_chunks": [0-9]*' $O/r8_1.log $O/r8_c8k.log | head
timeout 600 python bench.py --workload build --steps 10000 --warmup 8 --no-cpu > $O/b1.log 2> $O/b1.err; tail -c 600 $O/b1.log
