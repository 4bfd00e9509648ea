#!/bin/bash
# round 6: eight-rank rehearsal, three launches (-> profiles/r06_merge_rehearsal_8ranks.json: per-rank minimum over the launches)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_s7; mkdir -p $O; rm -f $O/rehearsal.json
for k in 1 2 3; do
AVLMAPS_MERGE_TRACE=1 AVLMAPS_DIST_BACKEND=gloo AVLMAPS_SHARED_GPU_LOCK=/tmp/avl_gpu.lock timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 \
    --master-addr 127.0.0.1 --master-port 2971$k bench.py --gpus 8 --workload build --steps 10000 --warmup 8 --no-cpu --trajectory spiral --spiral-radius 4 > $O/r8_$k.log 2> $O/r8_$k.err
python tools/summarize_merge.py $O/r8_$k.log --json=$O/rehearsal.json 2>&1 | sed -n 4p | cut -c1-200
done
grep "merge2 trace" $O/r8_3.err | tail -8 | cut -c1-330
