cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r04; mkdir -p $O
name=spiral4; extra="--trajectory spiral --spiral-radius 4"
(timeout 600 python bench.py --workload build --steps 10000 --no-cpu $extra) > $O/build_10k_${name}_1rank.log 2>&1
AVLMAPS_DIST_BACKEND=gloo AVLMAPS_MERGE_TRACE=1 AVLMAPS_SHARED_GPU_LOCK=/tmp/avl_gpu.lock timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 \
    --master-addr 127.0.0.1 --master-port 29713 bench.py --gpus 8 --workload build --steps 10000 --warmup 8 --no-cpu $extra > $O/build_8ranks_one_gpu_${name}.log 2> $O/build_8ranks_one_gpu_${name}.err
grep "merge trace" $O/build_8ranks_one_gpu_${name}.err > $O/build_8ranks_one_gpu_${name}_trace.txt
python tools/summarize_merge.py $O/build_8ranks_one_gpu_${name}.log | tee $O/build_8ranks_one_gpu_${name}_summary.txt
timeout 900 python -m pytest tests/test_api_gpu.py -m gpu -x -q 2>&1 | tail -3
python - <<'PY'
import time, sys
sys.path.insert(0, '.')
import torch, numpy as np
from avlmaps_amd import ops
torch.zeros(1).cuda(); torch.cuda.synchronize()
for cap in (1<<18, 1<<20):
    t=time.perf_counter(); a=ops.VoxelAccumulator(1000,0.05,30,512,capacity=cap); torch.cuda.synchronize(); t1=time.perf_counter()
    a.enable_replay_log(600*7776); torch.cuda.synchronize(); t2=time.perf_counter(); a.close()
    print(f"create cap {cap}: {1e3*(t1-t):.1f} ms, replay log {1e3*(t2-t1):.1f} ms")
PY
timeout 300 python tools/probe_pipeline.py 600 2>&1 | grep -v "Temporarily\|save:\|amdgpu" | tail -10
