#!/bin/bash
# round 6: gather-plan merge -- single-GPU merge path + eight-rank rehearsal (exploration radius 4)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_s1; mkdir -p $O
timeout 600 python bench.py --workload build --steps 10000 --warmup 8 --no-cpu --deferred-fuse > $O/b1.log 2> $O/b1.err; tail -c 300 $O/b1.err
python - <<'PY'
import json
for l in open("gpurun_out/r06_s1/b1.log"):
    if l.startswith("{"):
        d = json.loads(l)
        e = d["extra"]
        print("frames/s", round(e["frames_per_s"]), "fuse", e["fuse_seconds_max_rank"], "merge+fin", e["merge_finalize_seconds"])
        s = e["single_gpu_merge_path"]
        print({k: s.get(k) for k in ("compute_s", "compute_total_s", "plain_finalize_s", "merged_voxels", "plan")})
PY
AVLMAPS_DIST_BACKEND=gloo AVLMAPS_MERGE_TRACE=1 AVLMAPS_SHARED_GPU_LOCK=/tmp/avl_gpu.lock timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 \
    --master-addr 127.0.0.1 --master-port 29713 bench.py --gpus 8 --workload build --steps 10000 --warmup 8 --no-cpu --trajectory spiral --spiral-radius 4 > $O/r8.log 2> $O/r8.err
python tools/summarize_merge.py $O/r8.log 2>&1 | head -6 | cut -c1-700
tail -5 $O/r8.err | cut -c1-400
