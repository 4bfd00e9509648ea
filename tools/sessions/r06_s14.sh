#!/bin/bash
# round 6 (final code): eight-rank rehearsal, five launches -> profiles/r06_merge_rehearsal_8ranks.json; single-GPU build lines
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_s14; mkdir -p $O; rm -f $O/rehearsal.json
for k in 1 2 3 4 5; do
AVLMAPS_DIST_BACKEND=gloo AVLMAPS_SHARED_GPU_LOCK=/tmp/avl_gpu.lock timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 \
    --master-addr 127.0.0.1 --master-port 2973$k bench.py --gpus 8 --workload build --steps 10000 --warmup 8 --no-cpu --trajectory spiral --spiral-radius 4 > $O/r8_$k.log 2> $O/r8_$k.err
python tools/summarize_merge.py $O/r8_$k.log --json=$O/rehearsal.json 2>&1 | sed -n 4p | cut -c1-120
done
for mode in "--deferred-fuse" ""; do
timeout 600 python bench.py --workload build --steps 10000 --warmup 8 --no-cpu $mode > $O/b1.log 2> $O/b1.err
python - <<'PY'
import json
for l in open("gpurun_out/r06_s14/b1.log"):
    if l.startswith("{"):
        d = json.loads(l); e = d["extra"]; s = e["single_gpu_merge_path"]
        print("deferred", e["deferred_fuse"], "frames/s", round(e["frames_per_s"]), "merge+fin", round(e["merge_finalize_seconds"], 5), {k: s.get(k) for k in ("compute_total_s", "plain_finalize_s", "merge_cold_s")})
PY
done
