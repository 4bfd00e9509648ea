#!/bin/bash
# round 6: log segments prepared on the builder's own stream at the start of a merge; finalize shares the cached form -- tests, rehearsal, single-GPU path
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_s13; mkdir -p $O; rm -f $O/rehearsal.json
timeout 1500 python -m pytest tests/test_builder_gpu.py tests/test_merge2_gpu.py tests/test_api_gpu.py -m gpu -x -q 2>&1 | tail -4
for k in 1 2 3; do
AVLMAPS_DIST_BACKEND=gloo AVLMAPS_SHARED_GPU_LOCK=/tmp/avl_gpu.lock timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 \
    --master-addr 127.0.0.1 --master-port 2972$k bench.py --gpus 8 --workload build --steps 10000 --warmup 8 --no-cpu --trajectory spiral --spiral-radius 4 > $O/r8_$k.log 2> $O/r8_$k.err
python tools/summarize_merge.py $O/r8_$k.log --json=$O/rehearsal.json 2>&1 | sed -n 4p | cut -c1-120
done
timeout 600 python bench.py --workload build --steps 10000 --warmup 8 --no-cpu --deferred-fuse > $O/b1.log 2> $O/b1.err
python - <<'PY'
import json
for l in open("gpurun_out/r06_s13/b1.log"):
    if l.startswith("{"):
        d = json.loads(l); e = d["extra"]; s = e["single_gpu_merge_path"]
        print("frames/s", round(e["frames_per_s"]), "merge+fin", e["merge_finalize_seconds"], {k: s.get(k) for k in ("compute_s", "compute_total_s", "plain_finalize_s", "merge_cold_s")})
PY
