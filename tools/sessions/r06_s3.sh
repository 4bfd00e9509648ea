#!/bin/bash
# round 6: gather-plan merge with key-presorted lists -- tests, kernel table (8 ranks in one process), eight-rank gloo rehearsal, single-GPU path
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_s3; mkdir -p $O
timeout 900 python -m pytest tests/test_merge2_gpu.py tests/test_api_gpu.py -m gpu -x -q 2>&1 | tail -5
export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o m2 -- python tools/probe_merge2.py 8 10000 3 > $O/probe.log 2>&1
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r06_s3/prof/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:40]:
        if "pipe_kernel" in r["Name"] or "at::native" in r["Name"]: continue
        print(f'{r["Name"][:100]:100s} calls {r["Calls"]:>6s} avg_us {float(r["AverageNs"])/1e3:9.1f} total_ms {float(r["TotalDurationNs"])/1e6:9.2f}')
PY
AVLMAPS_MERGE_TRACE=1 AVLMAPS_DIST_BACKEND=gloo AVLMAPS_SHARED_GPU_LOCK=/tmp/avl_gpu.lock timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 \
    --master-addr 127.0.0.1 --master-port 29713 bench.py --gpus 8 --workload build --steps 10000 --warmup 8 --no-cpu --trajectory spiral --spiral-radius 4 > $O/r8.log 2> $O/r8.err
python tools/summarize_merge.py $O/r8.log --json=$O/rehearsal.json 2>&1 | head -6 | cut -c1-900; grep "merge2 trace" $O/r8.err | tail -8 | cut -c1-400
timeout 600 python bench.py --workload build --steps 10000 --warmup 8 --no-cpu --deferred-fuse > $O/b1.log 2> $O/b1.err
python - <<'PY'
import json
for l in open("gpurun_out/r06_s3/b1.log"):
    if l.startswith("{"):
        d = json.loads(l)
        e = d["extra"]
        print("frames/s", round(e["frames_per_s"]), "fuse", e["fuse_seconds_max_rank"], "merge+fin", e["merge_finalize_seconds"])
        s = e["single_gpu_merge_path"]
        print({k: s.get(k) for k in ("compute_s", "compute_total_s", "plain_finalize_s", "merged_voxels", "merge_cold_s", "merge_cold_compute_s")})
PY
