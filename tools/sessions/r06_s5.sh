#!/bin/bash
# round 6: full GPU suite after the replay-scratch / code-object changes, then the cold merge trace
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_s5; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
AVLMAPS_MERGE_TRACE=1 timeout 600 python bench.py --workload build --steps 10000 --warmup 8 --no-cpu --deferred-fuse > $O/b1.log 2> $O/b1.err
grep "merge2 trace" $O/b1.err | cut -c1-500
python - <<'PY'
import json
for l in open("gpurun_out/r06_s5/b1.log"):
    if l.startswith("{"):
        d = json.loads(l); e = d["extra"]; s = e["single_gpu_merge_path"]
        print("frames/s", round(e["frames_per_s"]), {k: s.get(k) for k in ("compute_total_s", "plain_finalize_s", "merge_cold_s")})
        print(json.dumps(d.get("summary"))[:600])
PY
