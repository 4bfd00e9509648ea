#!/bin/bash
# round 6: where the final save of a pipeline run goes (parts), with the patches written chunk by chunk
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_s20; mkdir -p $O
timeout 600 python -m pytest tests/test_api_gpu.py -x -q -k "checkpoint or save or pipeline or builder" > $O/t_api.log 2>&1; tail -3 $O/t_api.log
timeout 900 python bench.py > $O/bench_default.log 2> $O/bench_default.err; python - <<'PY'
import json
for l in open("gpurun_out/r06_s20/bench_default.log"):
    if l.startswith("{"):
        d = json.loads(l); print(json.dumps(d["summary"]["pipeline_frames_per_s"]), d["summary"]["projected_speedup_8gpu"]["value"])
        for k, v in d["extra"]["vlmapbuilder_pipeline"].items():
            if isinstance(v, dict): print(k, v.get("final_save_s"), v.get("final_save_parts"), v.get("checkpoint_log"))
PY
