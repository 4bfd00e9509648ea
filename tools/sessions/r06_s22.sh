#!/bin/bash
# round 6: the default bench three times: does the cold first VLMapBuilder leg stall, and in which step?
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_s22; mkdir -p $O
for k in 1 2 3; do
timeout 900 python bench.py > $O/bench_$k.log 2> $O/bench_$k.err
python - $O/bench_$k.log <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l); print(d["value"], d["roofline"]["frac"], json.dumps(d["summary"]["pipeline_frames_per_s"]))
        for k, v in d["extra"]["vlmapbuilder_pipeline"].items():
            if isinstance(v, dict): print("  ", k, v.get("slow_steps_over_50ms"), v.get("final_save_parts"))
PY
done
