#!/bin/bash
# round 6, final tree: full GPU suite, smoke, default bench, three rehearsal launches
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_s25; mkdir -p $O; rm -f $O/rehearsal.json
timeout 2400 python -m pytest tests -m gpu -x -q > $O/t_gpu.log 2>&1; tail -3 $O/t_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python bench.py > $O/bench_default.log 2> $O/bench_default.err; python - <<'PY'
import json
for l in open("gpurun_out/r06_s25/bench_default.log"):
    if l.startswith("{"):
        d = json.loads(l); print(d["value"], d["roofline"]["frac"], d["ms_per_step"]); print(json.dumps(d["summary"])[:1800])
PY
for k in 1 2 3 4 5; do
AVLMAPS_DIST_BACKEND=gloo AVLMAPS_SHARED_GPU_LOCK=/tmp/avl_gpu.lock timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 \
    --master-addr 127.0.0.1 --master-port 2976$k bench.py --gpus 8 --workload build --steps 10000 --warmup 8 --no-cpu --trajectory spiral --spiral-radius 4 > $O/r8_$k.log 2> $O/r8_$k.err
python tools/summarize_merge.py $O/r8_$k.log --json=$O/rehearsal.json 2>&1 | sed -n 4p | cut -c1-140
done
