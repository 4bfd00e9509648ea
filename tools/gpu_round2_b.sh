#!/bin/bash
# GPU box, round 2 call B: full GPU test suite + range-guard / ring A/B + default bench
cd "$(dirname "$0")/.."
O=gpurun_out/r02b; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -25 $O/pytest_gpu.log
timeout 900 python tools/ab_sim.py --reps 2 stock noguard ring3 > $O/ab_sim.log 2>&1
tail -12 $O/ab_sim.log
timeout 600 python tools/ab_sim.py --reps 1 --modes prepared --shapes 2000000x512x64,2000000x1536x128 stock > $O/ab_sim_prepared.log 2>&1
tail -4 $O/ab_sim_prepared.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
head -c 3000 $O/bench_default.json
AVLMAPS_FORCE_COLLECTIVES=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29655 timeout 600 python bench.py --workload build --steps 10000 --build-batch 16 --no-cpu > $O/build_rccl_1rank.json 2> $O/build_rccl_1rank.err; echo "rccl1 rc=$?"
tail -3 $O/build_rccl_1rank.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r02b/build_rccl_1rank.json') if l.startswith('{')][0])
print(d['value'], d['extra']['single_gpu_merge_path'])
PY
