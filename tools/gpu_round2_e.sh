#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r02e; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=" $O/pytest_gpu.log | head -20; grep -n "^E  " $O/pytest_gpu.log | head -30
timeout 600 python bench.py --workload build --steps 512 --warmup 8 --no-cpu --feature-standin vit-l16 > $O/build_vit_standin.json 2> $O/build_vit_standin.err; echo "vit rc=$?"
timeout 600 python bench.py --workload build --steps 512 --warmup 8 --no-cpu --feature-standin vit-l16 --build-batch 16 > $O/build_vit_standin_b16.json 2>> $O/build_vit_standin.err
python - <<'PY'
import json
for f in ('build_vit_standin','build_vit_standin_b16'):
    try:
        d=json.loads([l for l in open(f'gpurun_out/r02e/{f}.json') if l.startswith('{')][0]); e=d['extra']
        print(f, d['value'], 'frames/s', e['seconds'], e['fuse_seconds_max_rank'], e['merge_finalize_seconds'])
    except Exception as ex: print(f, 'failed', ex)
PY
tail -3 $O/build_vit_standin.err
timeout 300 python tools/ab_sim.py --reps 2 --shapes 2000000x512x65,2000000x512x2,2000000x512x32 stock > $O/ab_q.log 2>&1; tail -4 $O/ab_q.log
timeout 300 python tools/power_probe.py 3 2>&1 | grep -v '^/sys/class/drm\|amdgpu.ids' > $O/power_probe.txt; tail -6 $O/power_probe.txt
