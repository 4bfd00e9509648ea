#!/bin/bash
# round 5: the fold places the received rows with avl_scatter_rows
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_s18; mkdir -p $O
timeout 300 python tools/probe_merge_plan.py 2250000 > $O/merge_plan_ops.txt 2>&1
head -12 $O/merge_plan_ops.txt | grep iter
timeout 900 python -m pytest tests/test_api_gpu.py tests/test_merge_kernels_gpu.py -m gpu -x -q > $O/pytest_api.txt 2>&1; tail -3 $O/pytest_api.txt
timeout 600 python bench.py --no-pmc > $O/bench_default.log 2>&1
grep '^{"metric"' $O/bench_default.log | python -c "
import sys,json
j=json.loads(sys.stdin.read()); e=j['extra']['map_build_strong']; print(json.dumps(e['single_gpu_merge_path'])[:1500]); print('finalize', e['merge_finalize_seconds'])"
