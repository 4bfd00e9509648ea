#!/bin/bash
# round 5: replay log sorted without the selection pass, replay walk with four entries in flight
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_s33; mkdir -p $O
timeout 1500 python -m pytest tests/test_merge_kernels_gpu.py tests/test_api_gpu.py tests/test_builder_gpu.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
AVLMAPS_MERGE_PROFILE=0 timeout 600 python bench.py --workload build --steps 10000 --no-cpu > $O/b0.log 2> $O/prof0.txt
for i in 1 2; do
timeout 600 python bench.py --workload build --steps 10000 --no-cpu 2>/dev/null > $O/b$i.log
grep '^{"metric"' $O/b$i.log | python -c "
import sys,json
j=json.loads(sys.stdin.read()); s=j['extra']['single_gpu_merge_path']; print({k:round(1e3*v,2) for k,v in s['wall_s'].items()}, 'total', round(1e3*s['compute_total_s'],2), 'plain finalize', round(1e3*s['plain_finalize_s'],2))"
done
