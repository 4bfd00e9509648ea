#!/bin/bash
# round 5: side records packed / unpacked by kernels (avl_merge_side_pack / _unpack), narrow rows_add, and the single-GPU merge stand-in
# now sorts the replay log inside the timed merge (it was reusing the untimed merge's sorted log)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_s32; mkdir -p $O
timeout 900 python -m pytest tests/test_merge_kernels_gpu.py tests/test_api_gpu.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
for i in 1 2; do
AVLMAPS_MERGE_PROFILE=0 timeout 600 python bench.py --workload build --steps 10000 --no-cpu > $O/b$i.log 2> $O/prof$i.txt
grep '^{"metric"' $O/b$i.log | python -c "
import sys,json
j=json.loads(sys.stdin.read()); s=j['extra']['single_gpu_merge_path']; print({k:round(1e3*v,2) for k,v in s['wall_s'].items()}, 'total', round(1e3*s['compute_total_s'],2))"
done
