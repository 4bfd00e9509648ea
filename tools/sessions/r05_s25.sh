#!/bin/bash
# round 5: pooled temporaries of the log segments / finalize; strided-view test; merge path timing
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_s25; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
timeout 600 python bench.py --no-pmc > $O/bench_default.log 2>&1
grep '^{"metric"' $O/bench_default.log | python -c "
import sys,json
j=json.loads(sys.stdin.read()); e=j['extra']['map_build_strong']; s=e['single_gpu_merge_path']; print({k:round(1e3*v,2) for k,v in s['wall_s'].items()}, 'total', round(1e3*s['compute_total_s'],2)); print('finalize ms', 1e3*e['merge_finalize_seconds'], 'frames/s', e['frames_per_s'], e.get('host_loop'))
for k in ('map_build_strong_deferred_fuse','map_build_strong_batched64'):
    print(k, j['extra'][k]['frames_per_s'])"
