#!/bin/bash
# round 6: the forced-collectives one-rank line (RCCL with ONE rank: the only RCCL a 1-GPU box can run), communicator created in init_distributed
cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r06; mkdir -p $O
AVLMAPS_FORCE_COLLECTIVES=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29655 timeout 600 python bench.py --workload build --steps 10000 --build-batch 16 --no-cpu > $O/build_rccl_1rank.log 2> /tmp/e.log
grep "^{" $O/build_rccl_1rank.log | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); e = d['extra']; s = e['single_gpu_merge_path']
print(round(d['value']), 'merge+fin', e['merge_finalize_seconds'], 'cold', s.get('merge_cold_s'), 'warm', s.get('compute_total_s'), s.get('in_collectives_s'), d['extra'].get('collectives', {}).get('backend'))"
