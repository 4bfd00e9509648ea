#!/bin/bash
# round 6: N = 2 / N = 4 builds on one GPU (gloo), chunked exchange (default 1024 MB) against one exchange (AVLMAPS_MERGE_CHUNK_MB=1000000), twice each
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_s28; mkdir -p $O
for rep in 1 2; do for n in 2 4; do for mb in 1024 1000000; do
AVLMAPS_MERGE_CHUNK_MB=$mb AVLMAPS_DIST_BACKEND=gloo AVLMAPS_SHARED_GPU_LOCK=/tmp/avl_gpu.lock timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n \
    --master-addr 127.0.0.1 --master-port 2978$n bench.py --gpus $n --workload build --steps 10000 --warmup 8 --no-cpu > $O/r${n}_${mb}_$rep.log 2> $O/r${n}_${mb}_$rep.err
python - $O/r${n}_${mb}_$rep.log $mb <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l); e = d["extra"]; mb = e.get("merge_breakdown") or {}
        print("n_gpus", d["n_gpus"], "chunk MB", sys.argv[2], "chunks", mb.get("exchange_chunks"), "merge+finalize s", round(e.get("merge_finalize_seconds", 0), 3),
              "compute ms rank0", round(1e3 * (mb.get("compute_total_s") or 0), 2), {k: round(1e3 * v, 2) for k, v in (mb.get("compute_s") or {}).items()},
              "per rank", [round(1e3 * x, 1) for x in (e.get("merge_compute_s_per_rank") or [])])
PY
done; done; done
