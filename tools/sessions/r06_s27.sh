#!/bin/bash
# round 6: the default bench's build block at N = 2 and N = 4 on one GPU (gloo): blocks of 1.1 M / 0.56 M rows per owner -> the chunked exchange at the driver's sizes
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_s27; mkdir -p $O
for n in 2 4; do
AVLMAPS_DIST_BACKEND=gloo AVLMAPS_SHARED_GPU_LOCK=/tmp/avl_gpu.lock timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n \
    --master-addr 127.0.0.1 --master-port 2977$n bench.py --gpus $n --workload build --steps 10000 --warmup 8 --no-cpu > $O/r$n.log 2> $O/r$n.err
python - $O/r$n.log <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l); e = d["extra"]; mb = e.get("merge_breakdown") or {}
        print("n_gpus", d["n_gpus"], "frames/s", round(d["value"]), "voxels", e.get("voxels_merged"), "chunks", mb.get("exchange_chunks"), "chunk rows", mb.get("exchange_chunk_rows"),
              "buffers MB", round((mb.get("exchange_buffer_bytes") or 0) / 1e6), "payload MB", round((mb.get("payload_bytes_sent") or 0) / 1e6), "compute ms", round(1e3 * (mb.get("compute_total_s") or 0), 2))
PY
tail -2 $O/r$n.err | cut -c1-200
done
timeout 900 python -m pytest tests/test_api_gpu.py -x -q -k "two_ranks or seeded" > $O/t.log 2>&1; tail -2 $O/t.log
