#!/bin/bash
# round 6: ordered sums in the generic-width fuse kernel (D > 1536); full map save chunk by chunk (H5Dwrite_chunk)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_s19; mkdir -p $O
timeout 900 python -m pytest tests/test_builder_gpu.py -x -q -k "widths or reproducible or collisions" > $O/t_builder.log 2>&1; tail -4 $O/t_builder.log
timeout 400 python tools/fuzz_parity.py 150 21 > $O/fuzz_parity.log 2>&1; tail -4 $O/fuzz_parity.log | cut -c1-300
timeout 600 python tools/probe_parallel_save.py 1600000 /tmp > $O/parallel_save.log 2>&1; cat $O/parallel_save.log
timeout 600 python tools/probe_build_width.py 2048 1 400 > $O/width2048.log 2>&1; tail -3 $O/width2048.log | cut -c1-300
timeout 900 python bench.py > $O/bench_default.log 2> $O/bench_default.err; python - <<'PY'
import json
for l in open("gpurun_out/r06_s19/bench_default.log"):
    if l.startswith("{"):
        d = json.loads(l); print(json.dumps(d["summary"])[:1500]); print(d["value"], d["roofline"])
PY
