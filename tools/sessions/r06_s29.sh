#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_s29; mkdir -p $O
for ws in 2 4; do for mb in 1024 1000000; do
AVLMAPS_MERGE_CHUNK_MB=$mb timeout 600 python tools/probe_merge2.py $ws 10000 4 > $O/probe_ws${ws}_$mb.log 2>&1
python - $O/probe_ws${ws}_$mb.log $ws $mb <<'PY'
import ast, sys
rep = {}; cur = None
for l in open(sys.argv[1]):
    if l.startswith("--- merge"): cur = l.strip(); rep[cur] = []
    elif cur and l[:1].isdigit(): rep[cur].append(ast.literal_eval(l.split(" ", 1)[1]))
for k, v in list(rep.items())[1:3]:
    print("ws", sys.argv[2], "chunk MB", sys.argv[3], k, "chunks", v[0]["chunks"], "M", v[0]["M"], "n", v[0]["n"], "compute ms", [x["compute_ms"] for x in v], v[0]["phases_ms"])
PY
done; done
