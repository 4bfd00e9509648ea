#!/bin/bash
# round 6: what a chunked payload exchange costs a rank in compute -- eight ranks as threads of one process (no process hand-overs)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_s17; mkdir -p $O
for R in 1000000 65536 32768 8192; do
AVLMAPS_MERGE_CHUNK_ROWS=$R timeout 600 python tools/probe_merge2.py 8 10000 4 > $O/probe_R$R.log 2>&1
echo "== chunk rows $R"; python - <<PY
import ast
rep = {}
cur = None
for l in open("$O/probe_R$R.log"):
    if l.startswith("--- merge"):
        cur = l.strip(); rep[cur] = []
    elif cur and l[:1].isdigit():
        rep[cur].append(ast.literal_eval(l.split(" ", 1)[1]))
for k, v in rep.items():
    print(k, "chunks", v[0]["chunks"], "compute ms", [x["compute_ms"] for x in v])
print("  phases rank 3, merge 2:", list(rep.values())[2][3]["phases_ms"])
PY
done
