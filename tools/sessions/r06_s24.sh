#!/bin/bash
# round 6: the merge's small sorts as onesweep radix sorts (stock) against rocPRIM's default merge sort below 2^20 items (variant), same box, interleaved
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_s24; mkdir -p $O

for k in 1 2 3; do
for v in stock m2mergesort; do
  lib=""; [ $v != stock ] && lib="variants/libavlmaps_hip_$v.so"
  AVLMAPS_HIP_LIB=$lib timeout 600 python tools/probe_merge2.py 8 10000 4 > $O/probe_${v}_$k.log 2>&1
  python - $O/probe_${v}_$k.log $v <<'PY'
import ast, sys
rep = {}; cur = None
for l in open(sys.argv[1]):
    if l.startswith("--- merge"): cur = l.strip(); rep[cur] = []
    elif cur and l[:1].isdigit(): rep[cur].append(ast.literal_eval(l.split(" ", 1)[1]))
for k, v in list(rep.items())[1:3]:
    c = [x["compute_ms"] for x in v]; p = [x["phases_ms"]["plan"] for x in v]
    print(sys.argv[2], k, "compute ms min/median/max %.2f %.2f %.2f" % (min(c), sorted(c)[4], max(c)), "plan %.2f-%.2f" % (min(p), max(p)))
PY
done; done
