#!/bin/bash
# round 6: compact resident copy with residuals in fixed 2^-4 units (2 VALU per 2 elements) vs units of ulp(hi)/256 (5): same-box A/B + parity tests
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_s9; mkdir -p $O
timeout 900 python -m pytest tests/test_sim_gpu.py -m gpu -x -q 2>&1 | tail -4
python tools/ab_sim.py --reps 2 --shapes 2000000x512x64,2000000x512x65,2000000x1536x128 --modes compact,compactblocks stock ulpunits 2>&1 | tee $O/ab_compact.txt | tail -14
python - <<'PY'
import numpy as np, sys
sys.path.insert(0, ".")
import torch, bench
from avlmaps_amd import ops
feat, q = bench.make_index_inputs(torch, 200000, 512, 64, seed=1234)
pm = ops.prepare_map(feat, compact=True)
sc, am, _ = ops.sim_scores(pm, q)
want = feat.double() @ q.double().T
print("compact max abs err vs fp64 (200k x 512 x 64):", float((sc.double() - want).abs().max()), "argmax agreement", float((am.long() == want.argmax(1)).double().mean()))
PY
