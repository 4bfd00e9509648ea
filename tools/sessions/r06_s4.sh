#!/bin/bash
# round 6: where the cold first merge goes (single GPU, 10k frames)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_s4; mkdir -p $O
AVLMAPS_MERGE_TRACE=1 timeout 600 python bench.py --workload build --steps 10000 --warmup 8 --no-cpu --deferred-fuse > $O/b1.log 2> $O/b1.err
grep "merge2 trace" $O/b1.err | cut -c1-500
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/probe_exp.hip -o /tmp/probe_exp 2>/dev/null && /tmp/probe_exp
python - <<'PY'
import numpy as np, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from oracle import avl_oracle as O
g = np.load("tests/golden/g2a_builder_small.npz")
import test_builder_gpu as T
from avlmaps_amd import ops
Ts = O.pc_transforms(g["poses_rt"], g["base_transform"], g["base2cam_tf"])
acc = T.run_gpu_builder(ops, int(g["gs"]), float(g["cs"]), float(g["camera_height"]), g["calib"], Ts, g["depths"], g["rgbs"], g["feats"], g["samples"], capacity=2000, replay=True)
out = acc.finalize()
w = g["weight"].astype(np.float32)
print("g2a weight mismatches", int((out["weight"] != w).sum()), "of", len(w))
PY
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/probe_xcd_persist.hip -o /tmp/probe_xcd 2>/dev/null && for ov in 90 60 0; do /tmp/probe_xcd 2140 $ov 400; done | tee $O/xcd_persist.txt
