#!/bin/bash
# round 6: parallel full save (chunk-level pwrite) on the GPU box's host; weight-equality counts of the exact-replay tests
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_s11; mkdir -p $O
python tools/probe_parallel_save.py 1600000 /tmp 2>&1 | tee $O/parallel_save.txt
df -h /tmp | tail -1
python - <<'PY'
import numpy as np, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from oracle import avl_oracle as O
import test_builder_gpu as T
from avlmaps_amd import ops
for name in ("g2a_builder_small.npz", "g2b_builder_growth.npz"):
    g = np.load("tests/golden/" + name)
    Ts = O.pc_transforms(g["poses_rt"], g["base_transform"], g["base2cam_tf"])
    acc = T.run_gpu_builder(ops, int(g["gs"]), float(g["cs"]), float(g["camera_height"]), g["calib"], Ts, g["depths"], g["rgbs"], g["feats"], g["samples"], capacity=2000, replay=True)
    out = acc.finalize()
    w = g["weight"].astype(np.float32)
    print(name, "weight mismatches", int((out["weight"] != w).sum()), "of", len(w), "rgb mismatches", int((out["grid_rgb"] != np.floor(g["grid_rgb"]).astype(np.uint8)).sum()))
# the medium oracle scene
rng = np.random.default_rng(7)
H, W, Hf, Wf, D, nfr, rate = 120, 160, 58, 77, 64, 24, 5
gs, cs, cam_h = 1000, 0.05, 1.5
calib = np.array([W / 2, 0, W / 2, 0, W / 2, H / 2, 0, 0, 1.0])
depths, rgbs, feats, poses = T.synth_scene(rng, nfr, H, W, Hf, Wf, D)
b2c, bt = O.setup_transforms([1, 0, 0, 0, -1, 0, 0, 0, -1], cam_h, [0, 0, -1], [-1, 0, 0], [0, 1, 0])
Ts = O.pc_transforms(poses, bt, b2c)
rs = np.random.RandomState(3)
samples = [O.sample_indices(rs, H * W, rate) for _ in range(nfr)]
om = O.OracleMap(gs, cs, cam_h, D)
for i in range(nfr):
    om.integrate(depths[i], calib, Ts[i], samples[i], feats[i], rgbs[i])
ref = om.export()
accr = T.run_gpu_builder(ops, gs, cs, cam_h, calib, Ts, depths, rgbs, feats, samples, capacity=200_000, replay=True)
outr = accr.finalize()
print("medium scene: weight mismatches", int((outr["weight"] != ref["weight"]).sum()), "of", len(ref["weight"]), "max rel", float(np.max(np.abs(outr["weight"] - ref["weight"]) / ref["weight"])))
PY
