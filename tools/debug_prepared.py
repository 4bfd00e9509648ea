#!/usr/bin/env python3
"""GPU box: raw / prepared / prepared+row-scale vs float64 over a few shapes (debug aid)."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from avlmaps_amd import ops
from avlmaps_amd.device import DeviceArray

rng = np.random.default_rng(0)
for N, D, Q in ((1000, 64, 2), (7535, 64, 2), (1000, 128, 2), (1000, 192, 7), (1000, 512, 2), (1000, 512, 65), (300, 1536, 100)):
    feat = (rng.standard_normal((N, D)) * 3).astype(np.float32)
    q = rng.standard_normal((Q, D)).astype(np.float32) / np.sqrt(D)
    ref = feat.astype(np.float64) @ q.astype(np.float64).T
    out = {}
    for how in ("auto", "prep_unscaled", "prep_scaled"):
        if how == "auto":
            src = feat
        else:
            src = ops.prepare_map(DeviceArray.from_numpy(feat), scaled=(how == "prep_scaled"))
        sc, am, _ = ops.sim_scores(src, q)
        sc, am = (x.numpy() if not isinstance(x, np.ndarray) else x for x in (sc, am))
        rs = None
        if how == "prep_scaled":
            rs = src.row_scale.numpy()
        out[how] = (np.abs(sc - ref).max(), np.isnan(sc).sum(), np.mean(am == ref.argmax(1)), None if rs is None else (rs.min(), rs.max()))
    print(N, D, Q, out, flush=True)

print("--- argmax only (scores NULL)")
for N, D, Q in ((7535, 64, 2), (1000, 512, 2), (1000, 512, 65)):
    feat = (rng.standard_normal((N, 5)) @ rng.standard_normal((5, D))).astype(np.float32)
    feat = (feat / np.linalg.norm(feat, axis=1, keepdims=True) * 14.2857).astype(np.float16).astype(np.float32)
    q = rng.standard_normal((Q, D)).astype(np.float32) / np.sqrt(D)
    ref = feat.astype(np.float64) @ q.astype(np.float64).T
    for how in ("auto", "prep_unscaled", "prep_scaled"):
        src = feat if how == "auto" else ops.prepare_map(DeviceArray.from_numpy(feat), scaled=(how == "prep_scaled"))
        _, am, _ = ops.sim_scores(src, q, want_scores=False, want_argmax=True)
        am = am.numpy() if not isinstance(am, np.ndarray) else am
        _, am2, b2 = ops.sim_scores(src, q, want_scores=False, want_argmax=True, want_best=True)
        am2 = am2.numpy() if not isinstance(am2, np.ndarray) else am2
        print(N, D, Q, how, "agree", np.mean(am == ref.argmax(1)), "with best", np.mean(am2 == ref.argmax(1)), "zeros", np.mean(am == 0), flush=True)
