"""GPU box: the compact (3-byte) prepared map against the 4-byte prepared map: accuracy vs float64 and time per query pass."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(__file__), "..")))
import bench  # noqa: E402
from avlmaps_amd import ops  # noqa: E402

N, D, Q = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000, 512, 64
feat, q = bench.make_index_inputs(torch, N, D, Q, seed=1234)
ref = None
if N <= 300_000:
    ref = feat.double() @ q.double().T
pm24 = ops.prepare_map(feat, compact=True)
pm32 = ops.prepare_map(feat.clone(), scaled=True)
for name, pm in (("4-byte prepared", pm32), ("3-byte compact ", pm24)):
    sc, am, best = ops.sim_scores(pm, q, want_best=True)
    if ref is not None:
        err = (sc.double() - ref).abs()
        agree = (am.long() == ref.argmax(dim=1)).float().mean().item()
        print(f"{name}: max |score - float64| = {err.max().item():.3e}, rms = {err.pow(2).mean().sqrt().item():.3e}, argmax agreement {agree:.6f}")
    assert torch.equal(am.long(), sc.argmax(dim=1)) and torch.equal(best, sc.gather(1, am.long()[:, None])[:, 0])
    am2 = torch.empty((N,), dtype=torch.int32, device="cuda")
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            ops.sim_scores(pm, q, want_scores=False, out_argmax=am2)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 200
    print(f"{name}: {1e3 * dt:.4f} ms per pass ({N * Q / dt / 1e9:.1f} G similarities/s)")
