#!/usr/bin/env python3
"""ad-hoc: sustained ms per launch of the default similarity path over map sizes"""
import ctypes as C, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent))
from avlmaps_amd import _lib
from bench_sim import time_call
lib = _lib.load()
D = 512
for N in (50_000, 100_000, 200_000, 300_000, 500_000, 1_000_000, 2_000_000):
    feat = torch.randn((N, D), device="cuda")
    am = torch.empty((N,), dtype=torch.int32, device="cuda")
    out = []
    for Q in (2, 64):
        q = torch.randn((Q, D), device="cuda"); q /= q.norm(dim=1, keepdim=True)
        fn = lambda: lib.avl_sim_scores_ws(feat.data_ptr(), N, D, D, q.data_ptr(), Q, D, None, am.data_ptr(), None, 0, None, 0, None)
        ms, _ = time_call(lib, fn, iters=200, warmup=100)
        out.append(f"Q={Q}: {ms*1e3:8.1f} us {N*D*4/ms/1e6:7.0f} GB/s")
    print(f"N={N:8d}  " + "   ".join(out))
