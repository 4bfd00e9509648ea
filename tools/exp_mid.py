#!/usr/bin/env python3
"""ad-hoc: sustained us per launch at scene-sized maps for 2 / 65 query columns"""
import ctypes as C, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent))
from avlmaps_amd import _lib
from bench_sim import time_call
lib = _lib.load()
D = 512
for N in (100_000, 300_000, 600_000):
    feat = torch.randn((N, D), device="cuda"); am = torch.empty((N,), dtype=torch.int32, device="cuda")
    out = []
    for Q in (2, 41, 65):
        q = torch.randn((Q, D), device="cuda") / 22
        fn = lambda: lib.avl_sim_scores_ws(feat.data_ptr(), N, D, D, q.data_ptr(), Q, D, None, am.data_ptr(), None, 0, None, 0, None)
        ms, _ = time_call(lib, fn, iters=300, warmup=100)
        out.append(f"Q={Q}: {ms*1e3:7.1f} us")
    print(f"N={N:7d}  " + "  ".join(out))
