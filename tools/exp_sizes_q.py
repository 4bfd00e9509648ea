#!/usr/bin/env python3
"""ad-hoc: sustained us per launch over map sizes for a given query count / feature width"""
import ctypes as C, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent))
from avlmaps_amd import _lib
from bench_sim import time_call
lib = _lib.load()
D, Q = int(sys.argv[1]), int(sys.argv[2])
q = torch.randn((Q, D), device="cuda"); q /= q.norm(dim=1, keepdim=True)
wsb = C.c_size_t(); lib.avl_sim_workspace_bytes(D, Q, C.byref(wsb)); ws = torch.empty((wsb.value,), dtype=torch.uint8, device="cuda")
for N in (50_000, 100_000, 200_000, 300_000, 500_000, 1_000_000, 2_000_000):
    feat = torch.randn((N, D), device="cuda")
    am = torch.empty((N,), dtype=torch.int32, device="cuda")
    fn = lambda: lib.avl_sim_scores_ws(feat.data_ptr(), N, D, D, q.data_ptr(), Q, D, None, am.data_ptr(), None, 0, ws.data_ptr(), wsb.value, None)
    ms, _ = time_call(lib, fn, iters=150, warmup=80)
    print(f"D={D} Q={Q} N={N:8d}  {ms*1e3:8.1f} us {N*D*4/ms/1e6:7.0f} GB/s")
