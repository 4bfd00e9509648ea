#!/usr/bin/env python3
"""ad-hoc: per-step duration series of the similarity kernel from a cold start (power-management transient)"""
import ctypes as C, sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from avlmaps_amd import _lib
lib = _lib.load()
N, D, Q = 2_000_000, 512, 64
feat = torch.randn((N, D), device="cuda"); q = torch.randn((Q, D), device="cuda"); q /= q.norm(dim=1, keepdim=True)
am = torch.empty((N,), dtype=torch.int32, device="cuda")
wsb = C.c_size_t(); lib.avl_sim_workspace_bytes(D, Q, C.byref(wsb)); ws = torch.empty((wsb.value,), dtype=torch.uint8, device="cuda")
K = 800
evs = []
for _ in range(K + 1):
    e = C.c_void_p(); lib.avl_event_create(C.byref(e)); evs.append(e)
for rep in range(2):
    torch.cuda.synchronize(); time.sleep(2.0)
    for i in range(K):
        lib.avl_event_record(evs[i], None)
        lib.avl_sim_scores_ws(feat.data_ptr(), N, D, D, q.data_ptr(), Q, D, None, am.data_ptr(), None, 2, ws.data_ptr(), wsb.value, None)
    lib.avl_event_record(evs[K], None); lib.avl_event_sync(evs[K])
    ms = C.c_float(); t = []
    for i in range(K):
        lib.avl_event_elapsed_ms(evs[i], evs[i + 1], C.byref(ms)); t.append(ms.value)
    t = np.array(t)
    print("rep", rep, " ".join(f"[{a}:{b}]={t[a:b].mean():.3f}" for a, b in ((0, 5), (5, 15), (15, 30), (30, 55), (55, 100), (100, 150), (150, 200), (200, 300), (300, 500), (500, 800))))
