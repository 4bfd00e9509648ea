#!/usr/bin/env python3
"""ad-hoc: per-step time of the similarity kernel vs run length, data distribution and the best-score output"""
import ctypes as C, sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent))
from avlmaps_amd import _lib
import importlib.util
spec = importlib.util.spec_from_file_location("bench", str(Path(__file__).resolve().parent.parent / "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
lib = _lib.load()
N, D, Q = 2_000_000, 512, 64
feat_l, q_l = bench.make_index_inputs(torch, N, D, Q, 1234)
feat_r = torch.randn((N, D), device="cuda"); q_r = torch.randn((Q, D), device="cuda"); q_r /= q_r.norm(dim=1, keepdim=True)
am = torch.empty((N,), dtype=torch.int32, device="cuda"); best = torch.empty((N,), device="cuda")
wsb = C.c_size_t(); lib.avl_sim_workspace_bytes(D, Q, C.byref(wsb)); ws = torch.empty((wsb.value,), dtype=torch.uint8, device="cuda")
e0, e1 = C.c_void_p(), C.c_void_p(); lib.avl_event_create(C.byref(e0)); lib.avl_event_create(C.byref(e1))
def run(feat, q, bestp, steps, warm):
    f = lambda: lib.avl_sim_scores_ws(feat.data_ptr(), N, D, D, q.data_ptr(), Q, D, None, am.data_ptr(), bestp, 2, ws.data_ptr(), wsb.value, None)
    for _ in range(warm): f()
    lib.avl_event_record(e0, None)
    for _ in range(steps): f()
    lib.avl_event_record(e1, None); lib.avl_event_sync(e1)
    ms = C.c_float(); lib.avl_event_elapsed_ms(e0, e1, C.byref(ms)); return ms.value / steps
for rep in range(2):
    for dname, (f_, q_) in (("lseg-like", (feat_l, q_l)), ("randn", (feat_r, q_r))):
        for bname, bp in (("best", best.data_ptr()), ("nobest", None)):
            for steps, warm in ((50, 5), (500, 50), (3000, 300)):
                torch.cuda.synchronize(); time.sleep(0.3)
                print(f"{dname:10s} {bname:7s} steps={steps:5d} warm={warm:4d}  {run(f_, q_, bp, steps, warm):.4f} ms/step")
