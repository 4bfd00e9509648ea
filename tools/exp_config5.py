#!/usr/bin/env python3
"""ad-hoc: config 5 (2M x 1536, 128 queries) with dense vs modality-block-structured queries"""
import ctypes as C, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent))
from avlmaps_amd import _lib
from bench_sim import time_call
lib = _lib.load()
N, D, Q = 2_000_000, 1536, 128
feat = torch.randn((N, D), device="cuda")
am = torch.empty((N,), dtype=torch.int32, device="cuda")
wsb = C.c_size_t(); lib.avl_sim_workspace_bytes(D, Q, C.byref(wsb)); ws = torch.empty((wsb.value,), dtype=torch.uint8, device="cuda")
qd = torch.randn((Q, D), device="cuda"); qd /= qd.norm(dim=1, keepdim=True)
qb = qd.clone(); qb[:64, 512:] = 0; qb[64:, :512] = 0          # 64 text queries (visual block) then 64 audio queries
qi = qd.clone(); qi[0::2, 512:] = 0; qi[1::2, :512] = 0        # interleaved modalities: no MFMA tile is modality-pure
for name, q in (("dense", qd), ("block (tile-pure)", qb), ("block (interleaved)", qi)):
    q = q.contiguous()
    fn = lambda: lib.avl_sim_scores_ws(feat.data_ptr(), N, D, D, q.data_ptr(), Q, D, None, am.data_ptr(), None, 0, ws.data_ptr(), wsb.value, None)
    ms, _ = time_call(lib, fn, iters=60, warmup=40)
    ref = (feat[:4096].double() @ q.double().T).argmax(1)
    ok = (ref == am[:4096].long()).double().mean().item()
    print(f"{name:22s} {ms:7.3f} ms  {N*D*4/ms/1e6:6.0f} GB/s  argmax agreement {ok:.4f}")
