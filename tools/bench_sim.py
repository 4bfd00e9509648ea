#!/usr/bin/env python3
"""Micro-benchmark of the similarity kernels (HIP events on the launch stream).  GPU box only."""
import argparse
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from avlmaps_amd import _lib  # noqa: E402


def time_call(lib, fn, iters=60, warmup=40):
    """sustained per-launch time: `iters` back-to-back launches between one event pair after `warmup` untimed ones
    (the first ~30 launches after an idle gap run slow while the power controller settles); returns (mean, mean)"""
    e0, e1 = C.c_void_p(), C.c_void_p()
    lib.avl_event_create(C.byref(e0)); lib.avl_event_create(C.byref(e1))
    for _ in range(warmup):
        fn()
    lib.avl_event_record(e0, None)
    for _ in range(iters):
        fn()
    lib.avl_event_record(e1, None)
    lib.avl_event_sync(e1)
    ms = C.c_float()
    lib.avl_event_elapsed_ms(e0, e1, C.byref(ms))
    lib.avl_event_destroy(e0); lib.avl_event_destroy(e1)
    return ms.value / iters, ms.value / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=2_000_000)
    ap.add_argument("--D", type=int, default=512)
    ap.add_argument("--Q", type=int, nargs="+", default=[1, 2, 8, 16, 32, 64, 65, 128])
    ap.add_argument("--modes", nargs="+", default=["auto"])
    a = ap.parse_args()
    lib = _lib.load()
    feat = torch.randn((a.N, a.D), device="cuda")
    for Q in a.Q:
        q = torch.randn((Q, a.D), device="cuda")
        q /= q.norm(dim=1, keepdim=True)
        sc = torch.empty((a.N, Q), device="cuda")
        am = torch.empty((a.N,), dtype=torch.int32, device="cuda")
        best = torch.empty((a.N,), device="cuda")
        wsb = C.c_size_t()
        lib.avl_sim_workspace_bytes(a.D, Q, C.byref(wsb))
        ws = torch.empty((wsb.value,), dtype=torch.uint8, device="cuda")
        for mode in a.modes:
            prec = {"auto": 0, "exact": 1, "split_f16": 2, "exact_valu": 3}[mode]
            for name, scp in (("argmax-only", None), ("scores+argmax", sc.data_ptr())):
                def fn():
                    rc = lib.avl_sim_scores_ws(feat.data_ptr(), a.N, a.D, a.D, q.data_ptr(), Q, a.D, scp, am.data_ptr(),
                                               best.data_ptr(), prec, ws.data_ptr(), wsb.value, None)
                    assert rc == 0, lib.avl_last_error()
                try:
                    med, mn = time_call(lib, fn)
                except AssertionError as e:
                    print(f"Q={Q:4d} {mode:9s} {name:14s} unsupported: {e}")
                    continue
                gb = a.N * a.D * 4 / 1e9
                print(f"Q={Q:4d} {mode:9s} {name:14s} median {med:8.3f} ms  min {mn:8.3f} ms   "
                      f"{gb / med * 1e3:8.1f} GB/s (feat read)   {a.N * Q / med / 1e6:10.1f} Gsim/s", flush=True)


if __name__ == "__main__":
    main()
