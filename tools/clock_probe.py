#!/usr/bin/env python3
"""Clock-sensitivity experiment: HBM read probes and the similarity kernel under different sclk caps.  GPU box only.

Usage: python tools/clock_probe.py [cap_mhz ...]   (needs rocm-smi write access; restores defaults at exit)"""
import ctypes as C
import subprocess
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from avlmaps_amd import _lib  # noqa: E402
from bench_sim import time_call  # noqa: E402


def sh(cmd):
    p = subprocess.run(cmd, shell=True, capture_output=True, text=True)
    return (p.stdout + p.stderr).strip()


def measure(lib, feat, q, ws, wsb, am, label):
    N, D = feat.shape
    out = {}
    for pat, name in ((0, "coalesced"), (1, "rowline")):
        g = C.c_float()
        rc = lib.avl_hbm_read_probe(feat.data_ptr(), N, D, pat, 10, C.byref(g), None)
        out[name] = g.value if rc == 0 else float("nan")
    for prec, name in ((2, "split_f16"), (3, "exact_valu")):
        def fn():
            rc = lib.avl_sim_scores_ws(feat.data_ptr(), N, D, D, q.data_ptr(), q.shape[0], D, None, am.data_ptr(), None, prec,
                                       ws.data_ptr(), wsb.value, None)
            assert rc == 0, lib.avl_last_error()
        if name == "exact_valu" and q.shape[0] > 8:
            qq = q[:8].contiguous()
            def fn():  # noqa: F811
                rc = lib.avl_sim_scores_ws(feat.data_ptr(), N, D, D, qq.data_ptr(), 8, D, None, am.data_ptr(), None, prec,
                                           ws.data_ptr(), wsb.value, None)
                assert rc == 0, lib.avl_last_error()
        med, mn = time_call(lib, fn, iters=30, warmup=5)
        out[name + "_ms"] = float(med)
    clk = sh("rocm-smi --showclocks 2>/dev/null | grep -i 'sclk\\|mclk\\|fclk' | head -6")
    print(f"[{label}] " + "  ".join(f"{k}={v:.3f}" for k, v in out.items()))
    print(clk)
    return out


def main():
    caps = [int(x) for x in sys.argv[1:]] or [1400, 1700, 2000]
    lib = _lib.load()
    N, D, Q = 2_000_000, 512, 64
    feat = torch.randn((N, D), device="cuda")
    q = torch.randn((Q, D), device="cuda"); q /= q.norm(dim=1, keepdim=True)
    am = torch.empty((N,), dtype=torch.int32, device="cuda")
    wsb = C.c_size_t(); lib.avl_sim_workspace_bytes(D, Q, C.byref(wsb))
    ws = torch.empty((wsb.value,), dtype=torch.uint8, device="cuda")
    print(sh("rocm-smi --showperflevel --showpower 2>&1 | grep -v '^=\\|^$' | head -8"))
    measure(lib, feat, q, ws, wsb, am, "default")
    try:
        for cap in caps:
            print(sh(f"rocm-smi --setperfdeterminism {cap} 2>&1 | grep -v '^=\\|^$' | head -4"))
            measure(lib, feat, q, ws, wsb, am, f"cap {cap}")
    finally:
        print(sh("rocm-smi --resetperfdeterminism 2>&1 | grep -v '^=\\|^$' | head -3"))
    measure(lib, feat, q, ws, wsb, am, "reset")


if __name__ == "__main__":
    main()
