#!/usr/bin/env python3
"""ad-hoc: read-probe rate vs footprint (L2 / Infinity Cache / HBM resident) for the coalesced and the row-line pattern"""
import ctypes as C, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from avlmaps_amd import _lib
lib = _lib.load()
buf = torch.randn((2_000_000, 512), device="cuda")
g = C.c_float()
for rows in (4096, 8192, 16384, 32768, 65536, 131072, 262144, 524288, 2_000_000):
    mb = rows * 2048 / 2**20
    out = []
    for pat in (2, 3):
        iters = max(20, min(2000, int(4e9 / (rows * 2048))))
        lib.avl_hbm_read_probe(buf.data_ptr(), rows, 512, pat, iters, C.byref(g), None)
        out.append(g.value)
    print(f"{mb:8.0f} MiB  coalesced {out[0]:9.0f} GB/s   rowline {out[1]:9.0f} GB/s")
