#!/usr/bin/env python3
"""ad-hoc: row-line read probe at 1 / 2 / 3 workgroups (8 / 16 / 24 waves) per CU, sustained"""
import ctypes as C, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from avlmaps_amd import _lib
lib = _lib.load()
buf = torch.randn((2_000_000, 512), device="cuda")
g = C.c_float()
for rep in range(2):
    for name, pat in (("1 WG/CU", 3 | 4), ("2 WG/CU", 3), ("3 WG/CU", 3 | 8), ("coalesced", 2)):
        lib.avl_hbm_read_probe(buf.data_ptr(), 2_000_000, 512, pat, 300, C.byref(g), None)
        print(f"{name:10s} {g.value:8.0f} GB/s  {4096 / g.value:.3f} ms")
