"""GPU box: the resident kernel on 512 columns of rows with a wider stride (config 5's visual block) against contiguous rows.
probe_stride.py"""
import ctypes as C
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from avlmaps_amd import _lib  # noqa: E402

lib = _lib.load()
N, D, Q = 2_000_000, 512, 64
for ld in (512, 1536, 1024, 768, 2048):
    wide = torch.randn((N, ld), device="cuda")
    q = torch.randn((Q, D), device="cuda")
    q /= q.norm(dim=1, keepdim=True)
    am = torch.empty((N,), dtype=torch.int32, device="cuda")
    wsb = C.c_size_t()
    lib.avl_sim_workspace_bytes_n(N, D, Q, C.byref(wsb))
    ws = torch.empty((max(wsb.value, 64),), dtype=torch.uint8, device="cuda")
    for off in ((0,) if ld == D else (0, ld - D)):
        fn = lambda: _lib.check(lib.avl_sim_scores_ws(wide.data_ptr() + 4 * off, N, D, ld, q.data_ptr(), Q, D, None, am.data_ptr(), None, 0,
                                                      ws.data_ptr(), wsb.value, None))
        ms = bench.sustained_ms(lib, fn, launches=200, warm=80)
        print(f"row stride {ld:5d} floats, columns [{off}, {off + D}): {ms:.4f} ms  {N * D * 4 / ms / 1e6:.0f} GB/s", flush=True)
    del wide
