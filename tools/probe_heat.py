"""GPU box: avl_heatmap_from_mask at 2 M voxels, uniform and clustered targets; run under rocprofv3 --kernel-trace --stats for the split.
probe_heat.py [uniform|clustered] [reps]"""
import os
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from avlmaps_amd import _lib  # noqa: E402

lib = _lib.load()
kind = sys.argv[1] if len(sys.argv) > 1 else "uniform"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
N = 2_000_000
g = torch.Generator(device="cuda").manual_seed(11)
if os.environ.get("HEAT_GEOMETRY", "map") == "cube":      # the bench's first geometry: a 306^3 cube at 7 % occupancy (5 words per column)
    side = int(round((N / 0.07) ** (1 / 3))) + 1
    lin = torch.randperm(side ** 3, device="cuda", generator=g)[:N]
    pos = torch.stack([lin // (side * side), (lin // side) % side, lin % side], 1).to(torch.int32).contiguous()
else:                                                      # a map of the reference's shape: 1000 x 1000 x 30 cells, 2 M of them occupied
    lin = torch.randperm(1000 * 1000 * 30, device="cuda", generator=g)[:N]
    pos = torch.stack([lin // 30000, (lin // 30) % 1000, lin % 30], 1).to(torch.int32).contiguous()
heat = torch.empty((N,), dtype=torch.float32, device="cuda")
if kind == "uniform":
    mask = (torch.rand(N, device="cuda", generator=g) < 1 / 64).to(torch.uint8)
else:
    c = pos[torch.randint(0, N, (6,), device="cuda", generator=g)]
    mask = ((pos[:, None, :] - c[None]).abs().amax(dim=2) <= 12).any(dim=1).to(torch.uint8)
def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


ms = timed(lambda: _lib.check(lib.avl_heatmap_from_mask(pos.data_ptr(), mask.data_ptr(), N, 0.05, 0.01, heat.data_ptr(), None)))
ref = heat.clone()
print(f"{kind} targets ({int(mask.sum())} of {N}), stateless call: {ms:.3f} ms, nonzero heat {int((heat > 0).sum())}")
import ctypes as C  # noqa: E402
h = C.c_void_p()
torch.cuda.synchronize()
t = time.perf_counter()
_lib.check(lib.avl_heat_plan_create(C.byref(h), pos.data_ptr(), N, None))
t_plan = (time.perf_counter() - t) * 1e3
heat.zero_()
ms = timed(lambda: _lib.check(lib.avl_heatmap_from_mask_planned(h, mask.data_ptr(), 0.05, 0.01, heat.data_ptr(), None)))
print(f"{kind} targets, planned call: {ms:.3f} ms (plan built once in {t_plan:.2f} ms), same bits: {bool(torch.equal(ref, heat))}")
lib.avl_heat_plan_destroy(h)
