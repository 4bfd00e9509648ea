"""GPU box: where the final save of a map goes -- device finalisation, device-to-host copy, HDF5 write.  probe_final_save.py [voxels]"""
import sys
import tempfile
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from avlmaps_amd import _lib, ops  # noqa: E402
from avlmaps_amd.device import DeviceArray, PinnedBuffer  # noqa: E402
from avlmaps_amd.utils.mapping_utils import MapFileWriter  # noqa: E402

n, D = int(sys.argv[1]) if len(sys.argv) > 1 else 1_580_000, 512
lib = _lib.load()
gf = DeviceArray((n, D), np.float32)
torch.cuda.synchronize()
for rep in range(2):
    t = time.perf_counter()
    host = gf.numpy()
    print(f"pageable D2H of {gf.nbytes / 1e9:.2f} GB: {time.perf_counter() - t:.3f} s (fresh np.empty)")
    t = time.perf_counter()
    _lib.check(lib.avl_memcpy_d2h(host.ctypes.data, gf.ptr, gf.nbytes, None))
    print(f"  ... into the same (touched) array: {time.perf_counter() - t:.3f} s")
pb = PinnedBuffer()
t = time.perf_counter()
pb.reserve(256 << 20)
print(f"pinning 256 MB: {time.perf_counter() - t:.3f} s")
t = time.perf_counter()
for off in range(0, gf.nbytes, 256 << 20):
    m = min(256 << 20, gf.nbytes - off)
    _lib.check(lib.avl_memcpy_d2h(pb.ptr, gf.ptr + off, m, None))
print(f"pinned D2H in 256 MB pieces: {time.perf_counter() - t:.3f} s")
arrays = dict(grid_feat=host, grid_pos=np.zeros((n, 3), np.int32), weight=np.zeros(n, np.float32), grid_rgb=np.zeros((n, 3), np.uint8),
              occupied_ids=-np.ones((1000, 1000, 30), np.int32))
with tempfile.TemporaryDirectory() as tmp:
    w = MapFileWriter(Path(tmp) / "vlmaps.h5df")
    t = time.perf_counter()
    w.save(arrays, list(range(100)))
    print(f"MapFileWriter.save (full) of {n} voxels: {time.perf_counter() - t:.3f} s  {w.stats[-1]}")
for rep in range(2):
    t = time.perf_counter()
    host = gf.numpy()
    print(f"DeviceArray.numpy() (threaded first touch) of {gf.nbytes / 1e9:.2f} GB: {time.perf_counter() - t:.3f} s")
    del host
# (round 4: filling the chunks of grid_feat with pwrite from 8 threads -- addresses from H5Dget_chunk_info -- was tried against H5Dwrite:
#  2.9 s vs 0.58 s for 3.24 GB on the GPU box: buffered writes to ONE file serialise on the inode lock, and 25 k chunk-info calls
#  cost more than they save.  The map file is written by the library.)
