"""Full save of a map through MapFileWriter: grid_feat through H5Dwrite (write_threads = 1) against chunk-level pwrite()s from
several threads.  python tools/probe_parallel_save.py [voxels] [dir]"""
import os
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from avlmaps_amd.utils.mapping_utils import MapFileWriter, load_3d_map  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_600_000
    base = sys.argv[2] if len(sys.argv) > 2 else tempfile.gettempdir()
    D = 512
    rng = np.random.default_rng(0)
    feat = rng.standard_normal((n, D), dtype=np.float32)
    arr = dict(grid_feat=feat, grid_pos=rng.integers(0, 30, (n, 3)).astype(np.int32), weight=rng.random(n).astype(np.float32),
               grid_rgb=rng.integers(0, 255, (n, 3)).astype(np.uint8), occupied_ids=-np.ones((1000, 1000, 30), np.int32))
    print(f"{n} voxels, {feat.nbytes / 1e9:.2f} GB of grid_feat, {os.cpu_count()} cpus, directory {base}")
    for th in (1, 4, 8, 16, 1, 8):
        d = tempfile.mkdtemp(dir=base)
        w = MapFileWriter(os.path.join(d, "m.h5df"))
        w.write_threads = th
        t = time.perf_counter()
        w.save(arr, list(range(10)))
        dt = time.perf_counter() - t
        print(f"write_threads {th:2d}: {dt:.3f} s = {(feat.nbytes + 120e6) / dt / 1e9:.2f} GB/s")
        if th == 8:
            got = load_3d_map(os.path.join(d, "m.h5df"))
            assert np.array_equal(got[1], feat) and np.array_equal(got[2], arr["grid_pos"])
        os.system(f"rm -rf {d}")


if __name__ == "__main__":
    main()
