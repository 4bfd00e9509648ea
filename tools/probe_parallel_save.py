"""Full save of a map through MapFileWriter: grid_feat through ONE H5Dwrite, chunk by chunk through H5Dwrite_chunk
(the default since round 6), and through chunk-level pwrite()s from several threads into early-allocated chunks (write_threads > 1).
python tools/probe_parallel_save.py [voxels] [dir]"""
import os
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from avlmaps_amd.utils import h5lite  # noqa: E402
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
    default = h5lite.H5File.DIRECT_CHUNK_BYTES
    for label, direct, th in (("H5Dwrite", False, 1), ("H5Dwrite_chunk", True, 1), ("pwrite x 8", False, 8), ("H5Dwrite", False, 1),
                              ("H5Dwrite_chunk", True, 1), ("H5Dwrite_chunk", True, 1)):
        d = tempfile.mkdtemp(dir=base)
        h5lite.H5File.DIRECT_CHUNK_BYTES = default if direct else 1 << 62
        w = MapFileWriter(os.path.join(d, "m.h5df"))
        w.write_threads = th
        t = time.perf_counter()
        w.save(arr, list(range(10)))
        dt = time.perf_counter() - t
        print(f"{label:<15s}: {dt:.3f} s = {(feat.nbytes + 120e6) / dt / 1e9:.2f} GB/s", flush=True)
        if direct or th > 1:
            got = load_3d_map(os.path.join(d, "m.h5df"))
            assert np.array_equal(got[1], feat) and np.array_equal(got[2], arr["grid_pos"])
            h5dump = "/opt/conda/bin/h5dump" if os.path.exists("/opt/conda/bin/h5dump") else "h5dump"
            os.system(f"{h5dump} -H {d}/m.h5df 2>/dev/null | grep -A3 'DATASET \"grid_feat\"' | tr -s ' \n' ' '; echo")
        os.system(f"rm -rf {d}")


if __name__ == "__main__":
    main()
