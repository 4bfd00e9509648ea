"""Host only: ms per 720x1080 frame of the reference pixel sampling (avl_mt19937_shuffle_sample) and of the draws alone
(avl_mt19937_skip_shuffles), for the widest SIMD form of this host; AVL_NO_AVX512=1 / AVL_NO_AVX2=1 select the narrower ones."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from avlmaps_amd.map.vlmap_builder import VLMapBuilder  # noqa: E402

n_pix, reps = 720 * 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 200
np.random.seed(0)
VLMapBuilder.sample_pixels(n_pix, 100)
best_s = best_k = 1e9
for _ in range(5):
    t = time.perf_counter()
    for _ in range(reps // 5):
        VLMapBuilder.sample_pixels(n_pix, 100)
    best_s = min(best_s, (time.perf_counter() - t) / (reps // 5))
    t = time.perf_counter()
    VLMapBuilder.skip_pixel_shuffles(reps // 5, n_pix)
    best_k = min(best_k, (time.perf_counter() - t) / (reps // 5))
print(f"sample {1e3 * best_s:.3f} ms/frame, draws only {1e3 * best_k:.3f} ms/frame (best of 5 x {reps // 5})")
