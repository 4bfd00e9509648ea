"""Host only: frames/s of VLMapBuilder._frame_stream with reference pixel sampling (720x1080, rate 100, in-memory frames) for several sampler_workers settings -- the RNG walker + snapshot workers against the single sampler thread."""
import sys, time, numpy as np
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parent.parent))
from avlmaps_amd.map.vlmap_builder import VLMapBuilder
H, W, rate, n = 720, 1080, 100, 300
b = VLMapBuilder('/tmp', {}, None, [None]*n, [None]*n, np.eye(4), np.eye(4))
rgb, depth = np.zeros((H, W, 3), np.uint8), np.zeros((H, W), np.float32)
b.load_frame = lambda i: (rgb, depth)
b.prefetch_frames = 4
for w in (0, 1, 2, 3, 4, 0, 3):
    b.sampler_workers = w
    np.random.seed(1)
    t = time.perf_counter()
    k = sum(1 for _ in b._frame_stream(0, n, rate))
    dt = time.perf_counter() - t
    print(w, f"{k/dt:.0f} frames/s", {k_: round(v, 3) if isinstance(v, float) else v for k_, v in b.pipeline_stats.items()})
