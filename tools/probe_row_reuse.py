"""How often does a frame of BASELINE's build revisit accumulator rows that could still sit in the eight 4 MB L2s?  (CPU, NumPy.)
The build workload of bench.py: 720x1080 depth surfaces, 7 776 random pixels per frame, the loop trajectory.  Per frame: the
distinct voxels hit (K3's groups), the share of them also hit in the previous frame, and the hit rate of an LRU set of 8 000 rows
(32 MB / 4 KB) -- the best an XCD-affine, L2-resident accumulator scheme could get."""
import collections
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402


def main(frames=300, rows=8000):
    H, W, rate, nbuf = 720, 1080, 100, 4
    rng = np.random.default_rng(0)
    yy, xx = np.meshgrid(np.linspace(-1, 1, H), np.linspace(-1, 1, W), indexing="ij")
    depths = [(2.6 + 1.6 * np.sin(2.0 * xx + 0.37 * i) * np.cos(1.5 * yy) + 0.7 * yy + 0.02 * rng.standard_normal((H, W))) for i in range(nbuf)]
    rs = np.random.RandomState(5)
    samples = []
    for _ in range(nbuf):
        m = np.arange(H * W)
        rs.shuffle(m)
        samples.append(m[::rate])
    Ts = bench.pc_transforms(bench.trajectory(5000 + frames, "loop"))
    K = np.array([[540, 0, 540], [0, 540, 360], [0, 0, 1.0]])
    Kinv = np.linalg.inv(K)
    gs, cs, vh = 1000, 0.05, 30
    lru = collections.OrderedDict()
    prev = set()
    st = []
    for f in range(5000, 5000 + frames):
        b = f % nbuf
        pix = samples[b]
        u, v = pix % W + 0.5, pix // W + 0.5
        z = depths[b].reshape(-1)[pix]
        p = (Kinv @ np.stack([u, v, np.ones_like(u)])) * z
        ok = (p[2] > 0.1) & (p[2] < 6.0)
        g = Ts[f] @ np.vstack([p, np.ones(len(u))])
        row = (gs / 2 - np.trunc(g[0] / cs)).astype(np.int64)
        col = (gs / 2 - np.trunc(g[1] / cs)).astype(np.int64)
        h = np.trunc(g[2] / cs).astype(np.int64)
        ok &= (row >= 0) & (row < gs) & (col >= 0) & (col < gs) & (h >= 0) & (h < vh)
        cells = set(((row * gs + col) * vh + h)[ok].tolist())
        hit = 0
        for c in cells:
            if c in lru:
                hit += 1
                lru.move_to_end(c)
            else:
                lru[c] = 1
                if len(lru) > rows:
                    lru.popitem(last=False)
        if f >= 5020:
            st.append((len(cells), len(cells & prev) / max(1, len(cells)), hit / max(1, len(cells))))
        prev = cells
    st = np.array(st)
    print(f"{len(st)} frames of the loop trajectory (frames 5020..): {st[:, 0].mean():.0f} voxel groups per frame; {100 * st[:, 1].mean():.1f} % of them were also hit "
          f"by the previous frame; an LRU set of {rows} rows (= the eight L2s, 32 MB of 4 KB rows) would hold {100 * st[:, 2].mean():.1f} % of a frame's rows")


if __name__ == "__main__":
    main()
