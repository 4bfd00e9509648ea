#!/usr/bin/env python3
"""Write a small synthetic scene in the reference's dataset layout (dataset/README.md:76-93):
<out>/rgb/%06d.png, <out>/depth/%06d.npy (float32 metres), <out>/poses.txt (x y z qx qy qz qw)."""
import argparse
from pathlib import Path

import numpy as np
from PIL import Image


def make(out, frames=8, H=120, W=160, seed=0):
    out = Path(out)
    (out / "rgb").mkdir(parents=True, exist_ok=True)
    (out / "depth").mkdir(exist_ok=True)
    rng = np.random.default_rng(seed)
    yy, xx = np.meshgrid(np.linspace(-1, 1, H), np.linspace(-1, 1, W), indexing="ij")
    poses = []
    for i in range(frames):
        d = 2.2 + 1.0 * np.sin(2.0 * xx + 0.2 * i) * np.cos(1.5 * yy) + 0.6 * yy
        np.save(out / "depth" / f"{i:06d}.npy", d.astype(np.float32))
        rgb = np.stack([(127 + 120 * np.sin(3 * xx + i)), (127 + 120 * np.cos(2 * yy)), 255 * (xx > 0)], -1)
        rgb = np.clip(rgb + rng.normal(0, 3, rgb.shape), 0, 255).astype(np.uint8)
        Image.fromarray(rgb).save(out / "rgb" / f"{i:06d}.png")
        yaw = 0.08 * i
        poses.append([0.1 * i, 0.0, -0.05 * i, 0.0, np.sin(yaw / 2), 0.0, np.cos(yaw / 2)])
    np.savetxt(out / "poses.txt", np.array(poses))
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--height", type=int, default=120)
    ap.add_argument("--width", type=int, default=160)
    a = ap.parse_args()
    print(make(a.out, a.frames, a.height, a.width))
