#!/usr/bin/env python3
"""Where does the directory plan's time go?  One process, one GPU (or CPU with --cpu): n local voxels, every torch op of
plan_merge_directory / MixedExchange timed through the autograd profiler.  GPU box: python tools/probe_merge_plan.py [n]"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from avlmaps_amd import parallel  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1_250_000
    dev = "cpu" if "--cpu" in sys.argv else "cuda"
    rng = np.random.default_rng(0)
    cell = torch.from_numpy(rng.choice(30_000_000, n, replace=False).astype(np.int32)).to(dev)
    key = torch.from_numpy(((rng.integers(0, 1250, n).astype(np.int64)) << 32) | rng.permutation(n).astype(np.int64)).to(dev)
    for it in range(3):
        if dev == "cuda":
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        plan = parallel.plan_merge_directory(cell, key, grow_row=999_999)
        ex = parallel.MixedExchange(plan)
        if dev == "cuda":
            torch.cuda.synchronize()
        print(f"iter {it}: plan + exchange bookkeeping {1e3 * (time.perf_counter() - t0):.2f} ms (n = {n}, M = {plan.M})", flush=True)
    from torch.profiler import ProfilerActivity, profile
    acts = [ProfilerActivity.CPU] + ([ProfilerActivity.CUDA] if dev == "cuda" else [])
    with profile(activities=acts) as prof:
        plan = parallel.plan_merge_directory(cell, key, grow_row=999_999)
        ex = parallel.MixedExchange(plan)
        if dev == "cuda":
            torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=25))
    del ex


if __name__ == "__main__":
    main()
