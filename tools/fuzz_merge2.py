"""GPU box: random worlds through the gather-plan merge (time-boxed).  fuzz_merge2.py [seconds] [seed]
Every case: a random scene (frames, sample rate, cell size, feature width) built once by ONE accumulator and once sharded over a
random number of ranks (threads of this process, tests/thread_world.py); the HIP merge of the shards against the single-process map
(ids / order / weight / colour bit for bit, features to float64 summation order) and against the NumPy twin of the kernels (bit for bit)."""
import os
import sys
import time

import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from avlmaps_amd import _lib, ops  # noqa: E402
import test_merge2_gpu as T  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
_lib.load()
_lib.require_gpu()
t0, n, fails, nchunked = time.time(), 0, 0, 0
while time.time() - t0 < budget:
    ws = int(rng.integers(1, 9))
    cfg = dict(D=int(rng.choice([3, 5, 16, 30, 64, 256, 512, 768])), nfr=int(rng.integers(max(2, ws), 33)), seed=int(rng.integers(0, 1 << 30)),
               rate=int(rng.choice([1, 3, 5, 11])), cs=float(rng.choice([0.05, 0.1, 0.3])), gs=int(rng.choice([120, 400, 1000])))
    D = cfg.pop("D")
    if rng.random() < 0.5:
        cfg["chunk_rows"] = int(rng.choice([1, 17, 100, 1000, 5000]))          # the payload exchange in chunks of that many rows per owner
        nchunked += 1
    try:
        T.check_merge_world(ops, ws, D, **cfg)
    except Exception as e:      # noqa: BLE001
        fails += 1
        import traceback
        tb = traceback.extract_tb(e.__traceback__)[-1]
        print(f"FAIL ws={ws} D={D} {cfg}: {type(e).__name__}: {str(e)[:300]} at {tb.filename.split('/')[-1]}:{tb.lineno} `{tb.line}`", flush=True)
    n += 1
print(f"fuzz_merge2: {n} worlds (1-8 ranks, D 3-768; {nchunked} with a chunked exchange), {fails} failures, seed {seed}")
sys.exit(1 if fails else 0)
