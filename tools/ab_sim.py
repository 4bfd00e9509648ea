#!/usr/bin/env python3
"""Same-box A/B of library variants (tools/build_variant.py) on the index shapes.  GPU box only.

  python tools/ab_sim.py [--reps 2] [--shapes 2000000x512x64,...] stock noguard ring3
Each (variant, shape) is timed in its own subprocess (the library path is fixed at import), sustained launches after a settle
phase (HIP events on the launch stream), variants interleaved so that box drift averages out."""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def child(shape, mode):
    sys.path.insert(0, str(ROOT))
    import torch
    from avlmaps_amd import _lib
    sys.path.insert(0, str(ROOT))
    import bench
    lib = _lib.load()
    N, D, Q = shape
    feat, q = bench.make_index_inputs(torch, N, D, Q, seed=1234)
    am = torch.empty((N,), dtype=torch.int32, device="cuda")
    wsb = C.c_size_t()
    lib.avl_sim_workspace_bytes_n(N, D, Q, C.byref(wsb))
    ws = torch.empty((max(wsb.value, 64),), dtype=torch.uint8, device="cuda")
    from avlmaps_amd.ops import query_col_support
    cb, ce = query_col_support(q.cpu().numpy())
    blocks = D > 512 and mode.endswith("blocks")          # "rawblocks" / "preparedblocks" / "compactblocks": column-block launches
    base = mode.replace("blocks", "") or "raw"
    if base == "prepared":
        rs = torch.empty((N,), dtype=torch.float32, device="cuda")
        _lib.check(lib.avl_sim_prepare_map(feat.data_ptr(), N, D, D, rs.data_ptr(), None))
        if blocks:
            fn = lambda: _lib.check(lib.avl_sim_scores_blocks(feat.data_ptr(), rs.data_ptr(), N, D, D, q.data_ptr(), Q, D, cb.ctypes.data, ce.ctypes.data,
                                                              None, am.data_ptr(), None, _lib.SIM_PREPARED, ws.data_ptr(), wsb.value, None))
        else:
            fn = lambda: _lib.check(lib.avl_sim_scores_prepared(feat.data_ptr(), rs.data_ptr(), N, D, D, q.data_ptr(), Q, D, None, am.data_ptr(),
                                                                None, ws.data_ptr(), wsb.value, None))
    elif base == "compact":
        rs = torch.empty((N,), dtype=torch.float32, device="cuda")
        m24 = torch.empty((N, 3 * D), dtype=torch.uint8, device="cuda")
        _lib.check(lib.avl_sim_prepare_map24(feat.data_ptr(), N, D, D, m24.data_ptr(), rs.data_ptr(), None))
        torch.cuda.synchronize()
        del feat
        if blocks:
            fn = lambda: _lib.check(lib.avl_sim_scores_blocks(m24.data_ptr(), rs.data_ptr(), N, D, D, q.data_ptr(), Q, D, cb.ctypes.data, ce.ctypes.data,
                                                              None, am.data_ptr(), None, _lib.SIM_PREPARED24, ws.data_ptr(), wsb.value, None))
        else:
            fn = lambda: _lib.check(lib.avl_sim_scores_prepared24(m24.data_ptr(), rs.data_ptr(), N, D, q.data_ptr(), Q, D, None, am.data_ptr(),
                                                                  None, ws.data_ptr(), wsb.value, None))
    elif blocks:
        fn = lambda: _lib.check(lib.avl_sim_scores_blocks(feat.data_ptr(), None, N, D, D, q.data_ptr(), Q, D, cb.ctypes.data, ce.ctypes.data,
                                                          None, am.data_ptr(), None, 0, ws.data_ptr(), wsb.value, None))
    else:
        fn = lambda: _lib.check(lib.avl_sim_scores_ws(feat.data_ptr(), N, D, D, q.data_ptr(), Q, D, None, am.data_ptr(), None, 0,
                                                      ws.data_ptr(), wsb.value, None))
    ms = bench.sustained_ms(lib, fn, launches=300 if D <= 512 else 100, warm=120 if D <= 512 else 50)
    print(json.dumps(dict(ms=ms, gbs=N * D * 4 / ms / 1e6)))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        return child(tuple(int(x) for x in sys.argv[2].split("x")), sys.argv[3])
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--shapes", default="2000000x512x64,2000000x512x65,2000000x1536x128")
    ap.add_argument("--modes", default="raw")
    ap.add_argument("variants", nargs="+")
    a = ap.parse_args()
    res = {}
    for rep in range(a.reps):
        for shape in a.shapes.split(","):
            for mode in a.modes.split(","):
                for v in a.variants:
                    env = dict(os.environ)
                    if v.startswith("env:"):              # the stock library with an environment switch, e.g. env:AVL_SIM_PAIR=1
                        k, _, val = v[4:].partition("=")
                        env[k] = val or "1"
                    elif v != "stock":
                        env["AVLMAPS_HIP_LIB"] = str(ROOT / "variants" / f"libavlmaps_hip_{v}.so")
                    r = subprocess.run([sys.executable, __file__, "--child", shape, mode], env=env, capture_output=True, text=True, timeout=600)
                    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
                    if not line:
                        print(v, shape, mode, "FAILED", r.stderr[-400:])
                        continue
                    d = json.loads(line[-1])
                    res.setdefault((shape, mode, v), []).append(d["ms"])
                    print(f"rep{rep} {shape:>18s} {mode:9s} {v:12s} {d['ms']:.4f} ms  {d['gbs']:.0f} GB/s", flush=True)
    print("== mean")
    for (shape, mode, v), ms in res.items():
        print(f"{shape:>18s} {mode:9s} {v:12s} {sum(ms) / len(ms):.4f} ms  (n={len(ms)})")


if __name__ == "__main__":
    main()
