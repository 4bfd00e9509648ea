"""GPU box: randomised parity sweep (time-boxed).  fuzz_parity.py [seconds] [seed] [max cases per kind]
  sim:     random N / D / Q / row strides / precision modes / column windows / prepared maps against float64 NumPy
  builder: random frame shapes / grids / widths / sample rates, frame-by-frame vs deferred vs batched vs the C frame loop (plain / deferred, random pieces) vs the sequential oracle
  aux:     heat decay, argmax / top-k, pool_3d_label_to_2d, rgb top-down, obstacle map on random maps against the oracle (bit exact)
Prints one line per failure with the configuration that reproduces it; exit status 1 if anything failed."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(__file__), "..")))
from avlmaps_amd import _lib, ops  # noqa: E402
from avlmaps_amd.device import DeviceArray  # noqa: E402
from oracle import avl_oracle as O  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
max_cases = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 30
rng = np.random.default_rng(seed)
fails = []


def sim_case(i):
    N = int(rng.choice([1, 31, 33, 255, 257, 1000, 4097, 20011]))
    D = int(rng.choice([8, 60, 64, 128, 192, 320, 512, 640, 1024, 1536]))
    Q = int(rng.choice([1, 2, 7, 31, 32, 33, 64, 65, 78, 79, 96, 128, 129, 200]))
    f = rng.standard_normal((N, D)).astype(np.float32) * float(rng.choice([1e-3, 1.0, 14.0, 300.0]))
    q = (rng.standard_normal((Q, D)) / np.sqrt(D)).astype(np.float32)
    windows = None
    if D >= 256 and D % 128 == 0 and rng.random() < 0.5:            # block-structured queries
        cuts = sorted(set([0, D] + [int(c) for c in rng.choice(np.arange(128, D, 128), size=min(2, D // 128 - 1), replace=False)]))
        for qi in range(Q):
            k = int(rng.integers(0, len(cuts) - 1))
            lo, hi = cuts[k], cuts[k + 1]
            q[qi, :lo] = 0
            q[qi, hi:] = 0
        windows = "auto"
    if rng.random() < 0.2 and Q > 2:
        q[int(rng.integers(0, Q))] = 0
    if rng.random() < 0.2 and Q > 3:
        q[Q - 1] = q[0]                                              # exact tie: lowest index must win
    want = f.astype(np.float64) @ q.astype(np.float64).T
    tol = 2e-5 * max(1.0, float(np.abs(want).max()))
    # the split's own error scale: 2^-22 per product, summed over D products of random sign -- ~2.4e-7 |row| |query| / sqrt(D).  It only
    # exceeds the line above when every score of a case is small by cancellation (seen once: N = Q = 1, rows of norm 7 600, score 2.7,
    # error 7e-5 = 9e-9 of |row| |query|; NumPy's own float32 product is off by 1e-4 there)
    tol = max(tol, 1e-6 * float(np.linalg.norm(f, axis=1).max()) * float(np.linalg.norm(q, axis=1).max()) / np.sqrt(D))
    mode = str(rng.choice(["raw", "prepared", "prepared_unscaled", "exact", "compact"] if D % 128 == 0 else
                          (["raw", "prepared", "prepared_unscaled", "exact"] if D % 64 == 0 else ["raw", "exact"])))
    src = f
    kw = {}
    if mode == "prepared":
        src = ops.prepare_map(DeviceArray.from_numpy(f))
    elif mode == "prepared_unscaled":
        if float(np.abs(f).max()) > 6e4:
            return
        src = ops.prepare_map(DeviceArray.from_numpy(f), scaled=False)
    elif mode == "compact":                                          # 3-byte resident copy: the 1e-4 contract, relative to the row's size
        src = ops.prepare_map(DeviceArray.from_numpy(f), compact=True)
        tol = 1e-4 * max(1.0, float(np.linalg.norm(f, axis=1).max()) * float(np.linalg.norm(q, axis=1).max()) / 14.2857)
        windows = None
    elif mode == "exact":
        kw["precision"] = "exact"
    if windows is None and mode != "compact":
        kw["col_support"] = None
    if mode == "compact":
        kw = {}
    sc, am, best = ops.sim_scores(src, q, want_best=True, **kw)
    sc, am, best = (x.numpy() if not isinstance(x, np.ndarray) else x for x in (sc, am, best))
    cfg = dict(kind="sim", i=i, N=N, D=D, Q=Q, mode=mode, windows=windows)
    if mode == "prepared_unscaled":
        tol = max(tol, 1e-6 * float(np.abs(f).max()) * 4)           # rows far below 2^-7 lose relative precision unscaled
        tol = max(tol, 3e-4 if float(np.abs(f).max()) < 0.01 else tol)
    err = float(np.abs(sc - want).max())
    if not err < tol:
        fails.append((cfg, f"score error {err:.3e} > {tol:.3e}"))
    if not np.array_equal(am, np.argmax(sc, axis=1)):
        fails.append((cfg, "argmax != argmax(scores)"))
    if not np.array_equal(best, sc[np.arange(N), am]):
        fails.append((cfg, "best != scores[argmax]"))


def builder_case(i):
    H, W = int(rng.integers(8, 70)), int(rng.integers(8, 90))
    Hf, Wf = max(2, H // 2 - int(rng.integers(0, 3))), max(2, W // 2 - int(rng.integers(0, 3)))
    D = int(rng.choice([3, 8, 64, 100, 256, 300, 512, 700, 1200, 1536, 1600]))
    nfr = int(rng.integers(1, 7))
    rate = int(rng.choice([1, 2, 3, 7]))
    gs = int(rng.choice([20, 50, 200]))
    cs = float(rng.choice([0.05, 0.1, 0.33]))
    cam_h = 1.5
    calib = np.array([W / 2, 0, W / 2, 0, W / 2, H / 2, 0, 0, 1.0])
    yy, xx = np.meshgrid(np.linspace(-1, 1, H), np.linspace(-1, 1, W), indexing="ij")
    depths, rgbs, feats, poses = [], [], [], []
    for k in range(nfr):
        d = 2.0 + 1.5 * np.sin(2.0 * xx + 0.3 * k) * np.cos(1.5 * yy) + rng.normal(0, 0.05, xx.shape)
        d[rng.random(d.shape) < 0.05] = 0.0
        d[rng.random(d.shape) < 0.03] = 9.0
        depths.append(d.astype(np.float32))
        rgbs.append(rng.integers(0, 256, (H, W, 3), dtype=np.uint8))
        feats.append((rng.standard_normal((D, Hf, Wf)) * 3).astype(np.float32))
        yaw = 0.1 * k
        poses.append([0.1 * k, 0.0, -0.07 * k, 0.0, np.sin(yaw / 2), 0.0, np.cos(yaw / 2)])
    depths, rgbs, feats, poses = np.stack(depths), np.stack(rgbs), np.stack(feats), np.array(poses)
    b2c, bt = O.setup_transforms([1, 0, 0, 0, -1, 0, 0, 0, -1], cam_h, [0, 0, -1], [-1, 0, 0], [0, 1, 0])
    Ts = O.pc_transforms(poses, bt, b2c)
    rs = np.random.RandomState(int(rng.integers(0, 1 << 30)))
    samples = [O.sample_indices(rs, H * W, rate) for _ in range(nfr)]
    om = O.OracleMap(gs, cs, cam_h, D)
    pts = sum(om.integrate(depths[k], calib, Ts[k], samples[k], feats[k], rgbs[k]) for k in range(nfr))
    ref = om.export()
    cfg = dict(kind="builder", i=i, H=H, W=W, Hf=Hf, Wf=Wf, D=D, nfr=nfr, rate=rate, gs=gs, cs=cs)
    vh = int(cam_h / cs)
    fs = [np.ascontiguousarray(np.transpose(f, (1, 2, 0))) for f in feats]
    outs = {}
    for mode in ("frames", "deferred", "batch", "loop", "loop-deferred"):
        acc = ops.VoxelAccumulator(gs, cs, vh, D, capacity=int(rng.choice([4, 64, 5000])), deferred_fuse=mode.endswith("deferred"))
        acc.enable_replay_log(sum(len(s) for s in samples))
        if mode == "batch":
            k = 0
            while k < nfr:
                b = int(rng.integers(1, nfr - k + 1))
                sl = slice(k, k + b)
                acc.integrate_batch(list(depths[sl]), calib, Ts[sl], samples[sl], fs[sl], list(rgbs[sl]), frame_idx0=k)
                k += b
        elif mode.startswith("loop"):       # the frame loop in C, in random pieces (a frame's launch prepares the next frame's K1)
            k = 0
            while k < nfr:
                b = int(rng.integers(1, nfr - k + 1))
                sl = slice(k, k + b)
                if all(len(samples[j]) == len(samples[k]) for j in range(k, k + b)) and len(samples[k]) > 0:
                    plan = acc.make_batch_plan(list(depths[sl]), samples[sl], fs[sl], list(rgbs[sl]))
                    acc.integrate_frames(plan, calib, Ts[sl], frame_idx0=k)
                else:
                    for j in range(k, k + b):
                        acc.integrate_frame(depths[j], calib, Ts[j], samples[j], fs[j], rgbs[j], frame_idx=j)
                k += b
        else:
            for k in range(nfr):
                acc.integrate_frame(depths[k], calib, Ts[k], samples[k], fs[k], rgbs[k], frame_idx=k)
        try:
            nv, npt = acc.num_voxels(), acc.num_points()
        except Exception as e:       # e.g. the RGB projection left the image: the reference raises IndexError there too
            try:
                om_err = None
            finally:
                pass
            fails.append((dict(cfg, mode=mode), f"library error: {e}")) if "IndexError" not in str(e) and "outside the RGB image" not in str(e) else None
            acc.close()
            return
        out = outs[mode] = acc.finalize()
        acc.close()
        c = dict(cfg, mode=mode)
        if nv != len(ref["grid_pos"]) or npt != pts:
            fails.append((c, f"counts {nv}/{npt} != {len(ref['grid_pos'])}/{pts}"))
            return
        if nv == 0:
            continue
        if not (np.array_equal(out["grid_pos"], ref["grid_pos"]) and np.array_equal(out["occupied_ids"], ref["occupied_ids"])):
            fails.append((c, "voxel ids differ"))
        drgb = np.abs(out["grid_rgb"].astype(int) - ref["grid_rgb"].astype(int))
        # the reference's own knife edge (profiles/HISTORY.md 4.3): when the incoming colour equals the stored one, (c w + c a) / (w + a) is c or
        # c - 1 ulp depending on the last bit of exp(); the device and host exp differ by an ulp now and then -> at most 1 LSB, in at
        # most 1 % of the bytes -- or the three channels of ONE voxel of a map of a few dozen voxels (dense sampling into a 20-cell grid)
        if drgb.max() > 1 or (drgb != 0).sum() > max(3, 0.01 * drgb.size):      # (3: the channels of one voxel)
            fails.append((c, f"rgb differs: max {drgb.max()}, frac {(drgb != 0).mean():.4f}"))
        if not np.allclose(out["weight"], ref["weight"].astype(np.float32), rtol=3e-6, atol=0):
            fails.append((c, "weight differs"))
        if not np.allclose(out["grid_feat"], ref["grid_feat"], rtol=1e-4, atol=1e-4 * max(1.0, float(np.abs(ref["grid_feat"]).max()))):
            fails.append((c, f"grid_feat differs by {np.abs(out['grid_feat'] - ref['grid_feat']).max():.3e}"))
    if "frames" in outs and "deferred" in outs and len(ref["grid_pos"]):
        for k in ("grid_pos", "occupied_ids", "grid_rgb", "weight"):
            if not np.array_equal(outs["frames"][k], outs["deferred"][k]):
                fails.append((dict(cfg, mode="deferred-vs-frames"), f"{k} not identical"))
        fa, fb = outs["frames"]["grid_feat"], outs["deferred"]["grid_feat"]      # equal up to the summation order inside a list
        if not (np.mean(fa == fb) > 0.999 and np.allclose(fa, fb, rtol=1e-6, atol=1e-6 * max(1.0, float(np.abs(fa).max())))):
            fails.append((dict(cfg, mode="deferred-vs-frames"), "grid_feat differs beyond fp64 summation order"))
    for m in ("loop", "loop-deferred"):         # the C frame loop issues the same launches as single calls: everything bit for bit
        if m in outs and "frames" in outs and len(ref["grid_pos"]):
            for k in ("grid_pos", "occupied_ids", "grid_rgb", "weight", "grid_feat"):
                if not np.array_equal(outs[m][k], outs["frames"][k]):     # (incl. the generic-width kernel, D > 1536: ordered sums since round 6)
                    fails.append((dict(cfg, mode=m + "-vs-frames"), f"{k} not identical"))


def aux_case(i):
    gs = int(rng.choice([16, 50, 128, 300]))
    vh = int(rng.choice([4, 12, 30]))
    ncell = gs * gs * vh
    N = int(min(ncell, rng.choice([1, 5, 64, 1000, 20000])))
    cells = rng.choice(ncell, size=N, replace=False)
    pos = np.stack([cells // (gs * vh), (cells // vh) % gs, cells % vh], 1).astype(np.int32)
    cfg = dict(kind="aux", i=i, gs=gs, vh=vh, N=N)
    mask = rng.random(N) < float(rng.choice([0.001, 0.05, 0.5] if N <= 1000 else [0.001, 0.02]))   # the oracle's heat is O(N * targets)
    if not mask.any():
        mask[int(rng.integers(0, N))] = True
    cs, decay = float(rng.choice([0.05, 0.1])), float(rng.choice([0.01, 0.05, 0.3]))
    heat = ops.heatmap_from_mask(pos, mask, cs, decay)
    heat = heat.numpy() if not isinstance(heat, np.ndarray) else heat
    if not np.array_equal(heat, O.heatmap_from_mask(pos, mask, cs, decay)):
        fails.append((cfg, "heat differs"))
    plan = ops.HeatPlan(pos)                       # the map's cached cell order must give the same bits, also on a second mask
    mask2 = np.roll(mask, 1)
    for mk in (mask, mask2):
        if not np.array_equal(plan(mk, cs, decay).numpy(), heat if mk is mask else ops.heatmap_from_mask(pos, mk, cs, decay)):
            fails.append((cfg, "planned heat differs"))
    plan.close()
    idx, val = ops.argmax_f32(heat)
    if idx != int(np.argmax(heat)) or val != heat[idx]:
        fails.append((cfg, "argmax_f32 differs"))
    vals = rng.standard_normal(N).astype(np.float32)
    vals[rng.integers(0, N, size=max(1, N // 10))] = vals[0]                  # ties
    k = int(min(N, rng.choice([1, 3, 17, 64, 100])))
    ti, tv = ops.topk_f32(vals, k)
    order = np.argsort(-vals, kind="stable")[:k]
    if not (np.array_equal(np.asarray(ti), order) and np.array_equal(np.asarray(tv), vals[order])):
        fails.append((cfg, f"topk k={k} differs"))
    if not np.array_equal(np.asarray(ops.pool_label_2d(mask, pos, gs)), O.pool_3d_label_to_2d(mask, pos, gs)):
        fails.append((cfg, "pool_label_2d differs"))
    rgb = rng.integers(0, 256, (N, 3), dtype=np.uint8)
    if not np.array_equal(np.asarray(ops.rgb_topdown(pos, rgb, gs)), O.rgb_topdown(pos, rgb, gs)):
        fails.append((cfg, "rgb_topdown differs"))
    occ = -np.ones((gs, gs, vh), np.int32)
    occ[pos[:, 0], pos[:, 1], pos[:, 2]] = np.arange(N, dtype=np.int32)
    h_min, h_max = float(rng.choice([0.0, 0.1, -1.0])), float(rng.choice([0.3, 1.5, 100.0]))
    if not np.array_equal(np.asarray(ops.obstacle_map(occ, cs, h_min, h_max)), O.obstacle_map(occ, cs, h_min, h_max)):
        fails.append((cfg, "obstacle_map differs"))


t0 = time.time()
n_sim = n_b = n_aux = 0
while time.time() - t0 < budget and n_b < max_cases:
    before = len(fails)
    which = (n_sim + n_b + n_aux) % 3
    if which == 0:
        sim_case(n_sim)
        n_sim += 1
    elif which == 1:
        builder_case(n_b)
        n_b += 1
    else:
        aux_case(n_aux)
        n_aux += 1
    for cfg, msg in fails[before:]:
        print("FAIL", cfg, msg, flush=True)
print(f"fuzz: {n_sim} similarity cases, {n_b} builder cases, {n_aux} auxiliary cases, {len(fails)} failures, seed {seed}")
sys.exit(1 if fails else 0)
