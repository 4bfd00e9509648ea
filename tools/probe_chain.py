"""GPU box: where a frame-by-frame builder launch spends its time.  Needs the instrumented library
(python tools/build_variant.py probe --src avl_builder.hip -DAVL_PROBE_CHAIN; AVLMAPS_HIP_LIB=variants/libavlmaps_hip_probe.so):
every work item of ONE launch stamps s_memrealtime (100 MHz) after each hop of its dependent-load chain.
usage: probe_chain.py [frames_before=1500] [seq]   (seq: the probed launch is the third frame of an avl_builder_integrate_frames call)"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import bench  # noqa: E402
from avlmaps_amd import _lib, ops  # noqa: E402

H, W, Hf, Wf, D, rate = 720, 1080, 347, 520, 512, 100
nbuf = 4
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
seq = len(sys.argv) > 2 and sys.argv[2] == "seq"
depths, rgbs, feats = bench.make_build_inputs(torch, H, W, Hf, Wf, D, nbuf, seed=99)
Ts = bench.pc_transforms(bench.trajectory(n + 16))
calib = np.array([540, 0, 540, 0, 540, 360, 0, 0, 1.0])
rs = np.random.RandomState(5)
samples = []
for _ in range(nbuf):
    m = np.arange(H * W)
    rs.shuffle(m)
    samples.append(torch.from_numpy(m[::rate].astype(np.int32)).cuda())
lib = _lib.load()
setp = lib.avl_debug_set_probe
setp.restype, setp.argtypes = C.c_int, [C.c_void_p]
K12 = 16384
P = samples[0].numel()


def pct(a, name):
    a = np.asarray(a, dtype=np.float64) * 10.0   # ticks of 10 ns -> ns
    if a.size == 0:
        print(f"  {name:34s} (none)")
        return
    print(f"  {name:34s} n={a.size:5d}  p50 {np.percentile(a, 50):7.0f}  p90 {np.percentile(a, 90):7.0f}  p99 {np.percentile(a, 99):7.0f}  max {a.max():7.0f} ns")


for deferred in (False, True):
    acc = ops.VoxelAccumulator(1000, 0.05, 30, D, capacity=2_500_000, deferred_fuse=deferred)
    for i in range(n):
        b = i % nbuf
        acc.integrate_frame(depths[b], calib, Ts[i], samples[b], feats[b], rgbs[b], frame_idx=i)
    torch.cuda.synchronize()
    for rep in range(3):
        buf = torch.zeros((K12 + 8192) * 8, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        assert setp(buf.data_ptr()) == 0
        i = n + rep
        b = i % nbuf
        if seq:     # the C frame loop: three frames in one call, the stamps that remain are the LAST launch's (K1 from prepared records)
            idx = [(i + k) % nbuf for k in range(3)]
            plan = acc.make_batch_plan([depths[j] for j in idx], [samples[j] for j in idx], [feats[j] for j in idx], [rgbs[j] for j in idx])
            acc.integrate_frames(plan, calib, Ts[i:i + 3], frame_idx0=i)
        else:
            acc.integrate_frame(depths[b], calib, Ts[i], samples[b], feats[b], rgbs[b], frame_idx=i)
        torch.cuda.synchronize()
        assert setp(None) == 0
        t = buf.cpu().numpy().astype(np.uint64).reshape(-1, 8)
        k3, k12 = t[:P], t[K12:K12 + P]
        print(f"=== deferred={deferred} rep {rep}: one launch ({'pipe_kernel' if deferred else 'voxelize_link + fuse'})")
        live12 = k12[:, 0] != 0
        a = k12[live12]
        start = a[:, 0].min()
        end12 = (a[:, 7] & np.uint64((1 << 63) - 1)).max()
        print(f" K1+K2: {live12.sum()} samples; span first start -> last end {10.0 * float(end12 - start):.0f} ns")
        pct(a[:, 0] - start, "start skew")
        pct(a[:, 1] - a[:, 0], "hop: kernel arguments + sample index")
        pct(a[:, 2] - a[:, 1], "hop: depth")
        pct(a[:, 3] - a[:, 2], "geometry (fp64 divides, exp)")
        pct(a[:, 4] - a[:, 3], "hop: colour + cell_slot (+ CAS)")
        pct(a[:, 5] - a[:, 4], "counter atomic + publish + rec stores")
        e = a[:, 7] & np.uint64((1 << 63) - 1)
        pct(e - a[:, 5], "K2: poll + exch + stores")
        pct(e - a[:, 0], "whole chain")
        live3 = k3[:, 0] != 0
        c = k3[live3]
        own = c[:, 7] != 0
        o = c[own]
        if len(c):
            s3 = c[:, 0].min()
            e3 = max(o[:, 6].max() if len(o) else 0, c[~own][:, 1].max() if (~own).any() else 0)
            print(f" K3: {live3.sum()} waves, {own.sum()} owners; span {10.0 * float(e3 - s3):.0f} ns" + (f"; K3 start - K12 start {10.0 * (float(s3) - float(start)):.0f} ns"))
            pct(c[:, 0] - s3, "start skew (all waves)")
            pct(c[~own][:, 1] - c[~own][:, 0], "non-owner: record load -> exit")
            nm = (o[:, 7] & np.uint64(0xffffffff)).astype(np.int64)
            new = (o[:, 7] >> np.uint64(32)) != 0
            print(f"  members per group: " + " ".join(f"{k}:{(nm == k).sum()}" for k in range(1, min(nm.max(), 12) + 1)) + f"  max {nm.max()}; new voxels {new.sum()}")
            pct(o[:, 1] - o[:, 0], "hop: own record (s_load)")
            pct(o[:, 2] - o[:, 1], "hop: slot_key + head")
            for k, lab in ((1, "single"), (2, "2 members"), (3, "3 members")):
                g = o[nm == k]
                pct(g[:, 3] - g[:, 2], f"[{lab}] rows / list walk")
                pct(g[:, 4] - g[:, 3], f"[{lab}] member records")
                pct(g[:, 5] - g[:, 4], f"[{lab}] member rows + adds")
                pct(g[:, 6] - g[:, 5], f"[{lab}] stores")
                pct(g[:, 6] - g[:, 0], f"[{lab}] whole chain")
            g = o[nm >= 4]
            pct(g[:, 6] - g[:, 0], "[>= 4 members] whole chain")
            pct(o[:, 6] - s3, "owner end since K3 start")
    acc.close()
