#!/usr/bin/env python3
"""Samples package power and sclk from sysfs while one kernel runs in a loop (GPU box only): is a kernel power-capped?"""
import ctypes as C
import glob
import sys
import threading
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from avlmaps_amd import _lib  # noqa: E402


def read(p):
    try:
        return Path(p).read_text().strip()
    except Exception:
        return None


CARD = None


def pick_card():
    """the hwmon directory of the GPU this process sees (matched through its PCI bus id)"""
    import os
    try:
        bus = torch.cuda.get_device_properties(0).pci_bus_id
        want = f"{int(bus):02x}:" if isinstance(bus, int) else str(bus).lower()
    except Exception:
        want = None
    cands = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
    for h in cands:
        real = os.path.realpath(h).lower()
        if want and want in real:
            return h
    return None


def sensors():
    out = {}
    hs = [CARD] if CARD else sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
    for i, h in enumerate(hs):
        for name in ("power1_input", "power1_cap", "freq1_input", "freq2_input", "temp2_input"):
            v = read(f"{h}/{name}")
            if v is not None:
                out[name if CARD else f"{name}@{i}"] = int(v)
    return out


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
    lib = _lib.load()
    import os
    N, D, Q = 2_000_000, int(os.environ.get('PP_D', 512)), int(os.environ.get('PP_Q', 64))
    feat = torch.randn((N, D), device="cuda")
    prep = feat.clone()
    assert lib.avl_sim_prepare_map(prep.data_ptr(), N, D, D, None, None) == 0   # unscaled: bit-identical to the raw split
    q = torch.randn((Q, D), device="cuda"); q /= q.norm(dim=1, keepdim=True)
    am = torch.empty((N,), dtype=torch.int32, device="cuda")
    wsb = C.c_size_t(); lib.avl_sim_workspace_bytes(D, Q, C.byref(wsb))
    ws = torch.empty((wsb.value,), dtype=torch.uint8, device="cuda")
    g = C.c_float()

    def k_split():
        lib.avl_sim_scores_ws(feat.data_ptr(), N, D, D, q.data_ptr(), Q, D, None, am.data_ptr(), None, 2, ws.data_ptr(), wsb.value, None)

    def k_prep():
        lib.avl_sim_scores_ws(prep.data_ptr(), N, D, D, q.data_ptr(), Q, D, None, am.data_ptr(), None, 4, ws.data_ptr(), wsb.value, None)

    def k_exact_mfma():
        lib.avl_sim_scores_ws(feat.data_ptr(), N, D, D, q.data_ptr(), Q, D, None, am.data_ptr(), None, 1, ws.data_ptr(), wsb.value, None)

    def k_probe_row():
        lib.avl_hbm_read_probe(feat.data_ptr(), N, D, 3, 20, C.byref(g), None)   # 40 back-to-back launches

    def k_probe_coal():
        lib.avl_hbm_read_probe(feat.data_ptr(), N, D, 2, 20, C.byref(g), None)

    global CARD
    import os
    print("pci", torch.cuda.get_device_properties(0).pci_bus_id if hasattr(torch.cuda.get_device_properties(0), "pci_bus_id") else None,
          os.environ.get("HIP_VISIBLE_DEVICES"), os.environ.get("ROCR_VISIBLE_DEVICES"))
    for h in sorted(glob.glob("/sys/class/drm/card*/device")):
        print(h, os.path.realpath(h))
    CARD = pick_card()
    print("card:", CARD)
    print("idle", sensors())
    import os
    only = os.environ.get("PP_KERNELS")
    for name, fn, per in (("split_f16", k_split, 1), ("prepared", k_prep, 1), ("exact_f32_mfma", k_exact_mfma, 1),
                          ("probe_rowline", k_probe_row, 40), ("probe_coalesced", k_probe_coal, 40)):
        if only and name not in only.split(","):
            continue
        samples = []
        stop = False

        def sampler():
            while not stop:
                samples.append(sensors())
                time.sleep(0.02)
        th = threading.Thread(target=sampler)
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        th.start()
        t0 = time.time(); n = 0
        while time.time() - t0 < secs:
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            n += 20
        dt = time.time() - t0
        stop = True
        th.join()
        half = samples[len(samples) // 2:]
        keys = sorted({k for smp in half for k in smp})
        line = "  ".join(f"{k}={np.mean([smp[k] for smp in half if k in smp]) / (1e3 if k.startswith('temp') else 1e6):.0f}" for k in keys
                         if not k.startswith("power1_cap"))
        print(f"{name:16s} {dt / (n * per) * 1e3:7.3f} ms/launch  {line}  n_samples={len(samples)}")
        time.sleep(0.5)


if __name__ == "__main__":
    main()
