"""Python face of the hot-path kernels (thin: argument marshalling only, all compute in HIP)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .device import DeviceArray, as_device, _is_torch

PRECISION = {"auto": _lib.SIM_AUTO, "exact": _lib.SIM_EXACT, "split_f16": _lib.SIM_SPLIT_F16}


def sim_scores(feat, queries, want_scores=True, want_argmax=True, want_best=False, precision="auto", stream=None,
               out_scores=None, out_argmax=None, out_best=None):
    """scores = feat @ queries.T (raw dot product, clip_utils.py:227-229) and argmax(axis=1) (vlmap.py:123).

    feat (N, D) float32, queries (Q, D) float32: numpy arrays, DeviceArrays or torch CUDA tensors.
    Returns (scores, argmax, best): numpy arrays for numpy inputs, otherwise device-side objects of
    the same kind as `feat` (entries are None when not requested).
    """
    lib = _lib.load()
    _lib.require_gpu()
    fptr, fshape, fkeep = as_device(feat, np.float32, stream)
    qptr, qshape, qkeep = as_device(queries, np.float32, stream)
    if len(fshape) != 2 or len(qshape) != 2 or fshape[1] != qshape[1]:
        raise ValueError(f"shape mismatch: feat {fshape}, queries {qshape}")
    N, D = fshape
    Q = qshape[0]
    torch_mode = _is_torch(feat)

    def alloc(shape, dtype, given):
        if given is not None:
            p, s, k = as_device(given, dtype, stream)
            if tuple(s) != tuple(shape):
                raise ValueError(f"output buffer shape {s} != {shape}")
            return p, k
        if torch_mode:
            import torch
            t = torch.empty(shape, dtype={np.float32: torch.float32, np.int32: torch.int32}[dtype], device=feat.device)
            return t.data_ptr(), t
        d = DeviceArray(shape, dtype)
        return d.ptr, d

    sp = sk = ap = ak = bp = bk = None
    if want_scores or out_scores is not None:
        sp, sk = alloc((N, Q), np.float32, out_scores)
    if want_argmax or out_argmax is not None:
        ap, ak = alloc((N,), np.int32, out_argmax)
    if want_best or out_best is not None:
        bp, bk = alloc((N,), np.float32, out_best)
    rc = lib.avl_sim_scores(fptr, N, D, D, qptr, Q, D, sp, ap, bp, PRECISION[precision], stream)
    _lib.check(rc, "avl_sim_scores")
    if isinstance(feat, np.ndarray):
        _lib.check(lib.avl_stream_sync(stream))
        res = tuple(k.numpy(stream) if k is not None else None for k in (sk, ak, bk))
        return res
    return sk, ak, bk


def mask_from_argmax(argmax, cat_id, stream=None):
    lib = _lib.load()
    ap, ashape, ak = as_device(argmax, np.int32, stream)
    N = ashape[0]
    if _is_torch(argmax):
        import torch
        m = torch.empty((N,), dtype=torch.uint8, device=argmax.device)
        _lib.check(lib.avl_mask_from_argmax(ap, N, int(cat_id), m.data_ptr(), stream), "avl_mask_from_argmax")
        return m.bool()
    m = DeviceArray((N,), np.uint8)
    _lib.check(lib.avl_mask_from_argmax(ap, N, int(cat_id), m.ptr, stream), "avl_mask_from_argmax")
    return m.numpy(stream).astype(bool) if isinstance(argmax, np.ndarray) else m


def argmax_f32(vals, stream=None):
    """(index, value) of the first maximum of a float32 vector (habitat_lang_robot.py:427-430)."""
    lib = _lib.load()
    vp, vshape, vk = as_device(vals, np.float32, stream)
    idx, val = C.c_int64(), C.c_float()
    _lib.check(lib.avl_argmax_f32(vp, int(np.prod(vshape)), C.byref(idx), C.byref(val), stream), "avl_argmax_f32")
    return idx.value, val.value
