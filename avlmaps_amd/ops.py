"""Python face of the hot-path kernels (thin: argument marshalling only, all compute in HIP)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .device import DeviceArray, as_device, _is_torch

PRECISION = {"auto": _lib.SIM_AUTO, "exact": _lib.SIM_EXACT, "split_f16": _lib.SIM_SPLIT_F16, "exact_valu": _lib.SIM_EXACT_VALU,
             "prepared": _lib.SIM_PREPARED}


def query_col_support(queries):
    """(begin, end) int32 arrays: the non-zero column window of every row of a HOST query matrix"""
    q = np.asarray(queries)
    nz = q != 0
    any_nz = nz.any(axis=1)
    begin = np.where(any_nz, nz.argmax(axis=1), 0).astype(np.int32)
    end = np.where(any_nz, q.shape[1] - nz[:, ::-1].argmax(axis=1), 0).astype(np.int32)
    return begin, end


def sim_scores(feat, queries, want_scores=True, want_argmax=True, want_best=False, precision="auto", stream=None,
               out_scores=None, out_argmax=None, out_best=None, col_support="auto"):
    """scores = feat @ queries.T (raw dot product, clip_utils.py:227-229) and argmax(axis=1) (vlmap.py:123).

    feat (N, D) float32, queries (Q, D) float32: numpy arrays, DeviceArrays or torch CUDA tensors.
    Returns (scores, argmax, best): numpy arrays for numpy inputs, otherwise device-side objects of
    the same kind as `feat` (entries are None when not requested).
    col_support: (begin, end) per-query non-zero column windows for block-structured query sets (e.g. text queries living
    in the visual columns and audio queries in the audio columns of a fused map): queries are then scored against their own
    columns only (avl_sim_scores_blocks).  "auto" derives the windows when `queries` is a host array; None = dense.
    """
    lib = _lib.load()
    _lib.require_gpu()
    prepared = None
    if isinstance(feat, PreparedMap):
        prepared, feat, precision = feat, feat.feat, "prepared"
        if prepared.compact:
            return _sim_scores_compact(lib, prepared, queries, want_scores, want_argmax, want_best, stream, out_scores, out_argmax, out_best,
                                       col_support)
    if stream is None and _is_torch(feat):
        from .device import torch_stream_ptr
        stream = torch_stream_ptr()           # launch on torch's current stream so torch-side ordering holds
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
    rsp = None
    if prepared is not None and prepared.row_scale is not None:
        rs = prepared.row_scale
        rsp = rs.data_ptr() if _is_torch(rs) else rs.ptr
    if isinstance(col_support, str):
        col_support = query_col_support(queries) if (col_support == "auto" and isinstance(queries, np.ndarray) and D > 128) else None
    if col_support is not None and precision in ("auto", "split_f16", "prepared"):
        cb = np.ascontiguousarray(col_support[0], dtype=np.int32)
        ce = np.ascontiguousarray(col_support[1], dtype=np.int32)
        if cb.shape != (Q,) or ce.shape != (Q,):
            raise ValueError("col_support must be two (Q,) integer arrays")
        wsp, wsb = _sim_workspace(lib, N, D, Q, stream)
        rc = lib.avl_sim_scores_blocks(fptr, rsp, N, D, D, qptr, Q, D, cb.ctypes.data, ce.ctypes.data, sp, ap, bp, PRECISION[precision],
                                       wsp, wsb, stream)
    elif rsp is not None:
        wsp, wsb = _sim_workspace(lib, N, D, Q, stream)
        rc = lib.avl_sim_scores_prepared(fptr, rsp, N, D, D, qptr, Q, D, sp, ap, bp, wsp, wsb, stream)
    else:
        wsp, wsb = _sim_workspace(lib, N, D, Q, stream)
        rc = lib.avl_sim_scores_ws(fptr, N, D, D, qptr, Q, D, sp, ap, bp, PRECISION[precision], wsp, wsb, stream)
    _lib.check(rc, "avl_sim_scores")
    if isinstance(feat, np.ndarray):
        _lib.check(lib.avl_stream_sync(stream))
        res = tuple(k.numpy(stream) if k is not None else None for k in (sk, ak, bk))
        return res
    return sk, ak, bk


_WS_CACHE: dict = {}


def _sim_workspace(lib, N, D, Q, stream):
    """per-stream scratch of the similarity kernels (query image + range-guard words + column-block gather), grown on demand
    and reused: the library then allocates nothing per call (avl_sim_workspace_bytes_n)"""
    need = C.c_size_t()
    _lib.check(lib.avl_sim_workspace_bytes_n(int(N), int(D), int(Q), C.byref(need)), "avl_sim_workspace_bytes_n")
    key = int(stream or 0)
    ws = _WS_CACHE.get(key)
    if ws is None or ws.nbytes < need.value:
        ws = _WS_CACHE[key] = DeviceArray((max(need.value, 1 << 20),), np.uint8)
    return ws.ptr, ws.nbytes


class PreparedMap:
    """a device-resident map converted in place by avl_sim_prepare_map: `feat` (N, D) now holds fp16 hi | lo groups,
    `row_scale` (N,) float32 the per-row 2^-s (None: prepared without scaling).  Pass it to sim_scores as `feat`.
    compact=True: `feat` is the separate (N, 3 D) uint8 buffer of avl_sim_prepare_map24 (fp16 hi + one residual byte per element)."""
    __slots__ = ("feat", "row_scale", "shape", "compact")

    def __init__(self, feat, row_scale, shape, compact=False):
        self.feat, self.row_scale, self.shape, self.compact = feat, row_scale, tuple(shape), bool(compact)


def _sim_scores_compact(lib, pm, queries, want_scores, want_argmax, want_best, stream, out_scores, out_argmax, out_best, col_support="auto"):
    """sim_scores on a compact prepared map (avl_sim_scores_prepared24, or avl_sim_scores_blocks with AVL_SIM_PREPARED24 for
    block-structured query sets); results as DeviceArrays / torch tensors like the map"""
    torch_mode = _is_torch(pm.feat)
    if stream is None and torch_mode:
        from .device import torch_stream_ptr
        stream = torch_stream_ptr()
    N, D = pm.shape
    qptr, qshape, qkeep = as_device(queries, np.float32, stream)
    if len(qshape) != 2 or qshape[1] != D:
        raise ValueError(f"shape mismatch: map {pm.shape}, queries {qshape}")
    Q = qshape[0]

    def alloc(shape, dtype, given):
        if given is not None:
            p, s, k = as_device(given, dtype, stream)
            if tuple(s) != tuple(shape):
                raise ValueError(f"output buffer shape {s} != {shape}")
            return p, k
        if torch_mode:
            import torch
            t = torch.empty(shape, dtype={np.float32: torch.float32, np.int32: torch.int32}[dtype], device=pm.feat.device)
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
    fptr = pm.feat.data_ptr() if torch_mode else pm.feat.ptr
    rsp = pm.row_scale.data_ptr() if _is_torch(pm.row_scale) else pm.row_scale.ptr
    wsp, wsb = _sim_workspace(lib, N, D, Q, stream)
    if isinstance(col_support, str):
        col_support = query_col_support(queries) if (col_support == "auto" and isinstance(queries, np.ndarray) and D > 128) else None
    if col_support is not None:
        cb = np.ascontiguousarray(col_support[0], dtype=np.int32)
        ce = np.ascontiguousarray(col_support[1], dtype=np.int32)
        if cb.shape != (Q,) or ce.shape != (Q,):
            raise ValueError("col_support must be two (Q,) integer arrays")
        _lib.check(lib.avl_sim_scores_blocks(fptr, rsp, N, D, D, qptr, Q, D, cb.ctypes.data, ce.ctypes.data, sp, ap, bp, _lib.SIM_PREPARED24,
                                             wsp, wsb, stream), "avl_sim_scores_blocks")
    else:
        _lib.check(lib.avl_sim_scores_prepared24(fptr, rsp, N, D, qptr, Q, D, sp, ap, bp, wsp, wsb, stream), "avl_sim_scores_prepared24")
    return sk, ak, bk


def prepare_map(feat_dev, scaled=True, stream=None, compact=False):
    """Convert a DEVICE-resident float32 map in place into the split-fp16 layout (avl_sim_prepare_map) and return a
    PreparedMap for sim_scores.  scaled=True (default): every row gets its own power-of-two scale, so rows of any magnitude
    -- e.g. voxels observed once from far away, feat * exp(-r^2/1.2) -- score with float32-class accuracy.  scaled=False:
    scores bit-identical to the on-the-fly split of the raw map.  feat_dev: DeviceArray or torch CUDA tensor (N, D), D % 64 == 0.
    compact=True (D % 128 == 0): out of place into the 3-byte form (avl_sim_prepare_map24: a plane of fp16 hi values + a plane of
    residual bytes in units of ulp(hi)/256 per row, always row-scaled): a quarter less HBM traffic per query pass, max score error
    2.3e-6 instead of 1.4e-6 (float32-class); the float32 map is left untouched and can be freed by the caller."""
    lib = _lib.load()
    if isinstance(feat_dev, np.ndarray):
        raise TypeError("prepare_map works on a device-resident map (DeviceArray / torch CUDA tensor), not a host array")
    if stream is None and _is_torch(feat_dev):
        from .device import torch_stream_ptr
        stream = torch_stream_ptr()
    fptr, fshape, _ = as_device(feat_dev, np.float32, stream)
    if compact:
        N, D = fshape
        if D % 128:
            raise ValueError(f"the compact form needs a feature width that is a multiple of 128 (D = {D}); use the 4-byte prepared form")
        if _is_torch(feat_dev):
            import torch
            buf = torch.empty((N, 3 * D), dtype=torch.uint8, device=feat_dev.device)
            rs = torch.empty((N,), dtype=torch.float32, device=feat_dev.device)
            bptr, rptr = buf.data_ptr(), rs.data_ptr()
        else:
            buf, rs = DeviceArray((N, 3 * D), np.uint8), DeviceArray((N,), np.float32)
            bptr, rptr = buf.ptr, rs.ptr
        _lib.check(lib.avl_sim_prepare_map24(fptr, N, D, D, bptr, rptr, stream), "avl_sim_prepare_map24")
        return PreparedMap(buf, rs, fshape, compact=True)
    rs = None
    if scaled:
        if _is_torch(feat_dev):
            import torch
            rs = torch.empty((fshape[0],), dtype=torch.float32, device=feat_dev.device)
        else:
            rs = DeviceArray((fshape[0],), np.float32)
    rptr = None if rs is None else (rs.data_ptr() if _is_torch(rs) else rs.ptr)
    _lib.check(lib.avl_sim_prepare_map(fptr, fshape[0], fshape[1], fshape[1], rptr, stream), "avl_sim_prepare_map")
    return PreparedMap(feat_dev, rs, fshape)


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


def mask_bool_from_argmax(argmax, cat_id, stream=None) -> np.ndarray:
    """host bool mask (N,) = (argmax == cat_id) of a DEVICE-resident argmax (DeviceArray / torch tensor): the comparison and a
    64-to-1 bit packing run on the GPU (avl_mask_bits_from_argmax), N / 8 bytes cross PCIe instead of 4 N, np.unpackbits expands
    them -- what VLMap.index_map returns (vlmap.py:123-124)"""
    lib = _lib.load()
    ap, ashape, ak = as_device(argmax, np.int32, stream)
    N = int(ashape[0])
    bits = DeviceArray(((N + 63) // 64,), np.uint64)
    _lib.check(lib.avl_mask_bits_from_argmax(ap, N, int(cat_id), bits.ptr, stream), "avl_mask_bits_from_argmax")
    packed = bits.numpy(stream)
    bits.free()
    return np.unpackbits(packed.view(np.uint8), count=N, bitorder="little").view(np.bool_)


def argmax_f32(vals, stream=None):
    """(index, value) of the first maximum of a float32 vector (habitat_lang_robot.py:427-430)."""
    lib = _lib.load()
    vp, vshape, vk = as_device(vals, np.float32, stream)
    idx, val = C.c_int64(), C.c_float()
    _lib.check(lib.avl_argmax_f32(vp, int(np.prod(vshape)), C.byref(idx), C.byref(val), stream), "avl_argmax_f32")
    return idx.value, val.value


class BatchPlan:
    """resolved device-pointer tables of one batch of frames (VoxelAccumulator.make_batch_plan)"""
    __slots__ = ("B", "H", "W", "Hf", "Wf", "P", "depth", "samples", "feat", "rgb", "keep")

    def __init__(self, B, H, W, Hf, Wf, P, depth, samples, feat, rgb, keep):
        self.B, self.H, self.W, self.Hf, self.Wf, self.P = B, H, W, Hf, Wf, P
        self.depth, self.samples, self.feat, self.rgb, self.keep = depth, samples, feat, rgb, keep


class VoxelAccumulator:
    """Device-resident map under construction (handle over avl_builder_*).

    One instance = the state the reference keeps in grid_feat / grid_pos / weight / grid_rgb / occupied_ids
    inside VLMapBuilder.create_mobile_base_map (vlmap_builder.py:86-95), living in HBM.
    """

    def __init__(self, gs, cs, vh, D, capacity=None, n_rows=None, max_capacity=None, deferred_fuse=False):
        """grid (gs, gs, vh) -- or (n_rows, gs, vh) for the rectangular global multi-floor map.
        capacity: voxels the accumulators hold initially (default gs * n_rows like the reference, vlmap_builder.py:202-206);
        max_capacity: like the reference's _reserve_map_space (:286-311) the accumulators DOUBLE when the map outgrows them,
        up to this many voxels (default: every cell of the grid, capped at 2^31 - 1); max_capacity=0 keeps the capacity fixed.
        deferred_fuse: frame-by-frame calls take ONE launch each (avl_builder_set_deferred_fuse): the features of a frame are
        read by the NEXT call's launch, so this object keeps them alive until then; same map, bit for bit (K3 sums a voxel's
        samples in ascending sample order in either mode)."""
        lib = _lib.load()
        _lib.require_gpu()
        self.gs, self.cs, self.vh, self.D = int(gs), float(cs), int(vh), int(D)
        self.n_rows = int(n_rows) if n_rows else self.gs
        ncell = self.n_rows * self.gs * self.vh
        cap0 = int(capacity) if capacity else min(self.n_rows * self.gs, ncell)
        h = C.c_void_p()
        _lib.check(lib.avl_builder_create_grid(C.byref(h), self.n_rows, self.gs, self.vh, self.cs, self.D, cap0),
                   "avl_builder_create_grid")
        self._h = h
        self._has_log = False
        self._max_frame, self._imported = 0, False          # bounds of the first-touch keys (key_bits)
        if max_capacity is None:
            max_capacity = min(ncell, (1 << 31) - 1)
        if max_capacity and max_capacity > cap0:
            _lib.check(lib.avl_builder_set_max_capacity(h, int(max_capacity)), "avl_builder_set_max_capacity")
        if deferred_fuse:
            self.set_deferred_fuse(True)

    def _retain(self, keeps, stream):
        """Keep the inputs of a launch alive until the GPU is done with them.  With deferred fuse the features of frame i are read
        by the launch of frame i + 1, so two generations are always held.  Buffers this module staged itself (NumPy inputs ->
        pooled DeviceArrays) go back to a pool that other threads and streams allocate from (map upload, checkpoint writer):
        those are released only behind an event recorded after the launch that reads them last (at most 4 generations wait, the
        oldest is synchronised -- this path already pays a blocking host-to-device copy per input); torch tensors return to
        torch's own stream-ordered allocator and need no event."""
        self._keep_prev, self._keep = getattr(self, "_keep", None), keeps
        if not any(isinstance(k, DeviceArray) for k in keeps):
            return
        lib = _lib.load()
        q = self.__dict__.setdefault("_keep_events", [])
        free = self.__dict__.setdefault("_free_events", [])
        if free:
            ev = free.pop()
        else:
            ev = C.c_void_p()
            _lib.check(lib.avl_event_create(C.byref(ev)), "avl_event_create")
        _lib.check(lib.avl_event_record(ev, stream), "avl_event_record")
        q.append((ev, keeps))
        while len(q) > 4:
            old_ev, _old = q.pop(0)
            # the launches that read _old must have completed: its own launch -- or, with deferred fuse, the NEXT one, which
            # fuses that frame's features (an event later in the stream covers the earlier one)
            _lib.check(lib.avl_event_sync(q[0][0] if getattr(self, "_deferred", False) else old_ev), "avl_event_sync")
            free.append(old_ev)

    def _calib_pair(self, calib, calib_inv):
        """(K, inv(K)) as contiguous float64; the inverse of an unchanged calibration is computed once (np.linalg.inv is
        ~8 us, most of the host cost of a frame-by-frame call)"""
        K = np.ascontiguousarray(np.asarray(calib, dtype=np.float64).reshape(3, 3))
        if calib_inv is not None:
            return K, np.ascontiguousarray(np.asarray(calib_inv, dtype=np.float64).reshape(3, 3))
        key = K.tobytes()
        c = getattr(self, "_kinv_cache", None)
        if c is None or c[0] != key:
            c = self._kinv_cache = (key, np.ascontiguousarray(np.linalg.inv(K)))
        return K, c[1]

    def key_bits(self) -> int:
        """first-touch keys of this map lie below 2^key_bits: frame index << 32 | sample index, bit 62 once a map was imported
        (avl_builder_import_map: new voxels order after every imported one) -- the bits the merge's key sort has to look at"""
        return 63 if self._imported else 32 + max(1, int(self._max_frame).bit_length())

    def set_deferred_fuse(self, on, stream=None):
        _lib.check(_lib.load().avl_builder_set_deferred_fuse(self._h, int(bool(on)), stream), "avl_builder_set_deferred_fuse")
        self._deferred = bool(on)
        return self

    def flush(self, stream=None):
        """run the feature fusion a deferred-fuse accumulator still owes (every read of the map does this by itself)"""
        _lib.check(_lib.load().avl_builder_flush(self._h, stream), "avl_builder_flush")
        return self

    def release_scratch(self, keep_bytes=0, stream=None):
        """drop the cached sorted replay log and trim the library's stream-ordered pool to keep_bytes (after the last merge /
        save of a build that shares the GPU with a feature extractor)"""
        _lib.check(_lib.load().avl_builder_release_scratch(self._h, int(keep_bytes), stream), "avl_builder_release_scratch")
        return self

    @property
    def capacity(self):
        c = C.c_int64()
        _lib.check(_lib.load().avl_builder_capacity(self._h, C.byref(c)), "avl_builder_capacity")
        return c.value

    def has_replay_log(self):
        return self._has_log

    def drop_replay_cache(self):
        """forget the log sorted by voxel that avl_builder_replay_chain keeps until the next frame (bench.py: a second merge of the
        same map should pay for the sort again)"""
        from .device import torch_stream_ptr
        _lib.check(_lib.load().avl_builder_drop_replay_cache(self._h, torch_stream_ptr()), "avl_builder_drop_replay_cache")

    def close(self):
        if getattr(self, "_h", None):
            try:
                lib = _lib.load()
                lib.avl_builder_destroy(self._h)          # synchronises: nothing reads the retained inputs any more
                for ev in [e for e, _ in self.__dict__.pop("_keep_events", [])] + self.__dict__.pop("_free_events", []):
                    lib.avl_event_destroy(ev)
            except Exception:      # interpreter shutdown: the module globals are already gone, the driver frees the memory
                pass
            self._h = None

    __del__ = close

    def reset(self, stream=None):
        _lib.check(_lib.load().avl_builder_reset(self._h, stream), "avl_builder_reset")
        self._max_frame, self._imported = 0, False

    def enable_replay_log(self, max_samples):
        """log every sampled pixel so that finalize() replays the reference's sequential weight / grid_rgb exactly"""
        _lib.check(_lib.load().avl_builder_enable_replay_log(self._h, int(max_samples)), "avl_builder_enable_replay_log")
        self._has_log = True
        return self

    def integrate_frame(self, depth, calib, pc_transform, sample_idx, feat_hwc, rgb, frame_idx, calib_inv=None,
                        min_depth=0.1, max_depth=6.0, sigma_sq=0.6, stream=None):
        """Fuse one frame.  depth (H,W) f32, feat_hwc (Hf,Wf,D) f32 channels-last, rgb (H,W,3) u8,
        sample_idx (P,) i32 in sampling order; calib 3x3 and pc_transform 4x4 are host float64."""
        lib = _lib.load()
        dp, dshape, k1 = as_device(depth, np.float32, stream)
        fp_, fshape, k2 = as_device(feat_hwc, np.float32, stream)
        rp, rshape, k3 = as_device(rgb, np.uint8, stream)
        sp, sshape, k4 = as_device(sample_idx, np.int32, stream)
        if len(dshape) != 2 or len(fshape) != 3 or fshape[2] != self.D or tuple(rshape) != (dshape[0], dshape[1], 3):
            raise ValueError(f"bad frame shapes depth {dshape} feat {fshape} rgb {rshape}")
        K, Kinv = self._calib_pair(calib, calib_inv)
        T = np.ascontiguousarray(np.asarray(pc_transform, dtype=np.float64).reshape(4, 4))
        rc = lib.avl_builder_integrate_frame(self._h, dp, dshape[0], dshape[1], K.ctypes.data, Kinv.ctypes.data, T.ctypes.data,
                                             sp, int(np.prod(sshape)), fp_, fshape[0], fshape[1], rp, int(frame_idx),
                                             float(min_depth), float(max_depth), float(sigma_sq), stream)
        _lib.check(rc, "avl_builder_integrate_frame")
        self._max_frame = max(self._max_frame, int(frame_idx))
        self._retain((k1, k2, k3, k4), stream)   # inputs must outlive the asynchronous launches
        return self

    def integrate_batch(self, depths, calib, pc_transforms, sample_idxs=None, feats_hwc=None, rgbs=None, frame_idx0=0, calib_inv=None,
                        min_depth=0.1, max_depth=6.0, sigma_sq=0.6, stream=None):
        """Fuse len(depths) consecutive frames with one launch pair (avl_builder_integrate_batch).  Arguments are lists of
        per-frame arrays (numpy / DeviceArray / torch CUDA) with identical shapes; results equal frame-by-frame fusion."""
        if isinstance(depths, BatchPlan):
            plan = depths
        else:
            plan = self.make_batch_plan(depths, sample_idxs, feats_hwc, rgbs, stream)
        return self._integrate_plan(plan, calib, pc_transforms, frame_idx0, calib_inv, min_depth, max_depth, sigma_sq, stream)

    def integrate_frames(self, plan, calib, pc_transforms, frame_idx0=0, calib_inv=None, min_depth=0.1, max_depth=6.0, sigma_sq=0.6,
                         stream=None):
        """The frame-by-frame loop from C (avl_builder_integrate_frames): the frames of `plan` (make_batch_plan) one after the other,
        one launch pair -- with deferred fuse one launch -- per frame, exactly like len(plan) integrate_frame calls without the
        ~12 us of Python / ctypes per call.  Not a batch: no two frames share a launch."""
        lib = _lib.load()
        B = plan.B
        K, Kinv = self._calib_pair(calib, calib_inv)
        T = np.ascontiguousarray(np.asarray(pc_transforms, dtype=np.float64).reshape(B, 16))
        rc = lib.avl_builder_integrate_frames(self._h, B, plan.depth, plan.H, plan.W, K.ctypes.data, Kinv.ctypes.data, T.ctypes.data,
                                              plan.samples, plan.P, plan.feat, plan.Hf, plan.Wf, plan.rgb, int(frame_idx0),
                                              float(min_depth), float(max_depth), float(sigma_sq), stream)
        _lib.check(rc, "avl_builder_integrate_frames")
        self._max_frame = max(self._max_frame, int(frame_idx0) + B)
        self._retain(plan.keep, stream)
        return self

    def make_batch_plan(self, depths, sample_idxs, feats_hwc, rgbs, stream=None):
        """Resolve the per-frame device pointers of a batch once.  A pipeline that cycles through a ring of frame buffers can
        keep the plan and pass it as `depths` to integrate_batch (sample_idxs / feats_hwc / rgbs are then ignored)."""
        B = len(depths)
        assert B > 0 and len(sample_idxs) == len(feats_hwc) == len(rgbs) == B
        keep, dptr, sptr, fptr, rptr = [], [], [], [], []
        shapes = None
        for i in range(B):
            dp, dshape, k1 = as_device(depths[i], np.float32, stream)
            fp_, fshape, k2 = as_device(feats_hwc[i], np.float32, stream)
            rp, rshape, k3 = as_device(rgbs[i], np.uint8, stream)
            sp, sshape, k4 = as_device(sample_idxs[i], np.int32, stream)
            sh = (tuple(dshape), tuple(fshape), tuple(rshape), int(np.prod(sshape)))
            if shapes is None:
                shapes = sh
            elif sh != shapes:
                raise ValueError("all frames of a batch must share their shapes")
            keep += [k1, k2, k3, k4]
            dptr.append(dp); fptr.append(fp_); rptr.append(rp); sptr.append(sp)
        (H, W), (Hf, Wf, D), _, P = shapes
        if D != self.D:
            raise ValueError(f"feature dim {D} != {self.D}")
        arr = lambda ptrs: (C.c_void_p * B)(*ptrs)
        return BatchPlan(B, H, W, Hf, Wf, P, arr(dptr), arr(sptr), arr(fptr), arr(rptr), keep)

    def _integrate_plan(self, plan, calib, pc_transforms, frame_idx0, calib_inv, min_depth, max_depth, sigma_sq, stream):
        lib = _lib.load()
        B = plan.B
        K, Kinv = self._calib_pair(calib, calib_inv)
        T = np.ascontiguousarray(np.asarray(pc_transforms, dtype=np.float64).reshape(B, 16))
        rc = lib.avl_builder_integrate_batch(self._h, B, plan.depth, plan.H, plan.W, K.ctypes.data, Kinv.ctypes.data, T.ctypes.data,
                                             plan.samples, plan.P, plan.feat, plan.Hf, plan.Wf, plan.rgb, int(frame_idx0),
                                             float(min_depth), float(max_depth), float(sigma_sq), stream)
        _lib.check(rc, "avl_builder_integrate_batch")
        self._max_frame = max(self._max_frame, int(frame_idx0) + B)
        self._retain(plan.keep, stream)
        return self

    def integrate_frame_global(self, depth, calib, transform, sample_idx, feat_hwc, rgb, frame_idx, pcd_min, depth_div=1000.0,
                               calib_inv=None, min_depth=0.1, max_depth=100.0, sigma_sq=0.6, stream=None):
        """Global (multi-floor) fusion, vlmap_builder_multi_floor.py:137-199.  depth: (H,W) uint16 (metres = value /
        depth_div, like the reference's PNG / 1000.0) or float32 metres; transform = camera_pose_tf @ habitat2cam_rot_tf."""
        lib = _lib.load()
        depth_np_u16 = isinstance(depth, np.ndarray) and depth.dtype == np.uint16
        is_u16 = depth_np_u16 or (_is_torch(depth) and str(depth.dtype) in ("torch.uint16", "torch.int16"))
        if isinstance(depth, np.ndarray):
            dp, dshape, k1 = as_device(depth.view(np.int16) if depth_np_u16 else depth, np.int16 if is_u16 else np.float32, stream)
        else:
            dp, dshape, k1 = depth.data_ptr(), tuple(depth.shape), depth
        fp_, fshape, k2 = as_device(feat_hwc, np.float32, stream)
        rp, rshape, k3 = as_device(rgb, np.uint8, stream)
        sp, sshape, k4 = as_device(sample_idx, np.int32, stream)
        K, Kinv = self._calib_pair(calib, calib_inv)
        T = np.ascontiguousarray(np.asarray(transform, dtype=np.float64).reshape(4, 4))
        pm = np.ascontiguousarray(pcd_min, dtype=np.float64)
        rc = lib.avl_builder_integrate_frame_global(self._h, dp, int(is_u16), float(depth_div), dshape[0], dshape[1], K.ctypes.data,
                                                    Kinv.ctypes.data, T.ctypes.data, sp, int(np.prod(sshape)), fp_, fshape[0],
                                                    fshape[1], rp, int(frame_idx), float(min_depth), float(max_depth),
                                                    float(sigma_sq), pm.ctypes.data, stream)
        _lib.check(rc, "avl_builder_integrate_frame_global")
        self._max_frame = max(self._max_frame, int(frame_idx))
        self._retain((k1, k2, k3, k4), stream)
        return self

    def import_map(self, grid_feat, grid_pos, weight, grid_rgb=None, stream=None):
        """seed an empty accumulator from a finished map (resume, vlmap_builder.py:212-222)"""
        lib = _lib.load()
        fp_, fshape, k1 = as_device(np.asarray(grid_feat, dtype=np.float32) if isinstance(grid_feat, np.ndarray) else grid_feat,
                                    np.float32, stream)
        pp, _, k2 = as_device(np.asarray(grid_pos, dtype=np.int32) if isinstance(grid_pos, np.ndarray) else grid_pos, np.int32, stream)
        wp, _, k3 = as_device(np.asarray(weight, dtype=np.float32) if isinstance(weight, np.ndarray) else weight, np.float32, stream)
        rp = None
        if grid_rgb is not None:
            rgb8 = np.clip(np.asarray(grid_rgb), 0, 255).astype(np.uint8) if isinstance(grid_rgb, np.ndarray) else grid_rgb
            rp, _, k4 = as_device(rgb8, np.uint8, stream)
        _lib.check(lib.avl_builder_import_map(self._h, fshape[0], fp_, pp, wp, rp, stream), "avl_builder_import_map")
        self._imported = True
        return self

    def mark_resumed(self, stream=None):
        """this (empty) accumulator continues a map another rank imported: same first-touch key space as that rank, so that in the
        merge the voxels of new frames order after every imported one (avl_builder_import_map with n = 0)"""
        _lib.check(_lib.load().avl_builder_import_map(self._h, 0, None, None, None, None, stream), "avl_builder_import_map")
        self._imported = True
        return self

    def num_voxels(self, stream=None):
        n = C.c_int64()
        _lib.check(_lib.load().avl_builder_num_voxels(self._h, C.byref(n), stream), "avl_builder_num_voxels")
        return n.value

    def num_points(self, stream=None):
        n = C.c_int64()
        _lib.check(_lib.load().avl_builder_num_points(self._h, C.byref(n), stream), "avl_builder_num_points")
        return n.value

    def num_groups(self, stream=None):
        n = C.c_int64()
        _lib.check(_lib.load().avl_builder_num_groups(self._h, C.byref(n), stream), "avl_builder_num_groups")
        return n.value

    def finalize(self, stream=None, want_occupied=True, as_numpy=True, as_torch=False, want_dirty=False):
        """-> dict(grid_feat, grid_pos, weight, grid_rgb, occupied_ids) in the reference's voxel-id order
        (numpy arrays; DeviceArrays with as_numpy=False; torch CUDA tensors with as_torch=True).
        want_dirty: also 'row_dirty' (n,) uint8 = rows whose voxel was fused since the previous finalize(want_dirty=True)
        (the flags are cleared): what an incremental checkpoint has to rewrite (avl_builder_finalize_ex)."""
        lib = _lib.load()
        n = self.num_voxels(stream)
        if as_torch:
            import torch
            dev = torch.device("cuda", torch.cuda.current_device())
            out = dict(grid_feat=torch.empty((n, self.D), dtype=torch.float32, device=dev),
                       grid_pos=torch.empty((n, 3), dtype=torch.int32, device=dev),
                       weight=torch.empty((n,), dtype=torch.float32, device=dev),
                       grid_rgb=torch.empty((n, 3), dtype=torch.uint8, device=dev),
                       occupied_ids=torch.empty((self.n_rows, self.gs, self.vh), dtype=torch.int32, device=dev) if want_occupied else None)
            _lib.check(lib.avl_builder_finalize(self._h, n, out["grid_feat"].data_ptr(), out["grid_pos"].data_ptr(), out["weight"].data_ptr(),
                                                out["grid_rgb"].data_ptr(),
                                                out["occupied_ids"].data_ptr() if want_occupied else None, stream), "avl_builder_finalize")
            return out
        gf = DeviceArray((n, self.D), np.float32)
        gp = DeviceArray((n, 3), np.int32)
        w = DeviceArray((n,), np.float32)
        rgb = DeviceArray((n, 3), np.uint8)
        occ = DeviceArray((self.n_rows, self.gs, self.vh), np.int32) if want_occupied else None
        dirty = DeviceArray((n,), np.uint8) if want_dirty else None
        _lib.check(lib.avl_builder_finalize_ex(self._h, n, gf.ptr, gp.ptr, w.ptr, rgb.ptr, occ.ptr if occ else None,
                                               dirty.ptr if dirty else None, 1 if want_dirty else 0, stream), "avl_builder_finalize")
        out = dict(grid_feat=gf, grid_pos=gp, weight=w, grid_rgb=rgb, occupied_ids=occ)
        if want_dirty:
            out["row_dirty"] = dirty
        if as_numpy:
            out = {k: (v.numpy(stream) if v is not None else None) for k, v in out.items()}
        return out

    def finalize_rows(self, n_saved, stream=None):
        """Lean checkpoint: finalise on the device, then bring to the host ONLY the rows a checkpoint has to write -- the rows
        below n_saved whose voxel was fused since the previous finalize(want_dirty=True) / finalize_rows, and the new rows
        [n_saved, n) -- gathered on the device (avl_gather_rows) and copied into a page-locked staging buffer.  A full
        finalize() copies the whole map (4.8 GB at 2.25 M voxels: 0.3-0.5 s on the builder's thread, as much as LSeg needs for
        the 100 frames between two checkpoints).
        -> dict(n, n_saved, idx (k,) int64 ascending row indices, rows {grid_feat, grid_pos, weight, grid_rgb: (k, ...) host arrays})
        The row arrays alias this accumulator's staging buffer: they are valid until its next finalize_rows call."""
        from .device import PinnedBuffer
        lib = _lib.load()
        dev = self.finalize(stream=stream, want_occupied=False, as_numpy=False, want_dirty=True)
        n = int(dev["grid_pos"].shape[0])
        n_saved = min(int(n_saved), n)
        dirty = dev["row_dirty"].numpy(stream)
        idx = np.concatenate([np.flatnonzero(dirty[:n_saved]), np.arange(n_saved, n)]).astype(np.int64)
        names = ("grid_feat", "grid_pos", "weight", "grid_rgb")
        sizes = {}
        for k in names:
            row_shape = tuple(dev[k].shape[1:])
            sizes[k] = (row_shape, int(np.prod(row_shape, dtype=np.int64)) * dev[k].dtype.itemsize)
        stage = getattr(self, "_stage", None)
        if stage is None:
            stage = self._stage = PinnedBuffer()
        total = sum((idx.size * rb + 255) // 256 * 256 for _, rb in sizes.values())
        stage.reserve(max(total, 256))
        rows, off = {}, 0
        if idx.size:
            d_idx = DeviceArray.from_numpy(idx, stream)
            for k in names:
                row_shape, rb = sizes[k]
                dst = DeviceArray((idx.size,) + row_shape, dev[k].dtype)
                _lib.check(lib.avl_gather_rows(dev[k].ptr, rb, d_idx.ptr, idx.size, dst.ptr, stream), "avl_gather_rows")
                rows[k] = stage.view(off, (idx.size,) + row_shape, dev[k].dtype)
                _lib.check(lib.avl_memcpy_d2h(rows[k].ctypes.data, dst.ptr, dst.nbytes, stream), "avl_memcpy_d2h")
                off += (dst.nbytes + 255) // 256 * 256
        else:
            for k in names:
                rows[k] = np.empty((0,) + sizes[k][0], dev[k].dtype)
        return dict(n=n, n_saved=n_saved, idx=idx, rows=rows)

    def export_raw(self, stream=None, as_numpy=True):
        """raw accumulators of all voxels (see avl_builder_export_raw) for the multi-GPU merge"""
        lib = _lib.load()
        n = self.num_voxels(stream)
        arrs = dict(cell=DeviceArray((n,), np.int32), first_key=DeviceArray((n,), np.uint64),
                    sum_feat=DeviceArray((n, self.D), np.float64), sum_w4=DeviceArray((n, 4), np.float64),
                    first_feat=DeviceArray((n, self.D), np.float32), first_alpha=DeviceArray((n,), np.float64))
        _lib.check(lib.avl_builder_export_raw(self._h, n, *(a.ptr for a in arrs.values()), stream), "avl_builder_export_raw")
        _lib.check(lib.avl_stream_sync(stream))
        return {k: v.numpy(stream) for k, v in arrs.items()} if as_numpy else arrs


def points_bbox(minmax, depth, calib, transform, sample_idx, depth_div=1000.0, min_depth=0.1, max_depth=100.0, stream=None):
    """Pass 1 of the global builder (vlmap_builder_multi_floor.py:97-118): fold one frame's transformed sampled points into
    minmax (6,) float64 [min xyz, max xyz] in place.  depth: uint16 (value / depth_div metres) or float32 metres."""
    lib = _lib.load()
    is_u16 = isinstance(depth, np.ndarray) and depth.dtype == np.uint16
    dp, dshape, k1 = as_device(depth.view(np.int16) if is_u16 else depth, np.int16 if is_u16 else np.float32, stream)
    sp, sshape, k2 = as_device(sample_idx, np.int32, stream)
    Kinv = np.ascontiguousarray(np.linalg.inv(np.asarray(calib, dtype=np.float64).reshape(3, 3)))
    T = np.ascontiguousarray(np.asarray(transform, dtype=np.float64).reshape(4, 4))
    assert minmax.dtype == np.float64 and minmax.flags["C_CONTIGUOUS"] and minmax.shape == (6,)
    rc = lib.avl_points_bbox(dp, int(is_u16), float(depth_div), dshape[0], dshape[1], Kinv.ctypes.data, T.ctypes.data, sp,
                             int(np.prod(sshape)), float(min_depth), float(max_depth), minmax.ctypes.data, stream)
    _lib.check(rc, "avl_points_bbox")
    return minmax


def finalize_raw(raw, D, gs, vh, stream=None, n_rows=None):
    """Stateless finalisation of (merged) raw accumulators -> the reference's arrays (numpy in, numpy out)."""
    lib = _lib.load()
    n = len(raw["cell"])
    dev = {k: as_device(raw[k], dt, stream) for k, dt in (("cell", np.int32), ("sum_feat", np.float64), ("sum_w4", np.float64),
                                                          ("first_feat", np.float32), ("first_alpha", np.float64))}
    gf, gp = DeviceArray((n, D), np.float32), DeviceArray((n, 3), np.int32)
    w, rgb = DeviceArray((n,), np.float32), DeviceArray((n, 3), np.uint8)
    occ = DeviceArray((n_rows or gs, gs, vh), np.int32)
    _lib.check(lib.avl_memset(occ.ptr, 0xFF, occ.nbytes, stream))
    rc = lib.avl_finalize_raw(n, D, gs, vh, dev["cell"][0], dev["sum_feat"][0], dev["sum_w4"][0], dev["first_feat"][0],
                              dev["first_alpha"][0], gf.ptr, gp.ptr, w.ptr, rgb.ptr, occ.ptr, stream)
    _lib.check(rc, "avl_finalize_raw")
    return dict(grid_feat=gf.numpy(stream), grid_pos=gp.numpy(stream), weight=w.numpy(stream), grid_rgb=rgb.numpy(stream),
                occupied_ids=occ.numpy(stream))


def finalize_merged(merged, D, gs, vh, stream=None, n_rows=None):
    """Finalise accumulators merged by parallel.merge_raw / merge_raw_local: merged = dict(cell (M,) int32, acc (M, D+4) f64)
    in voxel-id order with the first-touch term folded in (avl_finalize_merged).  numpy / torch in, numpy out."""
    lib = _lib.load()
    M = len(merged["cell"])
    cp, _, k1 = as_device(merged["cell"], np.int32, stream)
    ap, ashape, k2 = as_device(merged["acc"], np.float64, stream)
    assert tuple(ashape) == (M, D + 4), ashape
    gf, gp = DeviceArray((M, D), np.float32), DeviceArray((M, 3), np.int32)
    w, rgb = DeviceArray((M,), np.float32), DeviceArray((M, 3), np.uint8)
    occ = DeviceArray((n_rows or gs, gs, vh), np.int32)
    _lib.check(lib.avl_memset(occ.ptr, 0xFF, occ.nbytes, stream))
    _lib.check(lib.avl_finalize_merged(M, 0, D, gs, vh, cp, ap, D + 4, gf.ptr, gp.ptr, w.ptr, rgb.ptr, occ.ptr, stream),
               "avl_finalize_merged")
    return dict(grid_feat=gf.numpy(stream), grid_pos=gp.numpy(stream), weight=w.numpy(stream), grid_rgb=rgb.numpy(stream),
                occupied_ids=occ.numpy(stream))


_HOST_PLANS = []          # [(weakref to the host array, fingerprint, HeatPlan)]: most recent first, at most two maps


def _fingerprint(a: np.ndarray) -> int:
    """hash of the WHOLE array (a fingerprint of sampled rows misses an in-place edit elsewhere and the stale cell order
    gives a silently wrong heat).  xxh3 runs at ~10 GB/s (2 ms for a 2 M-voxel map's 24 MB); zlib.crc32 is the fall-back."""
    buf = memoryview(a).cast("B")
    try:
        import xxhash
        return xxhash.xxh3_64_intdigest(buf)
    except ImportError:
        import zlib
        return zlib.crc32(buf)


def _host_heat_plan(grid_pos: np.ndarray, stream=None):
    """HeatPlan for a HOST position array that is passed again and again (upstream's AVLMap.index_object and the navigator hand
    the map's grid_pos to get_heatmap_from_mask_3d on every query): kept per array object, checked against a hash of ALL of
    its contents so that an array edited in place gets a new plan.  Only for arrays that reach the library unchanged (int32,
    C-contiguous): a converted temporary is a new object on every call and would build -- and keep alive -- a plan per query.
    None if the map cannot have one."""
    import weakref
    fp = (len(grid_pos), grid_pos.__array_interface__["data"][0], _fingerprint(grid_pos))
    for k, (ref, f, plan) in enumerate(_HOST_PLANS):
        if ref() is grid_pos and f == fp:
            if k:
                _HOST_PLANS.insert(0, _HOST_PLANS.pop(k))
            return plan
    stale = [e for e in _HOST_PLANS if e[0]() is None or e[0]() is grid_pos]
    _HOST_PLANS[:] = [e for e in _HOST_PLANS if e[0]() is not None and e[0]() is not grid_pos]
    for _, _, old in stale:                    # the array died or was edited in place: its plan's device buffers go now
        if old is not None:
            old.close()
    plan = HeatPlan.for_positions(DeviceArray.from_numpy(grid_pos), stream)
    try:
        _HOST_PLANS.insert(0, (weakref.ref(grid_pos), fp, plan))
    except TypeError:
        return plan
    for _, _, old in _HOST_PLANS[2:]:
        if old is not None:
            old.close()
    del _HOST_PLANS[2:]
    return plan


def heatmap_from_mask(grid_pos, mask, cell_size=0.05, decay_rate=0.01, stream=None, reuse_plan=False):
    """visualize_utils.py:29-49 on the GPU.  grid_pos (N,3) int32, mask (N,) bool/uint8 -> (N,) float32.
    reuse_plan (host grid_pos only, int32 and C-contiguous, i.e. the SAME array object on every call): keep the map's positions
    and their cell order on the device between calls (HeatPlan) -- the second query on the same array uploads the mask alone and
    runs the 3-4x faster planned kernel; same bits.  The whole array is hashed per call (~2 ms at 2 M voxels), so an in-place
    edit gets a fresh plan; VLMap / AVLMap.index_object hold their own plan and skip the hash."""
    lib = _lib.load()
    _lib.require_gpu()
    if (reuse_plan and isinstance(grid_pos, np.ndarray) and grid_pos.ndim == 2 and len(grid_pos) > 0 and grid_pos.dtype == np.int32
            and grid_pos.flags.c_contiguous):
        plan = _host_heat_plan(grid_pos, stream)
        if plan is not None:
            return plan(np.asarray(mask), cell_size, decay_rate, stream).numpy(stream)
    if _is_torch(mask):
        import torch
        if mask.dtype == torch.bool:
            mask = mask.to(torch.uint8)
    elif isinstance(mask, np.ndarray):
        mask = mask.astype(np.uint8)
    pp, pshape, k1 = as_device(grid_pos, np.int32, stream)
    mp, mshape, k2 = as_device(mask, np.uint8, stream)
    N = pshape[0]
    if _is_torch(grid_pos):
        import torch
        heat = torch.empty((N,), dtype=torch.float32, device=grid_pos.device)
        _lib.check(lib.avl_heatmap_from_mask(pp, mp, N, float(cell_size), float(decay_rate), heat.data_ptr(), stream),
                   "avl_heatmap_from_mask")
        return heat
    heat = DeviceArray((N,), np.float32)
    _lib.check(lib.avl_heatmap_from_mask(pp, mp, N, float(cell_size), float(decay_rate), heat.ptr, stream),
               "avl_heatmap_from_mask")
    return heat.numpy(stream) if isinstance(grid_pos, np.ndarray) else heat


class HeatPlan:
    """The part of heatmap_from_mask that depends on the voxel positions only, kept for a map that is queried repeatedly
    (avl_heat_plan: bounding box, voxels in cell order, zeroed grid buffers).  plan(mask, cell_size, decay_rate) returns the same
    bits as heatmap_from_mask(grid_pos, mask, ...), ~3x sooner.  grid_pos: DeviceArray or CUDA tensor (N,3) int32, kept alive here.
    HeatPlan.for_positions returns None when the map's bounding box is too large for a plan (>= 2^32 cells)."""

    def __init__(self, grid_pos, stream=None):
        lib = _lib.load()
        _lib.require_gpu()
        if isinstance(grid_pos, np.ndarray):
            grid_pos = DeviceArray.from_numpy(np.ascontiguousarray(grid_pos, dtype=np.int32))
        pp, pshape, self._keep = as_device(grid_pos, np.int32, stream)
        self.grid_pos, self.N = grid_pos, int(pshape[0])
        self._torch = _is_torch(grid_pos)
        h = C.c_void_p()
        _lib.check(lib.avl_heat_plan_create(C.byref(h), pp, self.N, stream), "avl_heat_plan_create")
        self._h, self._lib = h, lib

    @classmethod
    def for_positions(cls, grid_pos, stream=None):
        try:
            return cls(grid_pos, stream)
        except _lib.AvlError as e:
            if "2^32" in str(e):
                return None
            raise

    def __call__(self, mask, cell_size=0.05, decay_rate=0.01, stream=None):
        if self._h is None:
            raise RuntimeError("HeatPlan used after close()")
        if _is_torch(mask):
            import torch
            if mask.dtype == torch.bool:
                mask = mask.to(torch.uint8)
        elif isinstance(mask, np.ndarray):
            mask = mask.astype(np.uint8)
        mp, mshape, keep = as_device(mask, np.uint8, stream)
        if int(np.prod(mshape)) != self.N:
            raise ValueError(f"mask has {int(np.prod(mshape))} entries, the plan {self.N} voxels")
        if self._torch:
            import torch
            heat = torch.empty((self.N,), dtype=torch.float32, device=self.grid_pos.device)
            hp = heat.data_ptr()
        else:
            heat = DeviceArray((self.N,), np.float32)
            hp = heat.ptr
        _lib.check(self._lib.avl_heatmap_from_mask_planned(self._h, mp, float(cell_size), float(decay_rate), hp, stream),
                   "avl_heatmap_from_mask_planned")
        return heat

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h is not None:
            self._lib.avl_heat_plan_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---------------------------------------------------------------------------------------- top-down 2-D products
def _dev_u8(x, stream):
    """bool / uint8 host array, DeviceArray or torch tensor -> uint8 device pointer"""
    if isinstance(x, np.ndarray):
        x = np.ascontiguousarray(x).astype(np.uint8, copy=False)
    elif _is_torch(x):
        import torch
        if x.dtype == torch.bool:
            x = x.to(torch.uint8)
    return as_device(x, np.uint8, stream)


def pool_label_2d(mask_3d, grid_pos, gs, stream=None):
    """(gs, gs) bool: mask_2d[row, col] |= mask_3d[i]  (visualize_utils.py:77-83).  Inputs host or device-resident."""
    lib = _lib.load()
    _lib.require_gpu()
    pp, pshape, k1 = as_device(grid_pos, np.int32, stream)
    mp, mshape, k2 = _dev_u8(mask_3d, stream)
    out = DeviceArray((gs, gs), np.uint8)
    _lib.check(lib.avl_pool_label_2d(pp, mp, pshape[0], int(gs), out.ptr, stream), "avl_pool_label_2d")
    return out.numpy(stream).astype(bool)


def rgb_topdown(grid_pos, grid_rgb, gs, stream=None):
    """(gs, gs, 3) uint8 colour map, last voxel of a column wins  (map.py:106-113)"""
    lib = _lib.load()
    _lib.require_gpu()
    pp, pshape, k1 = as_device(grid_pos, np.int32, stream)
    rgb8 = np.ascontiguousarray(grid_rgb).astype(np.uint8) if isinstance(grid_rgb, np.ndarray) else grid_rgb
    rp, _, k2 = as_device(rgb8, np.uint8, stream)
    out = DeviceArray((gs, gs, 3), np.uint8)
    _lib.check(lib.avl_rgb_topdown(pp, rp, pshape[0], int(gs), out.ptr, stream), "avl_rgb_topdown")
    return out.numpy(stream)


def obstacle_map(occupied_ids, cs, h_min=0, h_max=1.5, stream=None):
    """(n0, n1) bool, True = free: no voxel id > 0 with h_min < height < h_max  (map.py:79-95)"""
    lib = _lib.load()
    _lib.require_gpu()
    op, oshape, k1 = as_device(occupied_ids, np.int32, stream)
    n0, n1, vh = oshape
    heights = np.arange(0, vh) * cs                                     # float64, exactly as upstream
    sel = np.flatnonzero(np.logical_and(heights > h_min, heights < h_max))
    h0, h1 = (int(sel[0]), int(sel[-1]) + 1) if sel.size else (0, 0)   # heights ascend: the mask is one index range
    out = DeviceArray((n0, n1), np.uint8)
    _lib.check(lib.avl_obstacle_map(op, n0, n1, vh, h0, h1, out.ptr, stream), "avl_obstacle_map")
    return out.numpy(stream).astype(bool)


def obstacle_scatter(grid_pos, predict, obs_inds, n_classes, obstacles_cropped, rmin, cmin, stream=None):
    """index_utils.py:163-177: cropped map, True = free, after marking the voxels whose class is in obs_inds.
    predict may be the device-resident argmax of sim_scores."""
    lib = _lib.load()
    _lib.require_gpu()
    pp, pshape, k1 = as_device(grid_pos, np.int32, stream)
    cp, cshape, k2 = as_device(predict, np.int32, stream)
    crop = np.ascontiguousarray(np.asarray(obstacles_cropped) != 0).astype(np.uint8)
    H, W = crop.shape
    fp_, _, k3 = as_device(crop, np.uint8, stream)
    table = np.zeros((int(n_classes),), dtype=np.uint8)
    table[list(obs_inds)] = 1
    out = DeviceArray((H, W), np.uint8)
    _lib.check(lib.avl_obstacle_scatter(pp, cp, pshape[0], table.ctypes.data, int(n_classes), int(rmin), int(cmin), H, W, fp_, out.ptr,
                                        stream), "avl_obstacle_scatter")
    return out.numpy(stream).astype(bool)


def export_raw_torch(acc: "VoxelAccumulator", device=None, stream=None):
    """VoxelAccumulator.export_raw into torch tensors on the GPU (input of parallel.merge_raw)."""
    import torch
    lib = _lib.load()
    n = acc.num_voxels(stream)
    device = device or torch.device("cuda", torch.cuda.current_device())
    t = dict(cell=torch.empty((n,), dtype=torch.int32, device=device),
             first_key=torch.empty((n,), dtype=torch.int64, device=device),
             sum_feat=torch.empty((n, acc.D), dtype=torch.float64, device=device),
             sum_w4=torch.empty((n, 4), dtype=torch.float64, device=device),
             first_feat=torch.empty((n, acc.D), dtype=torch.float32, device=device),
             first_alpha=torch.empty((n,), dtype=torch.float64, device=device))
    _lib.check(lib.avl_builder_export_raw(acc._h, n, *(v.data_ptr() for v in t.values()), stream), "avl_builder_export_raw")
    _lib.check(lib.avl_stream_sync(stream))
    return t


def topk_f32(vals, k, stream=None):
    """(indices (k,) int64, values (k,) float32) of the k largest entries, descending, ties by ascending index."""
    lib = _lib.load()
    vp, vshape, vk = as_device(vals, np.float32, stream)
    idx = np.empty((k,), dtype=np.int64)
    val = np.empty((k,), dtype=np.float32)
    _lib.check(lib.avl_topk_f32(vp, int(np.prod(vshape)), int(k), idx.ctypes.data, val.ctypes.data, stream), "avl_topk_f32")
    return idx, val
