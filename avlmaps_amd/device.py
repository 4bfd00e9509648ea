"""Device-buffer plumbing for the ctypes layer.

Accepts three kinds of array arguments and always hands a raw device pointer to the C ABI:
  * numpy arrays        -> staged through a library-allocated device buffer (avl_malloc + H2D)
  * torch CUDA tensors  -> zero-copy (data_ptr)
  * DeviceArray         -> this module's own minimal device array (no torch needed)
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


# Freed buffers are kept on a free list keyed by (device, size) (hipMalloc / hipFree cost ~100 us each and hipFree synchronises): the
# per-frame staging buffers of the builder are then recycled instead of reallocated.  Bounded; oversize buffers go back.
_POOL: dict = {}
_POOL_BYTES = [0]
_POOL_LIMIT = 2 << 30
_POOL_MAX_ITEM = 256 << 20
_H2D_PIECE = 64 << 20
_D2H_PARALLEL_MIN = 256 << 20


def _pool_device() -> int:
    """the GPU the calling thread allocates on (HIP keeps the current device per thread): pooled pointers are only ever handed
    back out on the device they were allocated on"""
    n = C.c_int(0)
    return n.value if _lib.load().avl_get_device(C.byref(n)) == _lib.AVL_OK else 0


def _pool_alloc(nbytes: int, device: int) -> int:
    lst = _POOL.get((device, nbytes))
    if lst:
        try:
            ptr = lst.pop()               # another thread (map upload, checkpoint writer) may have taken the last one
        except IndexError:
            ptr = None
        if ptr is not None:
            _POOL_BYTES[0] -= nbytes
            return ptr
    p = C.c_void_p()
    _lib.check(_lib.load().avl_malloc(C.byref(p), nbytes), "avl_malloc")
    return p.value or 0


def _pool_free(ptr: int, nbytes: int, device: int) -> None:
    if 0 < nbytes <= _POOL_MAX_ITEM and _POOL_BYTES[0] + nbytes <= _POOL_LIMIT:
        _POOL.setdefault((device, nbytes), []).append(ptr)
        _POOL_BYTES[0] += nbytes
    else:
        _lib.load().avl_free(ptr)


def empty_pool() -> None:
    for key, lst in list(_POOL.items()):
        for p in lst:
            _lib.load().avl_free(p)
    _POOL.clear()
    _POOL_BYTES[0] = 0


class DeviceArray:
    """Minimal owning device array (shape + dtype + pointer) backed by avl_malloc (pooled)."""

    def __init__(self, shape, dtype):
        self.shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        self.device = _pool_device()
        self.ptr = _pool_alloc(max(self.nbytes, 1), self.device)

    @classmethod
    def from_numpy(cls, a, stream=None):
        a = np.ascontiguousarray(a)
        d = cls(a.shape, a.dtype)
        if a.nbytes:
            lib, src = _lib.load(), a.ctypes.data
            # in pieces: one pageable copy of several GB keeps the process's memory map locked while its pages are pinned, which
            # stalls every other thread of the process that allocates or frees (seen: 0.8 s of a map upload on a worker thread)
            for off in range(0, a.nbytes, _H2D_PIECE):
                _lib.check(lib.avl_memcpy_h2d(d.ptr + off, src + off, min(_H2D_PIECE, a.nbytes - off), stream), "avl_memcpy_h2d")
            _lib.check(lib.avl_stream_sync(stream), "avl_stream_sync")   # source may be a temporary
        return d

    def zero_(self, stream=None):
        _lib.check(_lib.load().avl_memset(self.ptr, 0, self.nbytes, stream), "avl_memset")
        return self

    def numpy(self, stream=None):
        out = np.empty(self.shape, dtype=self.dtype)
        if not self.nbytes:
            return out
        lib = _lib.load()
        if self.nbytes < _D2H_PARALLEL_MIN:
            _lib.check(lib.avl_memcpy_d2h(out.ctypes.data, self.ptr, self.nbytes, stream), "avl_memcpy_d2h")
            return out
        # A multi-GB copy into a FRESH host array is bound by the page faults of its first touch, not by PCIe (measured on the GPU
        # box: 3.2 GB in 0.33 s into np.empty, 0.06 s into memory that has been touched before): several host threads copy -- and
        # fault in -- disjoint pieces at once.  The map's final save and the first full checkpoint come through here.
        import threading
        _lib.check(lib.avl_stream_sync(stream), "avl_stream_sync")
        dev, dst, errs = self.device, out.ctypes.data, []
        nthr = min(8, max(2, self.nbytes // (128 << 20)))
        piece = ((self.nbytes + nthr - 1) // nthr + 4095) & ~4095

        def work(k):
            try:
                _lib.set_device(dev)
                off = k * piece
                m = min(piece, self.nbytes - off)
                if m > 0:
                    _lib.check(lib.avl_memcpy_d2h(dst + off, self.ptr + off, m, None), "avl_memcpy_d2h")
            except BaseException as e:      # surfaced on the calling thread
                errs.append(e)
        ths = [threading.Thread(target=work, args=(k,), name="avl-d2h") for k in range(nthr)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        if errs:
            raise errs[0]
        return out

    def free(self):
        if getattr(self, "ptr", 0):
            _pool_free(self.ptr, max(self.nbytes, 1), getattr(self, "device", 0))
            self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class PinnedBuffer:
    """Reusable page-locked host buffer (avl_host_alloc).  view(nbytes-offset, shape, dtype) hands out NumPy arrays that ALIAS
    it: they are valid until the buffer is reused or freed."""

    def __init__(self):
        self.ptr, self.nbytes = 0, 0

    def reserve(self, nbytes: int) -> None:
        if nbytes <= self.nbytes:
            return
        self.free()
        p = C.c_void_p()
        cap = max(int(nbytes * 1.25), 1 << 20)
        _lib.check(_lib.load().avl_host_alloc(C.byref(p), cap), "avl_host_alloc")
        self.ptr, self.nbytes = p.value or 0, cap

    def view(self, offset: int, shape, dtype) -> np.ndarray:
        dtype = np.dtype(dtype)
        n = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
        assert offset + n <= self.nbytes
        buf = (C.c_char * max(n, 1)).from_address(self.ptr + offset)
        a = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape, dtype=np.int64))).reshape(shape)
        return a

    def free(self) -> None:
        if self.ptr:
            try:
                _lib.load().avl_host_free(self.ptr)
            except Exception:
                pass
            self.ptr, self.nbytes = 0, 0

    __del__ = free


class DeviceView:
    """non-owning (ptr, shape, dtype) view of device memory somebody else keeps alive (a FrameStager slot)"""
    __slots__ = ("ptr", "shape", "dtype", "nbytes")

    def __init__(self, ptr, shape, dtype):
        self.ptr, self.shape, self.dtype = int(ptr), tuple(int(s) for s in shape), np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize


class StagedFrame:
    """one frame's depth / rgb / sample list on the device (views of a FrameStager slot) + the event its copies complete behind"""
    __slots__ = ("depth", "rgb", "samples", "slot", "ready", "ring")

    def __init__(self, depth, rgb, samples, slot, ready, ring):
        self.depth, self.rgb, self.samples, self.slot, self.ready, self.ring = depth, rgb, samples, slot, ready, ring


class _StageRing:
    """the slots of one frame layout: page-locked host buffers, their device twins, a `ready` and a `free` event per slot"""

    def __init__(self, n, layout, offsets):
        lib = _lib.load()
        self.layout, self.offsets = layout, offsets
        self.pinned, self.dev, self.ready, self.free_ev, self.free_pending = [], [], [], [], []
        for _ in range(n):
            pb = PinnedBuffer()
            pb.reserve(offsets[3])
            self.pinned.append(pb)
            self.dev.append(DeviceArray((offsets[3],), np.uint8))
            evs = []
            for _e in range(2):
                ev = C.c_void_p()
                _lib.check(lib.avl_event_create(C.byref(ev)), "avl_event_create")
                evs.append(ev)
            self.ready.append(evs[0])
            self.free_ev.append(evs[1])
            self.free_pending.append(False)

    def close(self):
        lib = _lib.load()
        for k in range(len(self.dev)):
            if self.free_pending[k]:
                lib.avl_event_sync(self.free_ev[k])
            lib.avl_event_destroy(self.ready[k])
            lib.avl_event_destroy(self.free_ev[k])
            self.dev[k].free()
            self.pinned[k].free()
        self.pinned, self.dev, self.ready, self.free_ev, self.free_pending = [], [], [], [], []


class FrameStager:
    """Pinned, asynchronous host-to-device path of the builder's frame loop (a frame was 22 us of kernels inside
    ~1.5 ms of pageable, synchronous copies on the fusing thread).

    A ring of slots, each a page-locked host buffer [depth f32 | rgb u8 | samples i32] and its device twin.  A PRODUCER thread
    (the builder's stager thread) copies a decoded frame into a free slot's pinned buffer and queues ONE asynchronous copy of the
    whole slot on the stager's own copy stream, followed by an event; the FUSING thread makes its stream wait for that event on
    the device (avl_stream_wait_event: the host never blocks), launches the frame kernels on views of the slot, and records the
    slot's `free` event behind them.  A slot is refilled only after its free event has completed; the ring must therefore have more
    slots than frames can be in flight between stage() and release() (the builder sizes it: queue depth + batch + 3).  5.4 MB per
    720x1080 frame cross PCIe at the pinned rate (~0.1 ms) while earlier frames are being fused.  A change of the frame shape
    starts a NEW ring; the old one lives until close(), because frames staged in it may still be queued."""

    def __init__(self, slots: int, device=None):
        lib = _lib.load()
        self.n = max(2, int(slots))
        self.device = _lib.current_device() if device is None else int(device)
        st = C.c_void_p()
        _lib.check(lib.avl_stream_create(C.byref(st)), "avl_stream_create")
        self.copy_stream = st
        self.ring = None
        self.retired = []
        self.next = 0

    def _ring_for(self, depth, rgb, samples):
        lay = (tuple(depth.shape), tuple(rgb.shape), int(samples.shape[0]))
        r = self.ring
        if r is not None and r.layout[:2] == lay[:2] and r.layout[2] >= lay[2]:
            return r
        if r is not None:
            self.retired.append(r)
        nd, nr = depth.size * 4, rgb.size
        off_r = (nd + 255) & ~255
        off_s = (off_r + nr + 255) & ~255
        cap_s = max(lay[2], 1) + 64
        self.ring = _StageRing(self.n, (lay[0], lay[1], cap_s), (0, off_r, off_s, off_s + cap_s * 4))
        self.next = 0
        return self.ring

    def stage(self, depth: np.ndarray, rgb: np.ndarray, samples: np.ndarray) -> StagedFrame:
        """producer thread: copy the frame into the next slot and start its transfer"""
        lib = _lib.load()
        depth = np.asarray(depth, dtype=np.float32)
        rgb = np.asarray(rgb, dtype=np.uint8)
        samples = np.asarray(samples, dtype=np.int32).reshape(-1)
        ring = self._ring_for(depth, rgb, samples)
        k = self.next
        self.next = (k + 1) % self.n
        if ring.free_pending[k]:
            _lib.check(lib.avl_event_sync(ring.free_ev[k]), "avl_event_sync")      # the launches that read this slot have finished
            ring.free_pending[k] = False
        _, off_r, off_s, _total = ring.offsets
        pb = ring.pinned[k]
        np.copyto(pb.view(0, depth.shape, np.float32), depth)
        np.copyto(pb.view(off_r, rgb.shape, np.uint8), rgb)
        ns = int(samples.shape[0])
        np.copyto(pb.view(off_s, (ns,), np.int32), samples)
        d = ring.dev[k]
        _lib.check(lib.avl_memcpy_h2d(d.ptr, pb.ptr, off_s + ns * 4, self.copy_stream), "avl_memcpy_h2d")
        _lib.check(lib.avl_event_record(ring.ready[k], self.copy_stream), "avl_event_record")
        return StagedFrame(DeviceView(d.ptr, depth.shape, np.float32), DeviceView(d.ptr + off_r, rgb.shape, np.uint8),
                           DeviceView(d.ptr + off_s, (ns,), np.int32), k, ring.ready[k], ring)

    def acquire(self, staged: StagedFrame, stream=None) -> None:
        """fusing thread, before the launches that read the frame: `stream` waits (on the device) for the slot's copies"""
        _lib.check(_lib.load().avl_stream_wait_event(stream, staged.ready), "avl_stream_wait_event")

    def release(self, staged: StagedFrame, stream=None) -> None:
        """fusing thread, after the last launch that reads the frame's depth / rgb / samples has been queued on `stream`"""
        _lib.check(_lib.load().avl_event_record(staged.ring.free_ev[staged.slot], stream), "avl_event_record")
        staged.ring.free_pending[staged.slot] = True

    def close(self):
        try:
            lib = _lib.load()
            if self.copy_stream:
                lib.avl_stream_sync(self.copy_stream)
            for r in self.retired + ([self.ring] if self.ring is not None else []):
                r.close()
            self.retired, self.ring = [], None
            if self.copy_stream:
                lib.avl_stream_destroy(self.copy_stream)
                self.copy_stream = None
        except Exception:
            pass

    __del__ = close


def _is_torch(x):
    return type(x).__module__.startswith("torch") and hasattr(x, "data_ptr")


_TORCH_DTYPES = {}


def _torch_dtype(dtype):
    if not _TORCH_DTYPES:
        import torch
        _TORCH_DTYPES.update({np.dtype(np.float32): torch.float32, np.dtype(np.int32): torch.int32, np.dtype(np.uint8): torch.uint8,
                              np.dtype(np.int64): torch.int64, np.dtype(np.float64): torch.float64, np.dtype(np.int16): torch.int16})
    return _TORCH_DTYPES[dtype]


def as_device(x, dtype, stream=None):
    """-> (ptr, shape, keepalive).  Raises if a torch tensor is not a contiguous CUDA tensor of `dtype`."""
    dtype = np.dtype(dtype)
    if isinstance(x, (DeviceArray, DeviceView)):
        if x.dtype != dtype:
            raise TypeError(f"expected {dtype}, got {x.dtype}")
        return x.ptr, x.shape, x
    if _is_torch(x):
        want = _torch_dtype(dtype)
        if not x.is_cuda:
            raise TypeError("torch tensors passed to avlmaps_amd must live on the GPU")
        if x.dtype != want:
            raise TypeError(f"expected torch dtype {want}, got {x.dtype}")
        if not x.is_contiguous():
            x = x.contiguous()
        return x.data_ptr(), tuple(x.shape), x
    a = np.ascontiguousarray(x, dtype=dtype)
    d = DeviceArray.from_numpy(a, stream)
    return d.ptr, d.shape, d


def torch_stream_ptr():
    """raw hipStream_t of torch's current stream, or None when torch/GPU is unavailable"""
    try:
        import torch
        if torch.cuda.is_available():
            return torch.cuda.current_stream().cuda_stream or None
    except Exception:
        pass
    return None
