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
        if self.nbytes:
            _lib.check(_lib.load().avl_memcpy_d2h(out.ctypes.data, self.ptr, self.nbytes, stream), "avl_memcpy_d2h")
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
    if isinstance(x, DeviceArray):
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
