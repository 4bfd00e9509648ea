"""Minimal HDF5 reader / writer over the HDF5 C library through ctypes -- the on-disk format of the reference's map file.

The reference stores the map with h5py (`f.create_dataset(name, data=array)`, avlmaps/utils/mapping_utils.py:469-505) and
reads it back with `f[name][:]` (:508-541).  h5py is a binding of libhdf5; where h5py is not installed but the C library is
(e.g. /opt/conda/lib/libhdf5.so in the ROCm image) this module produces and reads the very same files: plain datasets with
the standard little-endian types h5py picks for int32 / float32 / uint8 / float64 / int64 arrays.  For checkpoints it adds
what h5py would express as `maxshape=(None, ...)` + `dset.resize()` + partial writes: chunked, extendible datasets whose
rows are appended or overwritten in place -- still read by `f[name][:]`.

Only what the map file needs is bound: no groups, attributes, compression or strings.
"""
from __future__ import annotations

import ctypes as C
import ctypes.util
import glob
import os
from typing import Dict, Iterable, Optional

import numpy as np

_LIB = None
_HID = C.c_int64
H5F_ACC_RDONLY, H5F_ACC_RDWR, H5F_ACC_TRUNC = 0, 1, 2
H5S_SELECT_SET = 0
H5S_UNLIMITED = (1 << 64) - 1
H5T_INTEGER, H5T_FLOAT = 0, 1

# numpy dtype -> (file type symbol, memory type symbol)
_TYPES = {
    np.dtype(np.float32): ("H5T_IEEE_F32LE_g", "H5T_NATIVE_FLOAT_g"),
    np.dtype(np.float64): ("H5T_IEEE_F64LE_g", "H5T_NATIVE_DOUBLE_g"),
    np.dtype(np.int32): ("H5T_STD_I32LE_g", "H5T_NATIVE_INT32_g"),
    np.dtype(np.int64): ("H5T_STD_I64LE_g", "H5T_NATIVE_INT64_g"),
    np.dtype(np.uint8): ("H5T_STD_U8LE_g", "H5T_NATIVE_UINT8_g"),
    np.dtype(np.int8): ("H5T_STD_I8LE_g", "H5T_NATIVE_INT8_g"),
    np.dtype(np.uint16): ("H5T_STD_U16LE_g", "H5T_NATIVE_UINT16_g"),
    np.dtype(np.int16): ("H5T_STD_I16LE_g", "H5T_NATIVE_INT16_g"),
    np.dtype(np.uint32): ("H5T_STD_U32LE_g", "H5T_NATIVE_UINT32_g"),
    np.dtype(np.uint64): ("H5T_STD_U64LE_g", "H5T_NATIVE_UINT64_g"),
}


_LITTLE_ENDIAN = np.dtype(np.int32).byteorder in ("=", "<") and np.little_endian     # file types are the *LE ones: memory == file layout


class H5Error(RuntimeError):
    pass


def _candidates() -> Iterable[str]:
    env = os.environ.get("AVLMAPS_HDF5_LIB")
    if env:
        yield env
    found = ctypes.util.find_library("hdf5") or ctypes.util.find_library("hdf5_serial")
    if found:
        yield found
    for pat in ("/opt/conda/lib/libhdf5.so*", "/usr/lib/x86_64-linux-gnu/libhdf5_serial.so*", "/usr/lib/x86_64-linux-gnu/libhdf5.so*",
                "/usr/lib64/libhdf5.so*", "/usr/local/lib/libhdf5.so*"):
        for p in sorted(glob.glob(pat)):
            yield p


def available() -> bool:
    try:
        _lib()
        return True
    except H5Error:
        return False


def _lib():
    global _LIB, _HID
    if _LIB is not None:
        return _LIB
    last = None
    for cand in _candidates():
        try:
            lib = C.CDLL(cand)
            lib.H5open.restype = C.c_int
            if lib.H5open() < 0:
                raise OSError("H5open failed")
            break
        except OSError as e:
            last = e
    else:
        raise H5Error(f"the HDF5 C library (libhdf5) was not found ({last}); set AVLMAPS_HDF5_LIB or install h5py")
    maj, mnr, rel = C.c_uint(), C.c_uint(), C.c_uint()
    lib.H5get_libversion(C.byref(maj), C.byref(mnr), C.byref(rel))
    lib.version = (maj.value, mnr.value, rel.value)
    _HID = C.c_int64 if (maj.value, mnr.value) >= (1, 10) else C.c_int      # hid_t grew to 64 bits in 1.10
    hid, hsz = _HID, C.c_uint64
    sig = {
        "H5Fcreate": (hid, [C.c_char_p, C.c_uint, hid, hid]), "H5Fopen": (hid, [C.c_char_p, C.c_uint, hid]),
        "H5Fclose": (C.c_int, [hid]), "H5Fflush": (C.c_int, [hid, C.c_int]),
        "H5Screate_simple": (hid, [C.c_int, C.POINTER(hsz), C.POINTER(hsz)]), "H5Sclose": (C.c_int, [hid]),
        "H5Screate": (hid, [C.c_int]),
        "H5Sget_simple_extent_ndims": (C.c_int, [hid]),
        "H5Sget_simple_extent_dims": (C.c_int, [hid, C.POINTER(hsz), C.POINTER(hsz)]),
        "H5Sselect_hyperslab": (C.c_int, [hid, C.c_int, C.POINTER(hsz), C.POINTER(hsz), C.POINTER(hsz), C.POINTER(hsz)]),
        "H5Sselect_elements": (C.c_int, [hid, C.c_int, C.c_size_t, C.c_void_p]),
        "H5Dcreate2": (hid, [hid, C.c_char_p, hid, hid, hid, hid, hid]), "H5Dopen2": (hid, [hid, C.c_char_p, hid]),
        "H5Dclose": (C.c_int, [hid]), "H5Dget_space": (hid, [hid]), "H5Dget_type": (hid, [hid]),
        "H5Dwrite": (C.c_int, [hid, hid, hid, hid, hid, C.c_void_p]), "H5Dread": (C.c_int, [hid, hid, hid, hid, hid, C.c_void_p]),
        "H5Dset_extent": (C.c_int, [hid, C.POINTER(hsz)]),
        "H5Tget_class": (C.c_int, [hid]), "H5Tget_size": (C.c_size_t, [hid]), "H5Tget_sign": (C.c_int, [hid]), "H5Tclose": (C.c_int, [hid]),
        "H5Pcreate": (hid, [hid]), "H5Pset_chunk": (C.c_int, [hid, C.c_int, C.POINTER(hsz)]), "H5Pclose": (C.c_int, [hid]),
        "H5Pset_alloc_time": (C.c_int, [hid, C.c_int]), "H5Pset_fill_time": (C.c_int, [hid, C.c_int]),
        "H5Dget_create_plist": (hid, [hid]), "H5Pget_chunk": (C.c_int, [hid, C.c_int, C.POINTER(hsz)]), "H5Pget_layout": (C.c_int, [hid]),
        "H5Lexists": (C.c_int, [hid, C.c_char_p, hid]),
        "H5Gget_num_objs": (C.c_int, [hid, C.POINTER(hsz)]),
        "H5Gget_objname_by_idx": (C.c_ssize_t, [hid, hsz, C.c_char_p, C.c_size_t]),
        "H5Eset_auto2": (C.c_int, [hid, C.c_void_p, C.c_void_p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    lib.has_chunk_info = (hasattr(lib, "H5Dget_chunk_info") and hasattr(lib, "H5Dget_num_chunks")
                          and hasattr(lib, "H5Dget_chunk_info_by_coord"))                               # HDF5 >= 1.10.5
    if lib.has_chunk_info:
        lib.H5Dget_chunk_info_by_coord.restype = C.c_int
        lib.H5Dget_chunk_info_by_coord.argtypes = [hid, C.POINTER(hsz), C.POINTER(C.c_uint), C.POINTER(C.c_uint64), C.POINTER(hsz)]
        lib.H5Dget_num_chunks.restype, lib.H5Dget_num_chunks.argtypes = C.c_int, [hid, hid, C.POINTER(hsz)]
        lib.H5Dget_chunk_info.restype = C.c_int
        lib.H5Dget_chunk_info.argtypes = [hid, hid, hsz, C.POINTER(hsz), C.POINTER(C.c_uint), C.POINTER(C.c_uint64), C.POINTER(hsz)]
    lib.has_write_chunk = hasattr(lib, "H5Dwrite_chunk")                                                # HDF5 >= 1.10.3
    if lib.has_write_chunk:
        lib.H5Dwrite_chunk.restype = C.c_int
        lib.H5Dwrite_chunk.argtypes = [hid, hid, C.c_uint32, C.POINTER(hsz), C.c_size_t, C.c_void_p]
    lib.H5Eset_auto2(0, None, None)          # errors are reported through return codes -> H5Error, not printed
    _LIB = lib
    return lib


def _tid(sym: str) -> int:
    return _HID.in_dll(_lib(), sym).value


def _dims(seq):
    arr = (C.c_uint64 * max(1, len(seq)))(*[int(s) for s in seq])
    return arr


def _ok(rc, what):
    if rc < 0:
        raise H5Error(f"HDF5: {what} failed")
    return rc


class H5File:
    """`with H5File(path, "w") as f: f.create_dataset("grid_feat", data)` -- the subset of h5py.File the map file uses"""

    def __init__(self, path, mode: str = "r"):
        lib = _lib()
        self.path = os.fspath(path)
        b = self.path.encode()
        if mode == "w":
            self.fid = lib.H5Fcreate(b, H5F_ACC_TRUNC, 0, 0)
        elif mode in ("r", "r+", "a"):
            if mode == "a" and not os.path.exists(self.path):
                self.fid = lib.H5Fcreate(b, H5F_ACC_TRUNC, 0, 0)
            else:
                self.fid = lib.H5Fopen(b, H5F_ACC_RDONLY if mode == "r" else H5F_ACC_RDWR, 0)
        else:
            raise ValueError(f"mode {mode!r}")
        if self.fid < 0:
            raise H5Error(f"HDF5: cannot open {self.path!r} (mode {mode})")

    # ------------------------------------------------------------------ lifetime
    def close(self):
        if getattr(self, "fid", -1) >= 0:
            _lib().H5Fclose(self.fid)
            self.fid = -1

    def flush(self):
        _ok(_lib().H5Fflush(self.fid, 1), "H5Fflush")

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    __del__ = close

    # ------------------------------------------------------------------ structure
    def keys(self):
        lib = _lib()
        n = C.c_uint64()
        _ok(lib.H5Gget_num_objs(self.fid, C.byref(n)), "H5Gget_num_objs")
        out = []
        buf = C.create_string_buffer(512)
        for i in range(n.value):
            _ok(lib.H5Gget_objname_by_idx(self.fid, i, buf, 512), "H5Gget_objname_by_idx")
            out.append(buf.value.decode())
        return out

    def __contains__(self, name):
        return _lib().H5Lexists(self.fid, name.encode(), 0) > 0

    # ------------------------------------------------------------------ whole datasets
    def create_dataset(self, name: str, data: Optional[np.ndarray] = None, shape=None, dtype=None, maxshape=None, chunks=None):
        """contiguous dataset like h5py's create_dataset(name, data=a); with `maxshape` (None = unlimited along that axis) a
        chunked, extendible one (chunks default: ~1 MiB slabs of whole rows)"""
        lib = _lib()
        if data is not None:
            data = np.asarray(data, order="C")          # (ascontiguousarray would turn a 0-d array into shape (1,))
            shape, dtype = data.shape, data.dtype
        dtype = np.dtype(dtype)
        if dtype not in _TYPES:
            raise H5Error(f"dtype {dtype} is not supported by h5lite")
        ftype, mtype = (_tid(t) for t in _TYPES[dtype])
        shape = tuple(int(s) for s in shape)
        dcpl = 0
        if maxshape is not None:
            maxd = _dims([H5S_UNLIMITED if m is None else m for m in maxshape])
            space = lib.H5Screate_simple(len(shape), _dims(shape), maxd)
            if chunks is None:
                row = int(np.prod(shape[1:], dtype=np.int64)) * dtype.itemsize if len(shape) > 1 else dtype.itemsize
                chunks = (max(1, min(1 << 16, (1 << 20) // max(1, row))),) + tuple(max(1, s) for s in shape[1:])
            dcpl = lib.H5Pcreate(_tid("H5P_CLS_DATASET_CREATE_ID_g"))
            _ok(lib.H5Pset_chunk(dcpl, len(shape), _dims(chunks)), "H5Pset_chunk")
        elif len(shape) == 0:
            space = lib.H5Screate(0)                      # H5S_SCALAR: a 0-d dataset like h5py's for np.array(7)
        else:
            space = lib.H5Screate_simple(len(shape), _dims(shape), None)
        _ok(space, "H5Screate_simple")
        dset = lib.H5Dcreate2(self.fid, name.encode(), ftype, space, 0, dcpl, 0)
        if dcpl:
            lib.H5Pclose(dcpl)
        lib.H5Sclose(space)
        _ok(dset, f"H5Dcreate2({name})")
        try:
            if data is not None and data.size:
                if (dcpl and lib.has_write_chunk and _LITTLE_ENDIAN and len(shape) >= 2 and tuple(chunks[1:]) == shape[1:]
                        and data.nbytes >= self.DIRECT_CHUNK_BYTES):
                    self._write_whole_chunks(dset, name, data, int(chunks[0]))
                else:
                    _ok(lib.H5Dwrite(dset, mtype, 0, 0, 0, data.ctypes.data), f"H5Dwrite({name})")
        finally:
            lib.H5Dclose(dset)

    DIRECT_CHUNK_BYTES = 32 << 20    # chunked datasets of whole-row chunks at least this large are written chunk by chunk (below)

    def _write_whole_chunks(self, dset, name, data: np.ndarray, chunk_rows: int) -> None:
        """A large chunked dataset of whole-row chunks, written with H5Dwrite_chunk: every chunk goes from the caller's array straight to
        its place in the file (one pwrite of the library's, no filter pipeline, no pass through the chunk cache).  H5Dwrite of the same
        array copies every 128 KB chunk through the cache first: 5.2-5.5 GB/s on the GPU box, where a plain write() of the bytes runs at
        12 GB/s (profiles/r06_file_write_ceiling.txt); half of a pipeline leg's time was this save.  The bytes in the file
        are the same (a partial last chunk is padded with zeros, which is what the library's fill value gives)."""
        lib = _lib()
        n = data.shape[0]
        row_bytes = data.nbytes // n
        cb = chunk_rows * row_bytes
        off = (C.c_uint64 * data.ndim)()
        base = data.ctypes.data
        write = lib.H5Dwrite_chunk
        full = n // chunk_rows
        for i in range(full):
            off[0] = i * chunk_rows
            if write(dset, 0, 0, off, cb, base + i * cb) < 0:
                raise H5Error(f"H5Dwrite_chunk({name}, chunk {i}) failed")
        if n % chunk_rows:
            tail = np.zeros((chunk_rows,) + data.shape[1:], dtype=data.dtype)
            tail[:n - full * chunk_rows] = data[full * chunk_rows:]
            off[0] = full * chunk_rows
            if write(dset, 0, 0, off, cb, tail.ctypes.data) < 0:
                raise H5Error(f"H5Dwrite_chunk({name}, last chunk) failed")

    def create_dataset_deferred(self, name: str, shape, dtype, chunks):
        """A chunked, row-extendible dataset whose chunks are ALLOCATED now (early allocation, no fill) but not written: returns
        [(file offset, first row, rows in the chunk)] for parallel_write_chunks, which fills them with plain pwrite()s from several
        threads once the file is closed -- H5Dwrite is one thread copying through the library at 3-4 GB/s, half of every pipeline
        leg's time at the end of a build.  None if this libhdf5 cannot tell the chunk addresses (< 1.10.5): the
        caller writes through create_dataset then.  Chunks must span whole rows (chunks[1:] == shape[1:])."""
        lib = _lib()
        dtype = np.dtype(dtype)
        shape = tuple(int(s) for s in shape)
        chunks = tuple(int(c) for c in chunks)
        if not lib.has_chunk_info or dtype not in _TYPES or chunks[1:] != shape[1:] or shape[0] == 0:
            return None
        ftype = _tid(_TYPES[dtype][0])
        space = lib.H5Screate_simple(len(shape), _dims(shape), _dims((H5S_UNLIMITED,) + shape[1:]))
        dcpl = lib.H5Pcreate(_tid("H5P_CLS_DATASET_CREATE_ID_g"))
        _ok(lib.H5Pset_chunk(dcpl, len(shape), _dims(chunks)), "H5Pset_chunk")
        _ok(lib.H5Pset_alloc_time(dcpl, 1), "H5Pset_alloc_time(EARLY)")
        _ok(lib.H5Pset_fill_time(dcpl, 1), "H5Pset_fill_time(NEVER)")
        dset = lib.H5Dcreate2(self.fid, name.encode(), ftype, space, 0, dcpl, 0)
        lib.H5Pclose(dcpl)
        lib.H5Sclose(space)
        _ok(dset, f"H5Dcreate2({name})")
        try:
            n = C.c_uint64()
            fsp = lib.H5Dget_space(dset)
            _ok(lib.H5Dget_num_chunks(dset, fsp, C.byref(n)), "H5Dget_num_chunks")
            want = (shape[0] + chunks[0] - 1) // chunks[0]
            if n.value != want:
                raise H5Error(f"{name}: {n.value} chunks allocated, {want} expected")
            out = []
            off = (C.c_uint64 * len(shape))()
            mask, addr, size = C.c_uint(), C.c_uint64(), C.c_uint64()
            row_bytes = int(np.prod(shape[1:], dtype=np.int64)) * dtype.itemsize
            # by COORDINATE: a search in the chunk index; H5Dget_chunk_info(index i) walks the index from its start for every call
            # (1.10.6: 5 s for the 25 000 chunks of a 1.6 M-voxel map)
            for i in range(want):
                r0 = i * chunks[0]
                off[0] = r0
                _ok(lib.H5Dget_chunk_info_by_coord(dset, off, C.byref(mask), C.byref(addr), C.byref(size)), "H5Dget_chunk_info_by_coord")
                if mask.value or size.value != chunks[0] * row_bytes or addr.value == 0xFFFFFFFFFFFFFFFF:
                    raise H5Error(f"{name}: chunk {i} is not a plain allocated chunk")
                out.append((int(addr.value), r0, min(chunks[0], shape[0] - r0)))
            return out
        finally:
            if 'fsp' in locals() and fsp >= 0:
                lib.H5Sclose(fsp)
            lib.H5Dclose(dset)

    def _open(self, name):
        dset = _lib().H5Dopen2(self.fid, name.encode(), 0)
        if dset < 0:
            raise KeyError(name)
        return dset

    def _info(self, dset):
        lib = _lib()
        space = lib.H5Dget_space(dset)
        nd = lib.H5Sget_simple_extent_ndims(space)
        dims = (C.c_uint64 * max(1, nd))()
        if nd > 0:
            lib.H5Sget_simple_extent_dims(space, dims, None)
        lib.H5Sclose(space)
        t = lib.H5Dget_type(dset)
        cls, size, sign = lib.H5Tget_class(t), lib.H5Tget_size(t), lib.H5Tget_sign(t)
        lib.H5Tclose(t)
        if cls == H5T_FLOAT:
            dt = {4: np.float32, 8: np.float64, 2: np.float16}.get(size)
        elif cls == H5T_INTEGER:
            dt = {(1, 0): np.uint8, (1, 1): np.int8, (2, 0): np.uint16, (2, 1): np.int16, (4, 0): np.uint32, (4, 1): np.int32,
                  (8, 0): np.uint64, (8, 1): np.int64}.get((size, sign))
        else:
            dt = None
        if dt is None or np.dtype(dt) not in _TYPES:
            raise H5Error(f"dataset type class {cls} size {size} is not supported by h5lite")
        return tuple(int(d) for d in dims[:nd]), np.dtype(dt)

    def shape_dtype(self, name):
        dset = self._open(name)
        try:
            return self._info(dset)
        finally:
            _lib().H5Dclose(dset)

    def read(self, name: str) -> np.ndarray:
        """f[name][:] (and f[name][()] for scalars)"""
        lib = _lib()
        dset = self._open(name)
        try:
            shape, dt = self._info(dset)
            out = np.empty(shape, dtype=dt)
            if out.size:
                _ok(lib.H5Dread(dset, _tid(_TYPES[dt][1]), 0, 0, 0, out.ctypes.data), f"H5Dread({name})")
            return out
        finally:
            lib.H5Dclose(dset)

    __getitem__ = read

    # ------------------------------------------------------------------ partial updates of extendible datasets
    def resize(self, name: str, n_rows: int):
        lib = _lib()
        dset = self._open(name)
        try:
            shape, _ = self._info(dset)
            _ok(lib.H5Dset_extent(dset, _dims((int(n_rows),) + shape[1:])), f"H5Dset_extent({name})")
        finally:
            lib.H5Dclose(dset)

    def write_rows(self, name: str, row0: int, data: np.ndarray):
        """dset[row0:row0 + len(data)] = data (the dataset must already be large enough: resize first)"""
        lib = _lib()
        data = np.ascontiguousarray(data)
        if data.shape[0] == 0:
            return
        dset = self._open(name)
        try:
            shape, dt = self._info(dset)
            if data.dtype != dt or tuple(data.shape[1:]) != shape[1:] or row0 + data.shape[0] > shape[0]:
                raise H5Error(f"write_rows({name}): data {data.shape} {data.dtype} does not fit dataset {shape} {dt} at row {row0}")
            fspace = lib.H5Dget_space(dset)
            start = _dims((row0,) + (0,) * (len(shape) - 1))
            count = _dims(data.shape)
            _ok(lib.H5Sselect_hyperslab(fspace, H5S_SELECT_SET, start, None, count, None), "H5Sselect_hyperslab")
            mspace = lib.H5Screate_simple(data.ndim, _dims(data.shape), None)
            rc = lib.H5Dwrite(dset, _tid(_TYPES[dt][1]), mspace, fspace, 0, data.ctypes.data)
            lib.H5Sclose(mspace)
            lib.H5Sclose(fspace)
            _ok(rc, f"H5Dwrite({name}, rows)")
        finally:
            lib.H5Dclose(dset)

    def _chunk_rows(self, dset, shape) -> int:
        """rows per chunk if the dataset is chunked in whole rows, else 0"""
        lib = _lib()
        pl = lib.H5Dget_create_plist(dset)
        if pl < 0:
            return 0
        try:
            if lib.H5Pget_layout(pl) != 2:                   # H5D_CHUNKED
                return 0
            cd = (C.c_uint64 * max(1, len(shape)))()
            if lib.H5Pget_chunk(pl, len(shape), cd) != len(shape) or tuple(int(c) for c in cd)[1:] != tuple(shape[1:]):
                return 0
            return int(cd[0])
        finally:
            lib.H5Pclose(pl)

    def write_row_runs(self, name: str, runs, array: np.ndarray):
        """dset[r0:r1] = array[r0:r1] for every (r0, r1) of `runs`: the dataset is opened once and each run goes out as one
        contiguous hyperslab straight from `array` (which holds ALL rows of the dataset's current extent) -- or, when the runs are
        made of whole chunks (MapFileWriter's are) and large, chunk by chunk with H5Dwrite_chunk like a full save"""
        lib = _lib()
        array = np.ascontiguousarray(array)
        dset = self._open(name)
        try:
            shape, dt = self._info(dset)
            if array.dtype != dt or tuple(array.shape[1:]) != shape[1:] or array.shape[0] > shape[0]:
                raise H5Error(f"write_row_runs({name}): array {array.shape} {array.dtype} does not fit dataset {shape} {dt}")
            runs = [(int(a), int(b)) for a, b in runs if int(b) > int(a)]
            for r0, r1 in runs:
                if r1 > array.shape[0]:
                    raise H5Error(f"write_row_runs({name}): run [{r0}, {r1}) beyond the array")
            row_bytes = int(np.prod(shape[1:], dtype=np.int64)) * dt.itemsize if len(shape) > 1 else dt.itemsize
            if (lib.has_write_chunk and _LITTLE_ENDIAN and len(shape) >= 2 and array.shape[0] == shape[0]
                    and sum(b - a for a, b in runs) * row_bytes >= self.DIRECT_CHUNK_BYTES):
                cr = self._chunk_rows(dset, shape)
                if cr and all(a % cr == 0 and (b % cr == 0 or b == shape[0]) for a, b in runs):
                    off = (C.c_uint64 * len(shape))()
                    cb = cr * row_bytes
                    base = array.ctypes.data
                    for r0, r1 in runs:
                        for c0 in range(r0, r1, cr):
                            off[0] = c0
                            if c0 + cr <= shape[0]:
                                rc = lib.H5Dwrite_chunk(dset, 0, 0, off, cb, base + c0 * row_bytes)
                            else:                             # the partial last chunk of the dataset: padded with zeros
                                tail = np.zeros((cr,) + tuple(shape[1:]), dtype=dt)
                                tail[:shape[0] - c0] = array[c0:]
                                rc = lib.H5Dwrite_chunk(dset, 0, 0, off, cb, tail.ctypes.data)
                            if rc < 0:
                                raise H5Error(f"H5Dwrite_chunk({name}, row {c0}) failed")
                    return
            fspace = lib.H5Dget_space(dset)
            mtype = _tid(_TYPES[dt][1])
            try:
                for r0, r1 in runs:
                    start = _dims((r0,) + (0,) * (len(shape) - 1))
                    count = _dims((r1 - r0,) + shape[1:])
                    _ok(lib.H5Sselect_hyperslab(fspace, H5S_SELECT_SET, start, None, count, None), "H5Sselect_hyperslab")
                    mspace = lib.H5Screate_simple(len(shape), count, None)
                    rc = lib.H5Dwrite(dset, mtype, mspace, fspace, 0, array.ctypes.data + r0 * row_bytes)
                    lib.H5Sclose(mspace)
                    _ok(rc, f"H5Dwrite({name}, rows {r0}:{r1})")
            finally:
                lib.H5Sclose(fspace)
        finally:
            lib.H5Dclose(dset)

    def write_scattered_rows(self, name: str, rows: np.ndarray, data: np.ndarray):
        """dset[rows[i]] = data[i] for ascending, distinct `rows`: consecutive runs go out as one hyperslab each"""
        rows = np.asarray(rows, dtype=np.int64)
        if rows.size == 0:
            return
        cuts = np.flatnonzero(np.diff(rows) != 1) + 1
        start = 0
        for end in list(cuts) + [rows.size]:
            self.write_rows(name, int(rows[start]), data[start:end])
            start = end

    def write_points(self, name: str, coords: np.ndarray, values: np.ndarray):
        """dset[tuple(coords[i])] = values[i] (element selection; e.g. the few new cells of occupied_ids)"""
        lib = _lib()
        coords = np.ascontiguousarray(coords, dtype=np.uint64)
        if coords.shape[0] == 0:
            return
        dset = self._open(name)
        try:
            shape, dt = self._info(dset)
            values = np.ascontiguousarray(values, dtype=dt)
            if coords.ndim != 2 or coords.shape[1] != len(shape) or values.shape != (coords.shape[0],):
                raise H5Error(f"write_points({name}): bad coords {coords.shape} / values {values.shape}")
            fspace = lib.H5Dget_space(dset)
            _ok(lib.H5Sselect_elements(fspace, H5S_SELECT_SET, coords.shape[0], coords.ctypes.data), "H5Sselect_elements")
            mspace = lib.H5Screate_simple(1, _dims((coords.shape[0],)), None)
            rc = lib.H5Dwrite(dset, _tid(_TYPES[dt][1]), mspace, fspace, 0, values.ctypes.data)
            lib.H5Sclose(mspace)
            lib.H5Sclose(fspace)
            _ok(rc, f"H5Dwrite({name}, points)")
        finally:
            lib.H5Dclose(dset)


def parallel_write_chunks(path, chunk_list, array: np.ndarray, threads: int = 8) -> None:
    """fill the chunks create_dataset_deferred allocated (the file is CLOSED by now) from `array` (rows x ...): pwrite() of whole
    chunks from `threads` threads (os.pwrite releases the GIL); the tail of a partial last chunk is written as zeros"""
    import os
    from concurrent.futures import ThreadPoolExecutor
    array = np.ascontiguousarray(array)
    row_bytes = array.strides[0] if array.ndim > 1 else array.itemsize
    flat = array.reshape(-1).view(np.uint8)
    chunk_rows = max(c[2] for c in chunk_list)
    # early allocation lays consecutive chunks out back to back: coalesce them into runs, cut the runs into ~8 MB pieces
    runs = []
    for addr, r0, nr in chunk_list:
        if runs and runs[-1][0] + runs[-1][2] * row_bytes == addr and runs[-1][1] + runs[-1][2] == r0 and runs[-1][2] * row_bytes < (8 << 20):
            runs[-1][2] += nr
        else:
            runs.append([addr, r0, nr])
    last_addr, last_r0, last_nr = chunk_list[-1]
    fd = os.open(str(path), os.O_WRONLY)
    try:
        def put(c):
            addr, r0, nr = c
            buf = memoryview(flat[r0 * row_bytes:(r0 + nr) * row_bytes])
            done = 0
            while done < len(buf):
                done += os.pwrite(fd, buf[done:], addr + done)
        if threads <= 1 or len(runs) < 4:
            for c in runs:
                put(c)
        else:
            with ThreadPoolExecutor(max_workers=threads, thread_name_prefix="avl-h5") as ex:
                list(ex.map(put, runs))
        if last_nr < chunk_rows:          # the tail of the partial last chunk: zeros, so that equal maps give equal files
            os.pwrite(fd, bytes((chunk_rows - last_nr) * row_bytes), last_addr + last_nr * row_bytes)
    finally:
        os.close(fd)


def write_datasets(path, data: Dict[str, np.ndarray]) -> None:
    """what `with h5py.File(path, "w") as f: [f.create_dataset(k, data=v) ...]` writes"""
    with H5File(path, "w") as f:
        for k, v in data.items():
            f.create_dataset(k, data=np.asarray(v))


def read_datasets(path) -> Dict[str, np.ndarray]:
    """{k: f[k][:] for k in f.keys()}"""
    with H5File(path, "r") as f:
        return {k: f.read(k) for k in f.keys()}
