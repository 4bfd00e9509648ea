"""Host-side geometry + map I/O with the names and argument meaning of the reference's
avlmaps/utils/mapping_utils.py (cited per function, path:line in the upstream repo).

Everything here is tiny float64 host math (one 4x4 per frame) or file I/O; per-point work runs in the
HIP builder kernels.
"""
from __future__ import annotations

import os

from pathlib import Path
from typing import List, Optional, Sequence

import numpy as np
from scipy.spatial.transform import Rotation as R

try:  # the reference's on-disk format is HDF5, written with h5py
    import h5py  # type: ignore
except Exception:  # pragma: no cover - absent in the build container
    h5py = None
from . import h5lite   # the same files through the HDF5 C library (ctypes) where h5py is not installed

MAP_DATASETS = ("mapped_iter_list", "grid_feat", "grid_pos", "weight", "occupied_ids", "grid_rgb")


def cvt_pose_vec2tf(pos_quat_vec: np.ndarray) -> np.ndarray:
    """(px, py, pz, qx, qy, qz, qw) -> 4x4 float64.  Reference: mapping_utils.py:18-26."""
    v = np.asarray(pos_quat_vec, dtype=np.float64).flatten()
    tf = np.eye(4)
    tf[:3, 3] = v[:3]
    tf[:3, :3] = R.from_quat(v[3:]).as_matrix()
    return tf


def get_sim_cam_mat(h: int, w: int) -> np.ndarray:
    """90-degree-FOV pinhole matrix of an (h, w) image.  Reference: mapping_utils.py:591-596."""
    m = np.eye(3)
    m[0, 0] = m[1, 1] = w / 2.0
    m[0, 2] = w / 2.0
    m[1, 2] = h / 2.0
    return m


def base_pos2grid_id_3d(gs: int, cs: float, x_base: float, y_base: float, z_base: float) -> List[int]:
    """Scalar host version for callers such as the Habitat dataloader.  Reference: mapping_utils.py:345-349."""
    return [int(gs / 2 - int(x_base / cs)), int(gs / 2 - int(y_base / cs)), int(z_base / cs)]


def grid_id2base_pos_3d(row: int, col: int, height: int, cs: float, gs: int):
    """Inverse of base_pos2grid_id_3d (cell origin).  Reference: mapping_utils.py (grid_id2base_pos_3d)."""
    return (gs / 2 - row) * cs, (gs / 2 - col) * cs, height * cs


def load_depth_npy(depth_filepath) -> np.ndarray:
    """depth/*.npy holds float32 metres (H, W).  Reference: dataset/README.md:76-93."""
    with open(depth_filepath, "rb") as f:
        return np.load(f)


def load_rgb_png(rgb_filepath) -> np.ndarray:
    """RGB uint8 (H, W, 3).  The reference does cv2.imread + BGR2RGB (vlmap_builder.py:118-119)."""
    from PIL import Image
    with Image.open(rgb_filepath) as im:
        return np.asarray(im.convert("RGB"), dtype=np.uint8)


def _npz_path(path) -> Path:
    p = Path(path)
    return p.with_name(p.name + ".npz")


def hdf5_backend() -> Optional[str]:
    """'h5py' | 'h5lite' (libhdf5 through ctypes) | None"""
    if h5py is not None:
        return "h5py"
    return "h5lite" if h5lite.available() else None


def write_map_datasets(save_path, data: dict) -> None:
    """{name: array} -> one HDF5 file (h5py, else libhdf5 through ctypes); `<save_path>.npz` only if neither exists"""
    backend = hdf5_backend()
    if backend == "h5py":
        with h5py.File(save_path, "w") as f:
            for k, v in data.items():
                f.create_dataset(k, data=v)
    elif backend == "h5lite":
        h5lite.write_datasets(save_path, data)
    else:
        np.savez(_npz_path(save_path), **data)


def read_map_datasets(map_path) -> dict:
    """every dataset of a map file as {name: array}; an .h5df without an HDF5 backend fails loudly"""
    if Path(map_path).exists():
        backend = hdf5_backend()
        if backend == "h5py":
            with h5py.File(map_path, "r") as f:
                return {k: f[k][()] for k in f.keys()}
        if backend == "h5lite":
            return h5lite.read_datasets(map_path)
        raise RuntimeError(f"{map_path} is an HDF5 map but neither h5py nor the HDF5 C library (libhdf5) is available")
    with np.load(_npz_path(map_path)) as z:
        return {k: z[k] for k in z.files}


def save_3d_map(save_path, grid_feat, grid_pos, weight, occupied_ids, mapped_iter_list, grid_rgb=None,
                init_height_id=None) -> None:
    """Write the datasets of the reference's map file, same names / dtypes / shapes (mapping_utils.py:469-505): an HDF5
    file through h5py, or through the HDF5 C library (utils/h5lite.py) where h5py is missing.  Only if neither exists are the
    same arrays written to `<save_path>.npz`."""
    data = dict(mapped_iter_list=np.array(sorted(mapped_iter_list), dtype=np.int32), grid_feat=np.asarray(grid_feat),
                grid_pos=np.asarray(grid_pos), weight=np.asarray(weight), occupied_ids=np.asarray(occupied_ids))
    if init_height_id is not None:
        data["init_height_id"] = np.array(init_height_id, dtype=np.int32)
    if grid_rgb is not None:
        data["grid_rgb"] = np.asarray(grid_rgb)
    backend = hdf5_backend()
    if backend == "h5py":
        with h5py.File(save_path, "w") as f:
            for k, v in data.items():
                f.create_dataset(k, data=v)
    elif backend == "h5lite":
        h5lite.write_datasets(save_path, data)
    else:
        np.savez(_npz_path(save_path), **data)


class MapFileWriter:
    """Checkpointing writer of one map file.  The reference rewrites the whole file every 100 frames (vlmap_builder.py:180-183:
    O(map) per checkpoint); voxel ids never change once assigned and new voxels are appended, so after the first save only
    the rows that changed (row_dirty from VoxelAccumulator.finalize(want_dirty=True)) and the new rows are written, plus the few
    new cells of occupied_ids.  The file stays the reference's layout -- the same six dataset names / dtypes, read back by
    `f[name][:]` -- with the per-voxel datasets chunked and extendible instead of contiguous.  Needs the HDF5 C library
    (utils/h5lite.py); otherwise every save is a full rewrite through save_3d_map."""

    ROW_SETS = ("grid_feat", "grid_pos", "weight", "grid_rgb")
    FEAT_CHUNK_ROWS = 64  # rows per HDF5 chunk of grid_feat (128 KB at D = 512).  An incremental save only ever writes WHOLE chunks
                          # of it (from the host mirror): a partial-chunk write makes the library read the chunk first, and with
                          # the 1 MiB chunks of the first version a checkpoint that changed 20 % of the rows cost twice a full rewrite
    MAX_RUNS = 2048      # one contiguous file write per run and dataset: scattered dirty rows are coalesced across small gaps
                         # (the rows in between are rewritten with their unchanged values) so that a checkpoint never costs
                         # more calls than this -- worst case (dirty rows everywhere) it degenerates to one full rewrite

    @classmethod
    def row_runs(cls, dirty_old: np.ndarray, n_old: int, n: int, max_runs=None):
        """[(r0, r1)] covering every dirty row below n_old and all rows [n_old, n), at most max_runs (default MAX_RUNS) runs"""
        max_runs = cls.MAX_RUNS if max_runs is None else int(max_runs)
        d = np.flatnonzero(dirty_old)
        starts, ends = [], []
        if d.size:
            cut = np.flatnonzero(np.diff(d) != 1) + 1
            starts = d[np.concatenate([[0], cut])]
            ends = d[np.concatenate([cut - 1, [d.size - 1]])] + 1
            if n > n_old and ends[-1] == n_old:            # the last dirty run touches the appended block
                ends[-1] = n
            elif n > n_old:
                starts, ends = np.append(starts, n_old), np.append(ends, n)
        elif n > n_old:
            starts, ends = np.array([n_old]), np.array([n])
        starts, ends = np.asarray(starts, dtype=np.int64), np.asarray(ends, dtype=np.int64)
        if starts.size > max_runs:
            gaps = starts[1:] - ends[:-1]
            thr = np.partition(gaps, starts.size - max_runs)[starts.size - max_runs]     # merge every gap <= thr
            keep = np.concatenate([[True], gaps > thr])
            starts = starts[keep]
            ends = np.concatenate([ends[:-1][keep[1:]], ends[-1:]])
        return [(int(a), int(b)) for a, b in zip(starts, ends)]

    MARKER = "checkpoint_complete"   # int32[1]: 0 while an in-place patch is under way (an extra dataset upstream's loader ignores)

    def __init__(self, path):
        self.path = Path(path)
        self.n_saved = None          # rows in the file, None = nothing written by this writer yet
        self.write_threads = 1       # > 1: full saves fill grid_feat's early-allocated chunks with pwrite()s from that many threads.  OFF: asking
                                     # libhdf5 1.10 where 25 000 chunks lie costs seconds (every H5Dget_chunk_info* call walks the index), and one
                                     # thread's plain write() already runs at 12 GB/s on the GPU box.  Full saves go chunk by chunk through
                                     # H5Dwrite_chunk instead (h5lite.H5File._write_whole_chunks; profiles/r06_parallel_save_probe.txt)
        self.stats = []              # per save: dict(mode, rows_written, rows_total)
        self.mirror = None           # host copy of the per-voxel datasets as last saved (adopted from a full save, patched by
                                     # save_packed): what lets a checkpoint ship only its changed rows across PCIe
        self.mirror_occ = None       # ... and of occupied_ids

    def save(self, arrays, mapped_iter_list, row_dirty=None) -> None:
        import time
        t0 = time.perf_counter()
        try:
            self._save(arrays, mapped_iter_list, row_dirty)
        finally:
            if self.stats:
                self.stats[-1].setdefault("seconds", time.perf_counter() - t0)

    def _save(self, arrays, mapped_iter_list, row_dirty=None) -> None:
        n = int(arrays["grid_pos"].shape[0])
        iters = np.array(sorted(mapped_iter_list), dtype=np.int32)
        incremental = (h5lite.available() and self.n_saved is not None and row_dirty is not None and n >= self.n_saved
                       and self.path.exists())
        if not h5lite.available():
            save_3d_map(self.path, arrays["grid_feat"], arrays["grid_pos"], arrays["weight"], arrays["occupied_ids"], iters,
                        arrays["grid_rgb"])
            self.stats.append(dict(mode="full (no libhdf5)", rows_written=n, rows_total=n))
            self.n_saved = n
            return
        if not incremental:
            # a full save never leaves a half-written map behind: written next to the target, then renamed over it
            tmp = self.path.with_name(self.path.name + ".tmp")
            deferred = None
            with h5lite.H5File(tmp, "w") as f:
                f.create_dataset("mapped_iter_list", data=iters, maxshape=(None,))
                for k in self.ROW_SETS:
                    a = np.asarray(arrays[k])
                    chunks = (self.FEAT_CHUNK_ROWS,) + a.shape[1:] if k == "grid_feat" else None
                    if k == "grid_feat" and self.write_threads > 1 and a.nbytes >= (64 << 20):
                        # 99 % of the file: its chunks are allocated here and filled below by plain pwrite()s from several threads
                        deferred = f.create_dataset_deferred(k, a.shape, a.dtype, chunks)
                        if deferred is not None:
                            continue
                    f.create_dataset(k, data=a, maxshape=(None,) + a.shape[1:], chunks=chunks)
                f.create_dataset("occupied_ids", data=np.asarray(arrays["occupied_ids"]))
                f.create_dataset(self.MARKER, data=np.ones(1, np.int32))
            if deferred is not None:
                h5lite.parallel_write_chunks(tmp, deferred, np.asarray(arrays["grid_feat"]), self.write_threads)
            os.replace(tmp, self.path)
            self.stats.append(dict(mode="full", rows_written=n, rows_total=n))
            self.n_saved = n
            self._adopt(arrays)
            return
        n_old = self.n_saved
        runs = self._patch_file(n_old, n, np.asarray(row_dirty[:n_old]), {k: np.asarray(arrays[k]) for k in self.ROW_SETS}, iters)
        self.stats.append(dict(mode="incremental", rows_written=int(sum(b - a for a, b in runs)), rows_total=n,
                               rows_dirty=int(np.count_nonzero(row_dirty[:n_old])) + n - n_old, runs=len(runs)))
        self.n_saved = n
        self._adopt(arrays)

    def _patch_file(self, n_old: int, n: int, dirty_old: np.ndarray, src, iters):
        """Bring the file from n_old rows to the n rows of `src` (arrays holding the WHOLE current map): grid_feat -- 99 % of the
        bytes -- is written in runs of whole chunks that contain a changed or new row, the small per-voxel datasets (19 B per voxel
        together) are rewritten, occupied_ids gets its new cells.  -> the grid_feat runs [(r0, r1)]"""
        C = self.FEAT_CHUNK_ROWS
        full_old = n_old // C                                   # chunks that lie completely below the old end of the file
        cd = np.zeros(full_old, np.uint8)
        d = np.flatnonzero(dirty_old[: full_old * C])
        if d.size:
            cd[np.unique(d // C)] = 1
        n_chunks = (n + C - 1) // C
        if n == n_old and not np.any(dirty_old[full_old * C:]):
            n_chunks = full_old                                 # nothing new and the partial last chunk is clean
        cruns = self.row_runs(cd, full_old, n_chunks, self.MAX_RUNS)
        runs = [(a * C, min(b * C, n)) for a, b in cruns]
        with h5lite.H5File(self.path, "r+") as f:
            # an in-place patch is not atomic: the marker is cleared (and flushed) first and set again last, so that a run resumed
            # after a crash in between can tell that mapped_iter_list and the feature rows may not belong together
            marked = self.MARKER in f
            if marked:
                f.write_rows(self.MARKER, 0, np.zeros(1, np.int32))
                f.flush()
            for k in self.ROW_SETS:
                f.resize(k, n)
                if k == "grid_feat":
                    f.write_row_runs(k, runs, src[k][:n])
                elif n:
                    f.write_rows(k, 0, src[k][:n])
            if n > n_old:
                # the new cells of occupied_ids: whole grid rows (gs * vh cells, contiguous in the file) that contain one, from the
                # host mirror of the grid -- a point selection of 100 k scattered cells costs seconds in the HDF5 library
                new_pos = np.asarray(src["grid_pos"][n_old:n])
                self.mirror_occ[new_pos[:, 0], new_pos[:, 1], new_pos[:, 2]] = np.arange(n_old, n, dtype=np.int32)
                touched = np.unique(new_pos[:, 0])
                cut = np.flatnonzero(np.diff(touched) != 1) + 1
                for a, b in zip(np.concatenate([[0], cut]), np.concatenate([cut, [touched.size]])):
                    r0, r1 = int(touched[a]), int(touched[b - 1]) + 1
                    f.write_rows("occupied_ids", r0, self.mirror_occ[r0:r1])
            f.resize("mapped_iter_list", len(iters))
            f.write_rows("mapped_iter_list", 0, iters)
            if marked:
                f.flush()
                f.write_rows(self.MARKER, 0, np.ones(1, np.int32))
        return runs

    def _adopt(self, arrays) -> None:
        """the arrays of a save that had the whole map on the host become the mirror (no copy)"""
        self.mirror = {k: np.asarray(arrays[k]) for k in self.ROW_SETS}
        self.mirror_occ = np.asarray(arrays["occupied_ids"])


    def save_packed(self, lean, mapped_iter_list) -> None:
        """incremental save from VoxelAccumulator.finalize_rows(self.n_saved): only the changed and the new rows reached the
        host.  They are folded into the writer's host mirror of the map (the arrays of the last full save: what the reference
        keeps as its working state), and the file is patched from the mirror in at most MAX_RUNS contiguous runs."""
        import time
        t0 = time.perf_counter()
        n, idx, rows = int(lean["n"]), np.asarray(lean["idx"], dtype=np.int64), lean["rows"]
        n_old = self.n_saved
        if n_old is None or self.mirror is None or not h5lite.available() or n < n_old or not self.path.exists():
            raise RuntimeError("MapFileWriter.save_packed needs a file this writer has already written in full")
        if int(lean.get("n_saved", n_old)) != n_old:
            raise RuntimeError(f"finalize_rows was called for {lean.get('n_saved')} saved rows, the file holds {n_old}")
        iters = np.array(sorted(mapped_iter_list), dtype=np.int32)
        for k in self.ROW_SETS:
            m = self.mirror[k]
            if n > m.shape[0]:                               # grow like the reference's arrays: double, copy once
                grown = np.empty((max(n, 2 * m.shape[0]),) + m.shape[1:], m.dtype)
                grown[:n_old] = m[:n_old]
                m = self.mirror[k] = grown
            m[idx] = rows[k]
        dirty = np.zeros(n_old, np.uint8)
        dirty[idx[idx < n_old]] = 1
        t_mirror = time.perf_counter() - t0
        runs = self._patch_file(n_old, n, dirty, self.mirror, iters)
        self.stats.append(dict(mode="incremental", rows_written=int(sum(b - a for a, b in runs)), rows_total=n, rows_dirty=int(idx.size),
                               runs=len(runs), host_bytes=int(sum(np.asarray(v).nbytes for v in rows.values())),
                               seconds=time.perf_counter() - t0, mirror_seconds=t_mirror))
        self.n_saved = n

    def current_map(self):
        """the writer's host mirror as the dict a full finalize() returns (views of the n_saved rows)"""
        if self.mirror is None or self.n_saved is None:
            return None
        n = self.n_saved
        out = {k: self.mirror[k][:n] for k in self.ROW_SETS}
        out["occupied_ids"] = self.mirror_occ
        return out


def read_map_dataset(map_path, name: str):
    """one dataset of a map file (e.g. mapped_iter_list: what the ranks of a resumed multi-GPU build need, not the 4 GB of features)"""
    if Path(map_path).exists():
        backend = hdf5_backend()
        if backend == "h5py":
            with h5py.File(map_path, "r") as f:
                return f[name][()] if name in f else None
        if backend == "h5lite":
            with h5lite.H5File(map_path, "r") as f:
                return f.read(name) if name in f else None
        raise RuntimeError(f"{map_path} is an HDF5 map but neither h5py nor the HDF5 C library (libhdf5) is available")
    with np.load(_npz_path(map_path)) as z:
        return z[name] if name in z.files else None


def map_checkpoint_complete(map_path) -> bool:
    """False if the file's last in-place checkpoint was interrupted (MapFileWriter.MARKER == 0): its mapped_iter_list may be
    older than its rows.  Files without the marker (upstream's, full saves of other writers) count as complete."""
    try:
        m = read_map_dataset(map_path, MapFileWriter.MARKER)
    except Exception:
        return True
    return m is None or int(np.asarray(m).ravel()[0]) != 0


def map_file_exists(map_path) -> bool:
    return Path(map_path).exists() or _npz_path(map_path).exists()


def load_3d_map(map_path):
    """-> (mapped_iter_list, grid_feat, grid_pos, weight, occupied_ids, grid_rgb[, init_height_id]).
    Reference: mapping_utils.py:508-541."""
    d = read_map_datasets(map_path)
    out = (d["mapped_iter_list"].tolist(), d["grid_feat"], d["grid_pos"], d["weight"], d["occupied_ids"], d.get("grid_rgb"))
    if "init_height_id" in d:
        return out + (d["init_height_id"],)
    return out
