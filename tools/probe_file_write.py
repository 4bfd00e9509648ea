"""GPU box host: how fast can 3.28 GB (a 1.6 M-voxel grid_feat) reach ONE file at all?  write() / pwrite() threads / mmap + threaded copies,
in /tmp (overlay) and /dev/shm -- the ceiling for any chunk-level parallel map save.  probe_file_write.py [GB] [dirs...]"""
import mmap
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

gb = float(sys.argv[1]) if len(sys.argv) > 1 else 3.28
dirs = sys.argv[2:] or ["/tmp", "/dev/shm"]
n = int(gb * 1e9) // (8 << 20) * (8 << 20)
src = np.random.default_rng(0).integers(0, 255, n, dtype=np.uint8)
PIECE = 8 << 20
pieces = [(o, min(PIECE, n - o)) for o in range(0, n, PIECE)]


def timed(label, fn, path):
    if os.path.exists(path):
        os.unlink(path)
    t0 = time.perf_counter()
    fn(path)
    dt = time.perf_counter() - t0
    ok = os.path.getsize(path) == n
    with open(path, "rb") as f:
        f.seek(pieces[-1][0])
        ok = ok and f.read(64) == src[pieces[-1][0]:pieces[-1][0] + 64].tobytes()
    os.unlink(path)
    print(f"  {label:<44s} {dt:6.3f} s = {n / dt / 1e9:6.2f} GB/s {'ok' if ok else 'BAD'}", flush=True)


def one_write(path):
    fd = os.open(path, os.O_WRONLY | os.O_CREAT, 0o644)
    mv, done = memoryview(src), 0
    while done < n:
        done += os.write(fd, mv[done:done + (1 << 30)])
    os.close(fd)


def pwrite_threads(k, prealloc):
    def fn(path):
        fd = os.open(path, os.O_WRONLY | os.O_CREAT, 0o644)
        if prealloc:
            os.posix_fallocate(fd, 0, n)
        else:
            os.ftruncate(fd, n)
        mv = memoryview(src)

        def put(p):
            o, m = p
            d = 0
            while d < m:
                d += os.pwrite(fd, mv[o + d:o + m], o + d)
        with ThreadPoolExecutor(k) as ex:
            list(ex.map(put, pieces))
        os.close(fd)
    return fn


def mmap_threads(k, prealloc):
    def fn(path):
        fd = os.open(path, os.O_RDWR | os.O_CREAT, 0o644)
        if prealloc:
            os.posix_fallocate(fd, 0, n)
        else:
            os.ftruncate(fd, n)
        m = mmap.mmap(fd, n, mmap.MAP_SHARED, mmap.PROT_WRITE | mmap.PROT_READ)
        dst = np.frombuffer(m, dtype=np.uint8)

        def put(p):
            o, c = p
            np.copyto(dst[o:o + c], src[o:o + c])           # (releases the GIL)
        with ThreadPoolExecutor(k) as ex:
            list(ex.map(put, pieces))
        del dst
        m.close()
        os.close(fd)
    return fn


print(f"{n / 1e9:.2f} GB, {os.cpu_count()} cpus")
for d in dirs:
    if not os.path.isdir(d):
        continue
    path = os.path.join(d, "avl_probe_write.bin")
    print(d, flush=True)
    for rep in range(2):
        timed("write() from one thread", one_write, path)
    for k in (1, 4, 16):
        timed(f"pwrite, {k} threads, ftruncate", pwrite_threads(k, False), path)
    timed("pwrite, 8 threads, posix_fallocate", pwrite_threads(8, True), path)
    for k in (1, 4, 8, 16, 32):
        timed(f"mmap + copies, {k} threads, ftruncate", mmap_threads(k, False), path)
    timed("mmap + copies, 16 threads, posix_fallocate", mmap_threads(16, True), path)
