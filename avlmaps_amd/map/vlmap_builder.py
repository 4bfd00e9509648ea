"""VLMapBuilder with the reference's interface (avlmaps/map/vlmap_builder.py:35-327), fusing frames on the MI355X.

Per frame the reference runs LSeg, back-projects ~7.8 k sampled pixels and updates the map in a Python loop
(vlmap_builder.py:102-183).  Here the per-point loop is three HIP launches (avl_builder_integrate_frame); the host
keeps only what is inherently sequential and tiny: the float64 pose chain (one 4x4 per frame) and the reference's
sampling order (np.random.shuffle on the GLOBAL NumPy state, so a seeded run samples the same pixels upstream and here --
with several ranks too: rank r first draws and discards the shuffles of the frames before its shard, see shard_sampling).

Feature extraction stays on PyTorch-ROCm: `feat_extractor(rgb_uint8_hwc) -> (Hf, Wf, D) float32 CUDA tensor`
(channels-last, stays on the device).  A reference-style (1, D, Hf, Wf) array is accepted and transposed.
With torch.distributed initialised (one process per GPU) frames are sharded contiguously over ranks and merged with one
sparse, row-sharded exchange (avlmaps_amd.parallel.merge_accumulator_sharded; merge_mode = "reduce" selects the dense single
RCCL reduce instead); every rank keeps its block of finished rows on the device (map_shard: what VLMap.shard_index_rows
scores), rank 0 gathers the float32 rows and writes the map file -- at the end and, like upstream's loop, every save_every
frames (per rank), so that a long multi-GPU build can be resumed.  A seeded N-rank build gives the map of the seeded
single-process build: grid_pos / occupied_ids / grid_rgb / weight bit-exact, grid_feat to float64 rounding.
"""
from __future__ import annotations

import os
from pathlib import Path
from typing import Callable, List, Optional

import numpy as np

from .. import ops, parallel
from ..utils.mapping_utils import (MapFileWriter, cvt_pose_vec2tf, load_3d_map, load_depth_npy, load_rgb_png, map_checkpoint_complete,
                                   map_file_exists, read_map_dataset, save_3d_map)
from .map import cfg_get


class VLMapBuilder:
    def __init__(self, data_dir: Path, map_config, pose_path: Path, rgb_paths: List[Path], depth_paths: List[Path],
                 base2cam_tf: np.ndarray, base_transform: np.ndarray, feat_extractor: Optional[Callable] = None):
        self.data_dir = Path(data_dir)
        self.pose_path = pose_path
        self.rgb_paths = rgb_paths
        self.depth_paths = depth_paths
        self.map_config = map_config
        self.base2cam_tf = base2cam_tf
        self.base_transform = base_transform
        self.feat_extractor = feat_extractor
        self.save_every = 100                      # vlmap_builder.py:181
        self.capacity = None                       # initial voxel capacity (default gs*gs like upstream; creating the 6 GB of accumulators
                                                   # takes < 4 ms on the device); doubles on demand
        self.max_capacity = None                   # growth limit (default: every cell of the grid); 0 = fixed capacity
        self.min_depth, self.max_depth = 0.1, 6    # vlmap_builder.py:129
        self.sigma_sq = 0.6                        # vlmap_builder.py:157
        self.exact_rgb = True                      # replay weight / grid_rgb sequentially at finalisation
        self.batch_frames = 1                      # >1: fuse that many frames per launch pair (same map, fewer launches)
        self.deferred_fuse = "auto"                # frame-by-frame runs: ONE launch per frame (the feature fusion of frame i runs
                                                   # inside the launch of frame i + 1: 22 -> 13 us per frame, the same map bit for
                                                   # bit).  The features of a frame are then read one call late, so the extractor
                                                   # must hand out a NEW tensor per frame.  "auto" (default) finds out: the first
                                                   # frames are fused at once while the builder holds on to each frame's feature
                                                   # tensor; an extractor that still returns the same storage refills one buffer
                                                   # -> stays frame-at-once; one that returns fresh storage (any torch model
                                                   # does: the caching allocator cannot reuse memory that is still referenced) ->
                                                   # deferral is switched on from the third frame.  True / False force it.
        self.frame_loop_frames = 4                 # frame-by-frame runs: whenever further frames are already staged (prefetch threads
                                                   # ahead of the fusing thread) up to this many consecutive frames go to the library
                                                   # in ONE call (avl_builder_integrate_frames: the frame loop in C -- still one launch
                                                   # (pair) per frame, the same map bit for bit, but frame i + 1's map-independent half
                                                   # of K1 runs inside frame i's launch and the per-call Python cost is paid once).
                                                   # Needs device-resident features whose storage the extractor does not recycle while
                                                   # they wait (found out like deferred_fuse "auto" does); 1 = one call per frame
        self.frame_loop_max_extract_s = 1e-3       # ... and only while the extractor call returns within this time
        self.frame_loop_eager = False              # True: hold frames until frame_loop_frames are together even if the next one is not staged yet
        self.prefetch_frames = 4                   # frames decoded ahead by host threads (0 = load inline like upstream)
        self.stage_frames = True                   # with prefetch_frames > 0: depth / rgb / sample lists travel through page-locked
                                                   # buffers on a copy stream (device.FrameStager) instead of three pageable,
                                                   # synchronous copies per frame on the fusing thread
        self.skip_busy_checkpoints = True          # a periodic checkpoint that comes due while the previous one is still being
                                                   # written is skipped (the next one carries its rows): the build is never
                                                   # throttled to the disk's speed; the final save always happens.  False = every
                                                   # save_every frames like upstream, waiting for the writer if need be
        self.skip_mapped_frames = False            # True: a resumed run skips the frames listed in the map file's
                                                   # mapped_iter_list (upstream restores the list but re-fuses every frame)
        self.incremental_checkpoints = True        # periodic saves write only the rows that changed + the new rows
                                                   # (utils.mapping_utils.MapFileWriter); False = full rewrite like upstream
        self.pixel_sampling = "reference"          # "reference" (default, also with several ranks): np.random.shuffle(arange(H*W))[::rate] on the
                                                   # global RNG -- the pixels a seeded upstream / single-process run samples, so that N ranks
                                                   # == 1 rank out of the box.  Its draws are serial by nature (0.3-0.5 ms per
                                                   # 720x1080 frame through avl_mt19937_skip_shuffles, see sampler_workers; NumPy's shuffle
                                                   # takes 6.6 ms), and rank r of a sharded build first fast-forwards the RNG past the frames
                                                   # of the ranks before it (~10 s for the last of 8 ranks of a 40 k-frame build; a line on
                                                   # stdout says so).  "uniform" (opt-in, for speed): the same distribution -- an ordered
                                                   # uniform sample without replacement -- from a per-frame generator seeded by ONE draw of
                                                   # the global RNG and the frame index (0.25 ms; not the reference's pixels, but
                                                   # reproducible under np.random.seed and independent of how the frames are sharded).
                                                   # "auto": "reference" in a single process, "uniform" with several ranks.  The resolved
                                                   # mode is recorded in build_times["pixel_sampling"]
        self.sampler_workers = "auto"              # "reference" sampling with prefetch_frames > 0: the serial part of a frame's
                                                   # shuffle is only its DRAWS (how far the RNG moves: avl_mt19937_skip_shuffles,
                                                   # half the cost of the sample).  The sampler thread walks the RNG frame by frame
                                                   # and hands a snapshot of the state to this many worker threads, which compute
                                                   # the pixel lists of different frames side by side (same lists, same final RNG
                                                   # state).  "auto" = 3 when the host has >= 6 cores, else 0 (= sample in the
                                                   # sampler thread as before)
        self.merge_mode = "sharded"                # several ranks: "sharded" = one all_to_all of every rank's OWN voxel rows to the
                                                   # owners of their final rows, finalised where they land (bytes ~ what a rank
                                                   # holds; the map stays row-sharded for the index kernels); "reduce" = ONE
                                                   # sum-reduce of a dense (M, D + 4) float64 buffer to rank 0 (north_star's wording;
                                                   # 9.3 GB per rank at 2.25 M voxels)
        self.map_shard = None                      # after a multi-rank build: this rank's block of the merged map, device tensors
                                                   # (parallel.merge_accumulator_sharded) -- VLMap.adopt_device_shard takes it as is
        self.shard_sampling = "replay"             # several ranks: "replay" = every rank first consumes the global NumPy RNG
                                                   # exactly as the frames before its shard would have (one discarded shuffle
                                                   # per skipped frame, ~6 ms each at 720x1080), so that a seeded N-rank run
                                                   # samples the pixels of the seeded single-process / reference run;
                                                   # "independent" = start at once from the rank's own RNG state (an
                                                   # unseeded production run: statistically the same map, not the same pixels)

    # ------------------------------------------------------------------ pose chain (host, float64)
    def frame_transforms(self, base_poses: np.ndarray) -> List[np.ndarray]:
        """pc_transform per frame.  Reference: vlmap_builder.py:64-76, :106-108, :133 (same matmul order)."""
        self.init_base_tf = self.base_transform @ cvt_pose_vec2tf(base_poses[0]) @ np.linalg.inv(self.base_transform)
        self.inv_init_base_tf = np.linalg.inv(self.init_base_tf)
        self.init_cam_tf = self.init_base_tf @ self.base2cam_tf
        self.inv_init_cam_tf = np.linalg.inv(self.init_cam_tf)
        out = []
        for posevec in base_poses:
            habitat_base_pose = cvt_pose_vec2tf(posevec)
            base_pose = self.base_transform @ habitat_base_pose @ np.linalg.inv(self.base_transform)
            tf = self.inv_init_base_tf @ base_pose
            out.append(tf @ self.base_transform @ self.base2cam_tf)
        return out

    @staticmethod
    def sample_pixels(n_pix: int, depth_sample_rate: int) -> np.ndarray:
        """shuffle_mask[::rate] on the global NumPy RNG.  Reference: vlmap_builder.py:275-277.  The permutation and the state
        the RNG is left in are NumPy's; the work is done by avl_mt19937_shuffle_sample (host C in the library: int32 indices,
        branch-free rejection sampling) because this call is the serial part of a pixel-faithful build."""
        st = np.random.get_state()
        if st[0] == "MT19937" and 0 < n_pix < (1 << 31):
            try:
                import ctypes as C
                from .. import _lib
                lib = _lib.load()
            except Exception:
                lib = None
            if lib is not None:
                key = np.ascontiguousarray(st[1], dtype=np.uint32).copy()
                pos = C.c_int(int(st[2]))
                scratch = _SCRATCH.__dict__.get("buf")          # per thread: a fresh 3 MB array per frame is 0.2 ms of page faults
                if scratch is None or scratch.shape[0] < n_pix:
                    scratch = _SCRATCH.buf = np.empty(n_pix, np.int32)
                out = np.empty((n_pix + depth_sample_rate - 1) // depth_sample_rate, np.int32)
                _lib.check(lib.avl_mt19937_shuffle_sample(key.ctypes.data, C.byref(pos), int(n_pix), int(depth_sample_rate),
                                                          scratch.ctypes.data, out.ctypes.data), "avl_mt19937_shuffle_sample")
                np.random.set_state((st[0], key, pos.value, st[3], st[4]))
                return out
        shuffle_mask = np.arange(n_pix)
        np.random.shuffle(shuffle_mask)
        return shuffle_mask[::depth_sample_rate].astype(np.int32)

    @staticmethod
    def _sample_from_state(key: np.ndarray, pos: int, n_pix: int, depth_sample_rate: int) -> np.ndarray:
        """sample_pixels for an explicit MT19937 state (a worker's private copy: the global RNG is not touched)"""
        import ctypes as C
        from .. import _lib
        lib = _lib.load()
        cpos = C.c_int(int(pos))
        scratch = _SCRATCH.__dict__.get("buf")
        if scratch is None or scratch.shape[0] < n_pix:
            scratch = _SCRATCH.buf = np.empty(n_pix, np.int32)
        out = np.empty((n_pix + depth_sample_rate - 1) // depth_sample_rate, np.int32)
        _lib.check(lib.avl_mt19937_shuffle_sample(key.ctypes.data, C.byref(cpos), int(n_pix), int(depth_sample_rate), scratch.ctypes.data,
                                                  out.ctypes.data), "avl_mt19937_shuffle_sample")
        return out

    def _n_sampler_workers(self) -> int:
        w = self.sampler_workers
        if w == "auto":
            import os
            try:
                cores = len(os.sched_getaffinity(0))
            except Exception:
                cores = os.cpu_count() or 1
            return 3 if cores >= 6 else 0
        return max(0, int(w or 0))

    @staticmethod
    def _announce_skip(n_frames: int, n_pix: int) -> None:
        est = n_frames * n_pix * 0.7e-9                          # 0.4-0.7 ns per index of a skipped shuffle (AVX-512 / AVX2)
        if est > 5.0:
            print(f"[avlmaps_amd] fast-forwarding the NumPy RNG past {n_frames} frames of the ranks before this one (~{est:.0f} s) so "
                  "that this seeded run samples the reference's pixels; pixel_sampling='uniform' or shard_sampling='independent' "
                  "start at once (same distribution, other pixels)", flush=True)

    def _resolve_pixel_sampling(self) -> None:
        """pixel_sampling = "auto" (opt-in; the default is "reference") becomes "reference" in a single process -- the pixels of a seeded upstream run,
        vlmap_builder.py:275-277 -- and "uniform" with several ranks, where the reference's one serial random stream would make
        the last rank fast-forward past every other rank's frames before its first one (profiles/HISTORY.md 5)."""
        if self.pixel_sampling != "auto":
            return
        rank, ws = _dist_rank_ws()
        self.pixel_sampling = "uniform" if ws > 1 else "reference"
        if ws > 1 and rank == 0:
            print(f"[avlmaps_amd] {ws} ranks: pixel_sampling defaults to 'uniform' (per-frame generators seeded by one draw of the "
                  "global NumPy RNG: reproducible, independent of the sharding, no serial RNG fast-forward); set "
                  "pixel_sampling='reference' for the pixels a seeded single-process / upstream run samples", flush=True)

    def _draw_samples(self, frame_i: int, n_pix: int, depth_sample_rate: int) -> np.ndarray:
        if self.pixel_sampling == "reference":
            return self.sample_pixels(n_pix, depth_sample_rate)
        k = (n_pix + depth_sample_rate - 1) // depth_sample_rate
        gen = np.random.default_rng([self._uniform_seed, int(frame_i)])
        return gen.choice(n_pix, size=k, replace=False, shuffle=True).astype(np.int32)

    @staticmethod
    def skip_pixel_shuffles(n_frames: int, n_pix: int) -> None:
        """Advance the global NumPy RNG as `n_frames` calls of sample_pixels(n_pix, .) would (the draws of a shuffle depend
        only on the array length): what rank r > 0 does for the frames [0, lo) of the other ranks' shards, so that its first
        frame samples the pixels it samples in the single-process run.  All frames are taken to have n_pix pixels."""
        if n_frames <= 0:
            return
        st = np.random.get_state()
        if st[0] == "MT19937" and n_pix < (1 << 32):
            # the draws of a shuffle without the shuffle (avl_mt19937_skip_shuffles): ~4x faster than permuting a scratch array
            import ctypes as C
            from .. import _lib
            key = np.ascontiguousarray(st[1], dtype=np.uint32).copy()
            pos = C.c_int(int(st[2]))
            _lib.check(_lib.load().avl_mt19937_skip_shuffles(key.ctypes.data, C.byref(pos), int(n_pix), int(n_frames)),
                       "avl_mt19937_skip_shuffles")
            np.random.set_state((st[0], key, pos.value, st[3], st[4]))
            return
        scratch = np.arange(n_pix)
        for _ in range(n_frames):
            np.random.shuffle(scratch)

    # ------------------------------------------------------------------ frame sources (overridable for in-memory data)
    def load_frame(self, frame_i: int):
        rgb = load_rgb_png(self.rgb_paths[frame_i])
        depth = load_depth_npy(self.depth_paths[frame_i])
        return rgb, depth

    def _frame_stream(self, lo: int, hi: int, depth_sample_rate: int, skip_shuffles: int = 0, stage: bool = False):
        """(frame_i, rgb, depth, samples, staged) in frame order; before the first frame is sampled the RNG is advanced past
        `skip_shuffles` frames (skip_pixel_shuffles).  With prefetch_frames > 0 three kinds of host threads keep the fusing
        thread fed (a frame was 22 us of kernels inside 1.5-3 ms of host work on one thread):
          * a pool decodes the PNG / npy files of the next frames;
          * ONE sampler thread draws the pixel lists strictly in frame order, so the global NumPy RNG is consumed exactly as
            in the reference loop (vlmap_builder.py:275-277) - provided nothing else draws from np.random while the map is
            being built (upstream's loop does not).  It does nothing else: the reference's shuffle is the one inherently serial
            stage (2 ms per 720x1080 frame), everything around it overlaps with it;
          * (stage=True) ONE stager thread copies depth / rgb / samples into page-locked slots and starts their asynchronous
            transfer on a copy stream (device.FrameStager); `staged` then carries device views + the event the fusing stream
            waits for.  staged is None otherwise (the arrays are copied synchronously by the fusing thread, as before).
        Pillow, np.load, the shuffle (host C) and large NumPy copies release the GIL."""
        n = int(self.prefetch_frames or 0)
        self._frames_ready = lambda: 0                                    # how many further frames could be taken without waiting
        self._resolve_pixel_sampling()
        if self.pixel_sampling not in ("reference", "uniform"):
            raise ValueError(f"pixel_sampling must be 'reference' or 'uniform', not {self.pixel_sampling!r}")
        if self.pixel_sampling == "uniform":
            self._uniform_seed = int(np.random.randint(0, 2**31 - 1))     # the one draw from the global RNG (np.random.seed applies)
            skip_shuffles = 0                                             # per-frame generators: nothing to fast-forward
        if n <= 0 or hi - lo <= 1:
            for i in range(lo, hi):
                rgb, depth = self.load_frame(i)
                if i == lo and skip_shuffles:
                    self._announce_skip(skip_shuffles, depth.shape[0] * depth.shape[1])
                    self.skip_pixel_shuffles(skip_shuffles, depth.shape[0] * depth.shape[1])
                yield i, rgb, depth, self._draw_samples(i, depth.shape[0] * depth.shape[1], depth_sample_rate), None
            return
        import queue
        import threading
        from concurrent.futures import ThreadPoolExecutor
        import time
        st_ = self.pipeline_stats = dict(sampler_busy_s=0.0, stager_busy_s=0.0, fuse_thread_wait_s=0.0, frames=0)
        sampled = queue.Queue(maxsize=n)                  # sampler -> stager (or straight to the consumer)
        out = queue.Queue(maxsize=n) if stage else sampled
        self._frames_ready = out.qsize
        stop = threading.Event()
        stager = None
        if stage:
            from .. import _lib
            from ..device import FrameStager
            dev = _lib.current_device()
            # slots: frames queued for the consumer + the one being staged + the one being fused + a batch held back for one launch
            stager = self._stager = FrameStager(n + max(1, int(self.batch_frames or 1), int(self.frame_loop_frames or 1)) + 3, device=dev)

        def put(q, item) -> bool:
            """blocking put that gives up once the consumer has gone (never blocks forever on a full queue)"""
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.1)
                    return True
                except queue.Full:
                    continue
            return False

        def sampler(ex):
            try:
                futs = {}
                nxt = lo
                for i in range(lo, hi):
                    while nxt < hi and nxt <= i + n:
                        futs[nxt] = ex.submit(self.load_frame, nxt)
                        nxt += 1
                    rgb, depth = futs.pop(i).result()
                    if i == lo and skip_shuffles:
                        self._announce_skip(skip_shuffles, depth.shape[0] * depth.shape[1])
                        self.skip_pixel_shuffles(skip_shuffles, depth.shape[0] * depth.shape[1])
                    t_ = time.perf_counter()
                    n_pix = depth.shape[0] * depth.shape[1]
                    smp = None
                    if workers is not None and 0 < n_pix < (1 << 31):
                        st = np.random.get_state()
                        if st[0] == "MT19937":
                            # the frame's pixel list is computed from a snapshot by a worker; this thread only moves the RNG on
                            key = np.ascontiguousarray(st[1], dtype=np.uint32).copy()
                            smp = workers.submit(timed_sample, key, int(st[2]), n_pix, depth_sample_rate)
                            self.skip_pixel_shuffles(1, n_pix)
                    if smp is None:
                        smp = self._draw_samples(i, n_pix, depth_sample_rate)
                    st_["sampler_busy_s"] += time.perf_counter() - t_
                    if not put(sampled, (i, rgb, depth, smp, None)):
                        return
                put(sampled, None)
            except BaseException as e:     # surfaced on the consuming thread
                put(sampled, e)

        def timed_sample(key, pos, n_pix, rate):
            t_ = time.perf_counter()
            smp = self._sample_from_state(key, pos, n_pix, rate)
            with st_lock:
                st_["sampler_workers_busy_s"] += time.perf_counter() - t_
            return smp

        def resolved(item):
            """the sample list of a queued frame: wait for the worker that computes it (frames stay in order)"""
            if item is None or isinstance(item, BaseException) or isinstance(item[3], np.ndarray):
                return item
            try:
                return item[:3] + (item[3].result(),) + item[4:]
            except BaseException as e:
                return e

        def stage_frames():
            try:
                from .. import _lib
                _lib.set_device(dev)       # HIP keeps the current device per thread
                while not stop.is_set():
                    try:
                        item = sampled.get(timeout=0.1)
                    except queue.Empty:
                        continue
                    item = resolved(item)
                    if item is None or isinstance(item, BaseException):
                        put(out, item)
                        return
                    i, rgb, depth, samples, _ = item
                    t_ = time.perf_counter()
                    sf = stager.stage(depth, rgb, samples)
                    st_["stager_busy_s"] += time.perf_counter() - t_
                    if not put(out, (i, rgb, depth, samples, sf)):
                        return
            except BaseException as e:
                put(out, e)

        nw = self._n_sampler_workers() if self.pixel_sampling == "reference" else 0
        st_["sampler_workers"] = nw
        st_["sampler_workers_busy_s"] = 0.0
        st_lock = threading.Lock()
        workers = ThreadPoolExecutor(max_workers=nw, thread_name_prefix="avl-sample") if nw > 0 else None
        with ThreadPoolExecutor(max_workers=min(n, 8), thread_name_prefix="avl-frame") as ex:
            threads = [threading.Thread(target=sampler, args=(ex,), name="avl-sampler", daemon=True)]
            if stage:
                threads.append(threading.Thread(target=stage_frames, name="avl-stager", daemon=True))
            for th in threads:
                th.start()
            try:
                while True:
                    t_ = time.perf_counter()
                    item = out.get()
                    if not stage:
                        item = resolved(item)
                    st_["fuse_thread_wait_s"] += time.perf_counter() - t_
                    if item is None:
                        break
                    if isinstance(item, BaseException):
                        raise item
                    st_["frames"] += 1
                    yield item
            finally:
                stop.set()
                for th in threads:
                    th.join()
                if workers is not None:
                    workers.shutdown(wait=True, cancel_futures=True)

    def _init_lseg(self):
        """Reference: vlmap_builder.py:226-264 builds LSegEncNet from demo_e200.ckpt.  The model is not part of this
        package: pass feat_extractor=..., or have the upstream `avlmaps` package (and its checkpoint) importable."""
        if self.feat_extractor is not None:
            return self.feat_extractor
        try:
            from ..lseg_adapter import load_upstream_lseg
        except Exception as e:  # pragma: no cover
            raise RuntimeError("no feat_extractor given and the upstream LSeg model is not importable") from e
        self.feat_extractor = load_upstream_lseg()
        return self.feat_extractor

    def _features_hwc(self, rgb):
        f = self.feat_extractor(rgb)
        if isinstance(f, np.ndarray):
            if f.ndim == 4:                          # reference layout (1, D, Hf, Wf)
                f = np.transpose(f[0], (1, 2, 0))
            return np.ascontiguousarray(f, dtype=np.float32)
        if f.dim() == 4:
            f = f[0].permute(1, 2, 0)
        return f.float().contiguous()

    # ------------------------------------------------------------------ the build
    def create_mobile_base_map(self):
        """Build the 3-D map centred at the first base frame.  Reference: vlmap_builder.py:54-185."""
        pose_info = cfg_get(self.map_config, "pose_info")
        camera_height = cfg_get(pose_info, "camera_height")
        cs = cfg_get(self.map_config, "cell_size")
        gs = cfg_get(self.map_config, "grid_size")
        depth_sample_rate = cfg_get(self.map_config, "depth_sample_rate")
        calib_mat = np.array(list(cfg_get(self.map_config, "cam_calib_mat")), dtype=np.float64).reshape((3, 3))
        calib_inv = np.linalg.inv(calib_mat)        # mapping_utils.py:237

        self.base_poses = np.loadtxt(self.pose_path).reshape((-1, 7))
        transforms = self.frame_transforms(self.base_poses)

        self.map_save_dir = self.data_dir / "vlmap"
        os.makedirs(self.map_save_dir, exist_ok=True)
        self.map_save_path = self.map_save_dir / "vlmaps.h5df"

        self._init_lseg()
        rank, ws = _dist_rank_ws()
        self._resolve_pixel_sampling()
        n_frames = min(len(self.rgb_paths), len(self.depth_paths), len(self.base_poses))
        lo, hi = parallel.shard_frames(n_frames, rank, ws)

        vh = int(camera_height / cs)                 # vlmap_builder.py:201
        if self.shard_sampling not in ("replay", "independent"):
            raise ValueError(f"shard_sampling must be 'replay' or 'independent', not {self.shard_sampling!r}")
        skip = lo if (ws > 1 and self.shard_sampling == "replay") else 0
        # checkpoint rounds of a multi-rank build: as many as the longest shard has full save_every blocks (known to every rank)
        rounds_total = ((n_frames + ws - 1) // ws) // self.save_every if (ws > 1 and self.save_every) else 0
        try:
            self._build_loop(lo, hi, depth_sample_rate, skip, rank, ws, gs, cs, vh, calib_mat, calib_inv, transforms, rounds_total)
        except _RanksAborted:
            raise
        except BaseException as e:
            # several ranks: the others are (or will be) waiting in a collective -- make every rank fail now instead of hanging
            # until the RCCL timeout.  WHICH collective decides how: between rounds the peers reach the
            # status all-reduce that opens the next round, and this rank joins it with its failure flag set; a rank that fails
            # INSIDE a round (the merge's all_to_alls / all_gathers, rank 0's file write before the round's last collective) has
            # peers sitting in a collective of another kind -- a mismatched all_reduce would hang or exchange garbage on RCCL --
            # so the job is torn down instead.
            if ws > 1:
                if getattr(self, "_in_round", False):
                    self._abort_ranks(rank, e)
                else:
                    self._announce_failure(rank, e)
            raise

    def _build_loop(self, lo, hi, depth_sample_rate, skip, rank, ws, gs, cs, vh, calib_mat, calib_inv, transforms, rounds_total):
        acc = None
        mapped_iter_set = set()
        pending = []
        rounds_done = 0
        probation = None
        pending_storage = None
        self.deferred_fuse_active = self.deferred_fuse is True and self.batch_frames <= 1
        import time
        stage = bool(self.stage_frames and (self.prefetch_frames or 0) > 0)
        self._stager = None
        self.pipeline_stats = {}
        self.build_times = dict(checkpoints_on_fusing_thread_s=0.0, checkpoints=0, frames_in_c_loop_calls=0, c_loop_calls=0,
                                pixel_sampling=self.pixel_sampling)
        t_loop = time.perf_counter()
        import collections
        group = []                  # frames waiting for one avl_builder_integrate_frames call
        group_shape = None
        recent = collections.deque(maxlen=2 * max(1, int(self.frame_loop_frames or 1)) + 2)     # storage ranges of the last frames' features
        ring_k = 1 << 30            # smallest distance at which the extractor was seen to recycle feature storage

        def issue_group():
            if not group:
                return
            kw = dict(calib_inv=calib_inv, min_depth=self.min_depth, max_depth=self.max_depth, sigma_sq=self.sigma_sq)
            if len(group) == 1:
                fi, d_, s_, f_, r_, _st = group[0]
                acc.integrate_frame(d_, calib_mat, transforms[fi], s_, f_, r_, frame_idx=fi, **kw)
            else:
                plan = acc.make_batch_plan([g[1] for g in group], [g[2] for g in group], [g[3] for g in group], [g[4] for g in group])
                acc.integrate_frames(plan, calib_mat, np.stack([transforms[g[0]] for g in group]), frame_idx0=group[0][0], **kw)
                self.build_times["frames_in_c_loop_calls"] += len(group)
                self.build_times["c_loop_calls"] += 1
            for g in group:
                if g[5] is not None:
                    self._stager.release(g[5])        # depth / rgb / samples are read by these launches only (features: deferred)
            group.clear()
        # steps of the fusing thread that took longer than 50 ms (a cold process has a few: the first frames of the stream, the
        # accumulator's allocation, a checkpoint): build_times["slow_steps"] = [(seconds, frame, what)], what a stalled build is asked first
        slow = self.build_times["slow_steps"] = []
        t_lap = [time.perf_counter()]

        def lap(what, frame):
            t = time.perf_counter()
            if t - t_lap[0] > 0.05 and len(slow) < 32:
                slow.append((round(t - t_lap[0], 4), int(frame), what))
            t_lap[0] = t
        for frame_i, rgb, depth, samples, staged in self._frame_stream(lo, hi, depth_sample_rate, skip_shuffles=skip, stage=stage):
            lap("waiting for the frame stream", frame_i)
            if self.skip_mapped_frames and acc is not None and frame_i in mapped_iter_set and frame_i in self._resumed_frames:
                continue        # the pixel shuffle of the skipped frame was still drawn, so later frames sample as upstream
            t_ext = time.perf_counter()
            feat = self._features_hwc(rgb)
            t_ext = time.perf_counter() - t_ext
            lap("feature extractor", frame_i)
            if acc is None:
                D = int(feat.shape[2])
                self.clip_feat_dim = D
                # the reference starts at gs*gs rows and doubles (_reserve_map_space, vlmap_builder.py:286-311); so does
                # the accumulator (max_capacity: every cell of the grid)
                if ws > 1:
                    D = self._agree_on_width(D)     # collective; ranks without frames join it below
                acc = ops.VoxelAccumulator(gs, cs, vh, D, capacity=self.capacity or max(gs * gs, 1 << 16), max_capacity=self.max_capacity,
                                           deferred_fuse=self.deferred_fuse is True and self.batch_frames <= 1)
                probation = [] if (self.deferred_fuse == "auto" and self.batch_frames <= 1) else None
                mapped_iter_set = self._resume(acc, ws, rank)
                self._resumed_frames = frozenset(mapped_iter_set)
                if self.skip_mapped_frames and frame_i in self._resumed_frames:
                    continue
                if not mapped_iter_set and self.exact_rgb:
                    # per-sample log -> finalize replays the reference's sequential weight / uint8 colour exactly (several
                    # ranks: the replay state is chained through the ranks in frame order, parallel.merge_accumulator)
                    npix = depth.shape[0] * depth.shape[1]
                    acc.enable_replay_log((hi - lo) * ((npix + depth_sample_rate - 1) // depth_sample_rate))
                lap("accumulator set-up (allocation, resume, replay log)", frame_i)
            if staged is not None:
                # the frame's arrays are already on their way to the device (page-locked slot, copy stream): the fusing stream
                # waits for them on the device, the host does not
                self._stager.acquire(staged)
                depth, samples, rgb = staged.depth, staged.samples, staged.rgb
            if self.batch_frames > 1:
                pending.append((frame_i, depth, samples, feat, rgb, staged))
                if len(pending) >= self.batch_frames:
                    self._flush(acc, pending, calib_mat, calib_inv, transforms)
            else:
                rng_ = _storage_range(feat)
                # how long ago did the extractor last hand out memory overlapping this frame's features?  (compared by address RANGE:
                # per-frame views of one batched output are different memory, a recycled ring buffer is the same)
                dist_ = next((k + 1 for k, (r, _ref) in enumerate(recent) if _ranges_overlap(r, rng_)), None)
                waiting = len(group) + (1 if (self.deferred_fuse_active and pending_storage is not None) else 0)
                if dist_ is not None and dist_ <= waiting:
                    # the extractor wrote this frame into storage in which an EARLIER frame's features still wait to be fused
                    # (probation saw fresh storage; the extractor started recycling later).  Those features are gone -- fail loudly
                    # instead of fusing the wrong bytes
                    raise RuntimeError(f"frame {frame_i}: the feature extractor reused the storage of the previous frame's features "
                                       "before they were fused (deferred fuse / frame_loop_frames); build with deferred_fuse=False and "
                                       "frame_loop_frames=1 for this extractor")
                if dist_ is not None and recent[dist_ - 1][1] is not None:
                    ring_k = min(ring_k, dist_)          # that tensor is still referenced here: the allocator did not hand its memory out
                                                         # again, the extractor itself recycles its output buffers at this distance
                # frames are held back only while the extractor is not what the loop waits for (a model that takes milliseconds per
                # frame gains nothing from sharing a call); the last frames' tensors are then kept referenced, so that an address
                # seen again within the window is the extractor recycling a buffer, not the allocator reusing freed memory
                fast = t_ext < self.frame_loop_max_extract_s and int(self.frame_loop_frames or 1) > 1
                recent.appendleft((rng_, feat if fast else None))
                # frames held back for ONE avl_builder_integrate_frames call: only features that are device tensors in storage the
                # extractor has not been seen to recycle within twice that distance, consecutive frame indices, equal shapes
                limit = 1
                if rng_ is not None and probation is None and fast:
                    limit = max(1, min(int(self.frame_loop_frames), min(len(recent), ring_k) // 2))
                shp = (tuple(depth.shape), tuple(feat.shape), tuple(np.shape(samples) if isinstance(samples, np.ndarray) else samples.shape))
                if group and (frame_i != group[-1][0] + 1 or shp != group_shape):
                    issue_group()
                group_shape = shp
                group.append((frame_i, depth, samples, feat, rgb, staged))
                if len(group) >= limit or (self._frames_ready() <= 0 and not self.frame_loop_eager):
                    issue_group()
                if probation is not None:
                    # deferred fuse on probation: does the extractor hand out fresh storage while the previous tensor is alive?
                    probation.append(feat)
                    if len(probation) == 2:
                        rr = [_storage_range(f) for f in probation]
                        fresh = rr[0] is None or not _ranges_overlap(rr[0], rr[1])     # NumPy features are staged by us: always fresh
                        if fresh:
                            acc.set_deferred_fuse(True)
                        self.deferred_fuse_active = bool(fresh)
                        probation = None
                pending_storage = rng_
            mapped_iter_set.add(frame_i)
            lap("fusing (launches, group bookkeeping)", frame_i)
            if ws == 1 and self.save_every and frame_i % self.save_every == self.save_every - 1:
                issue_group()
                self._flush(acc, pending, calib_mat, calib_inv, transforms)
                t_ck = time.perf_counter()
                if not (self.skip_busy_checkpoints and getattr(self, "_save_thread", None) is not None and self._save_thread.is_alive()):
                    print(f"Temporarily saving {acc.num_voxels()} features at iter {frame_i}...")
                self._checkpoint(acc, mapped_iter_set)
                self.build_times["checkpoints_on_fusing_thread_s"] += time.perf_counter() - t_ck
                self.build_times["checkpoints"] += 1
                lap("checkpoint", frame_i)
            elif ws > 1 and self.save_every and (frame_i - lo) % self.save_every == self.save_every - 1 and rounds_done < rounds_total:
                # upstream saves every 100 frames (vlmap_builder.py:181-183); with several ranks a checkpoint is a merge, i.e. a
                # collective: every rank joins round j after its (j + 1) * save_every-th frame (or at the end of its shard)
                issue_group()
                self._flush(acc, pending, calib_mat, calib_inv, transforms)
                self._checkpoint_ranks(acc, mapped_iter_set, rank, ws, final=False)
                rounds_done += 1
                lap("checkpoint round (merge)", frame_i)
        issue_group()
        lap("last launches", hi)
        if acc is None:
            if ws == 1:
                raise RuntimeError("no frames to map")
            # an empty shard (fewer frames than ranks would fill) still takes part in the collectives
            D = self._agree_on_width(0)
            if D <= 0:
                raise RuntimeError("no frames to map")
            acc = ops.VoxelAccumulator(gs, cs, vh, D, capacity=1 << 10, max_capacity=0)
            mapped_iter_set = self._resume(acc, ws, rank)
            if not mapped_iter_set and self.exact_rgb:
                acc.enable_replay_log(1)
        self._flush(acc, pending, calib_mat, calib_inv, transforms)
        if self._stager is not None:
            self._stager.close()                       # waits for the launches that still read its slots
            self._stager = None
        lap("draining the device (stager close)", hi)
        while ws > 1 and rounds_done < rounds_total:      # a short (or empty) shard: the checkpoint rounds the others still run
            self._checkpoint_ranks(acc, mapped_iter_set, rank, ws, final=False)
            rounds_done += 1
        acc.flush()
        acc.num_voxels()                                  # (synchronises: the frame loop's device work ends here)
        t_fin = time.perf_counter()
        self.build_times.update(frame_loop_s=t_fin - t_loop, checkpoints_skipped=getattr(self, "checkpoints_skipped", 0), **self.pipeline_stats)
        self._finish(acc, mapped_iter_set, rank, ws, gs, vh)
        self.build_times["final_save_s"] = time.perf_counter() - t_fin
        acc.release_scratch()       # sorted replay log + pool memory of the final merge / finalisation go back to the driver

    def _flush(self, acc, pending, calib_mat, calib_inv, transforms):
        """fuse the buffered frames (consecutive indices, equal shapes) with one launch pair"""
        if not pending:
            return
        i0 = pending[0][0]
        same = all(p[1].shape == pending[0][1].shape and tuple(p[3].shape) == tuple(pending[0][3].shape) for p in pending)
        if same and len(pending) > 1 and [p[0] for p in pending] == list(range(i0, i0 + len(pending))):
            acc.integrate_batch([p[1] for p in pending], calib_mat, [transforms[p[0]] for p in pending], [p[2] for p in pending],
                                [p[3] for p in pending], [p[4] for p in pending], frame_idx0=i0, calib_inv=calib_inv,
                                min_depth=self.min_depth, max_depth=self.max_depth, sigma_sq=self.sigma_sq)
        else:
            for fi, depth, samples, feat, rgb, _st in pending:
                acc.integrate_frame(depth, calib_mat, transforms[fi], samples, feat, rgb, frame_idx=fi, calib_inv=calib_inv,
                                    min_depth=self.min_depth, max_depth=self.max_depth, sigma_sq=self.sigma_sq)
        for p in pending:
            if p[5] is not None:
                self._stager.release(p[5])
        pending.clear()

    def create_camera_map(self):
        """Upstream returns (does not raise) NotImplementedError.  Reference: vlmap_builder.py:187-193."""
        return NotImplementedError

    # ------------------------------------------------------------------ helpers
    @staticmethod
    def _agree_on_width(D: int) -> int:
        """feature width over the ranks (a rank without frames passes 0): one tiny MAX all-reduce at the first frame -- of the same
        [value, failure flag] shape as the status all-reduce of the checkpoint rounds, so that a rank that failed before its first
        frame (_announce_failure) is understood here too"""
        import torch
        import torch.distributed as dist
        d = torch.tensor([int(D), 0], dtype=torch.int64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(d, op=dist.ReduceOp.MAX)
        if int(d[1].item()):
            raise _RanksAborted("multi-rank build aborted: another rank reported a failure (see its traceback)")
        return int(d[0].item())

    def _resume(self, acc, ws, rank=0):
        """Continue from an existing map file.  Reference: vlmap_builder.py:212-222 (note: upstream restores
        mapped_iter_set but never skips frames, so a resumed run re-fuses every frame; kept as is unless
        skip_mapped_frames).  Several ranks: rank 0 imports the map, the others only mark their accumulators as continuing
        one (same first-touch key space), every rank learns mapped_iter_list; the merge then keeps the file's voxel ids.
        A file whose last in-place checkpoint was interrupted (MapFileWriter.MARKER) keeps its rows but its frame list is
        not trusted: every frame is fused again, as upstream does anyway."""
        if ws > 1:
            # rank 0's view of the file decides for everybody (a node-local disk may show the file to some ranks only: a rank that
            # neither imports the map nor marks its accumulator as continuing one would hand out first-touch keys that sort BEFORE
            # the imported voxels and the file's voxel ids would be lost)
            import torch.distributed as dist
            info = [None]
            if rank == 0:
                try:
                    exists = bool(map_file_exists(self.map_save_path))
                    iters = read_map_dataset(self.map_save_path, "mapped_iter_list") if exists else None
                    info = [(exists, [] if iters is None else np.asarray(iters).tolist(),
                             bool(map_checkpoint_complete(self.map_save_path)) if exists else True)]
                except Exception as e:      # the peers wait in the broadcast below: they must hear about it THERE
                    info = [("error", f"{type(e).__name__}: {e}")]
            dist.broadcast_object_list(info, src=0)
            if info[0][0] == "error":
                raise _RanksAborted(f"multi-rank build aborted: rank 0 could not read {self.map_save_path}: {info[0][1]}")
            exists, iters, complete = info[0]
            if not exists:
                return set()
            if rank == 0:
                print(f"[avlmaps_amd] {self.map_save_path} exists: continuing that map ({len(iters)} frames fused so far) on {ws} ranks, "
                      "like upstream's single-process builder does (vlmap_builder.py:212-222); delete the file to start over, set "
                      "skip_mapped_frames to skip the frames it lists", flush=True)
                _, grid_feat, grid_pos, weight, _occ, grid_rgb = load_3d_map(self.map_save_path)[:6]
                acc.import_map(grid_feat, grid_pos, weight, grid_rgb)
            else:
                acc.mark_resumed()
            mapped = set(iters)
        else:
            if not map_file_exists(self.map_save_path):
                return set()
            mapped_iter_list, grid_feat, grid_pos, weight, _occ, grid_rgb = load_3d_map(self.map_save_path)[:6]
            acc.import_map(grid_feat, grid_pos, weight, grid_rgb)
            mapped = set(mapped_iter_list)
            complete = map_checkpoint_complete(self.map_save_path)
        if not complete:
            print(f"[avlmaps_amd] {self.map_save_path}: the last checkpoint was interrupted; keeping its voxels, re-fusing every frame")
            self.skip_mapped_frames = False
        return mapped

    def _announce_failure(self, rank, exc) -> None:
        """a rank that failed locally joins the collective the others will reach next -- the status all-reduce that opens every
        checkpoint round -- with its error flag set; they raise _RanksAborted there"""
        try:
            import torch
            import torch.distributed as dist
            print(f"[avlmaps_amd] rank {rank}: {type(exc).__name__}: {exc} -- telling the other ranks", flush=True)
            flags = torch.tensor([0, 1], dtype=torch.int64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(flags, op=dist.ReduceOp.MAX)
        except BaseException:      # the process group itself is gone: nothing more to do
            pass

    def _abort_ranks(self, rank, exc) -> None:
        """a rank that failed INSIDE a collective round cannot be matched by its peers' next collective: abort the process group
        (peers blocked in RCCL return with an error) and, unless AVLMAPS_ABORT_EXITS=0, leave the process with a non-zero status
        so that torchrun tears the job down at once.  The exception is printed first; with AVLMAPS_ABORT_EXITS=0 it propagates."""
        import sys
        import traceback
        print(f"[avlmaps_amd] rank {rank}: {type(exc).__name__}: {exc} inside a collective round -- aborting the job", flush=True)
        traceback.print_exception(type(exc), exc, exc.__traceback__)
        sys.stdout.flush()
        sys.stderr.flush()
        try:
            import torch.distributed as dist
            abort = getattr(dist.distributed_c10d, "_abort_process_group", None)
            if abort is not None:
                abort()
        except BaseException:
            pass
        if os.environ.get("AVLMAPS_ABORT_EXITS", "1") != "0":
            os._exit(70)

    def _finish(self, acc, mapped_iter_set, rank, ws, gs, vh):
        if ws == 1:
            self._checkpoint(acc, mapped_iter_set, background=False)
            return
        import torch
        import torch.distributed as dist
        self._checkpoint_ranks(acc, mapped_iter_set, rank, ws, final=True)
        self._join_save()
        # closing status all-reduce (same [busy, failed] shape as the one that opens a round) instead of a bare barrier: if rank 0's
        # final file write raised, its _announce_failure is THIS collective for the peers, and they raise too
        flags = torch.zeros(2, dtype=torch.int64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(flags, op=dist.ReduceOp.MAX)
        if int(flags[1].item()):
            raise _RanksAborted("multi-rank build aborted: another rank reported a failure (see its traceback)")

    def _checkpoint_ranks(self, acc, mapped_iter_set, rank, ws, final: bool) -> None:
        """One merge of the ranks' accumulators (a collective) + the map file written by rank 0.  Non-destructive: frames keep
        streaming into the same accumulators afterwards.  The file is written by a host thread (device-to-host copy included)
        unless `final`."""
        import torch
        import torch.distributed as dist
        # ONE tiny MAX all-reduce opens every round: [rank 0's writer still busy, some rank failed].  A periodic checkpoint is
        # only worth a merge if rank 0 can take it: while its writer thread is still busy with the previous one (a 2 M-voxel map
        # is ~5 GB of file) the round is skipped by everybody -- the build is never throttled to the disk's speed.  A failure
        # anywhere (a frame that could not be read, the writer thread's exception on rank 0) makes EVERY rank raise here.
        prev = getattr(self, "_save_thread", None)
        err = 1 if (rank == 0 and getattr(self, "_save_error", None) is not None and (prev is None or not prev.is_alive())) else 0
        flags = torch.tensor([1 if (not final and rank == 0 and prev is not None and prev.is_alive()) else 0, err], dtype=torch.int64,
                             device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(flags, op=dist.ReduceOp.MAX)
        if int(flags[1].item()):
            if err:
                self._join_save()                         # raises the writer thread's exception on rank 0
            raise _RanksAborted("multi-rank build aborted: another rank reported a failure (see its traceback)")
        if int(flags[0].item()):
            self.checkpoints_skipped = getattr(self, "checkpoints_skipped", 0) + 1
            return
        # from here to the round's last collective a local failure cannot be announced by an all-reduce (see create_mobile_base_map)
        self._in_round = True
        self.merge_timings = {}
        if self.merge_mode == "reduce":
            fin = parallel.merge_accumulator(acc, dst=0, exact_rgb=self.exact_rgb, timings=self.merge_timings)
            self.map_shard = None
        elif self.merge_mode == "sharded":
            shard = parallel.merge_accumulator_sharded(acc, exact_rgb=self.exact_rgb, timings=self.merge_timings, gather_to=0)
            fin = shard.pop("full", None)
            self.map_shard = shard if final else None
        else:
            raise ValueError(f"merge_mode must be 'sharded' or 'reduce', not {self.merge_mode!r}")
        sets = [None] * ws
        dist.all_gather_object(sets, sorted(mapped_iter_set))
        self._in_round = False      # the round's collectives are done: what fails now (rank 0's file write) is announced at the next status all-reduce
        if rank != 0:
            return
        iters = set(i for s in sets for i in s)
        if not final:
            print(f"Temporarily saving {int(fin['grid_pos'].shape[0])} features of {len(iters)} frames ({ws} ranks)...")

        def to_host():
            return {k: v.cpu().numpy() for k, v in fin.items()}
        self._save_3d_map(to_host, iters, background=not final)

    def _checkpoint(self, acc, mapped_iter_set, background: bool = True) -> None:
        """Save while frames keep coming (and the final save of a single-process run).  After the first full write only the rows
        that changed and the new rows cross PCIe (VoxelAccumulator.finalize_rows -> MapFileWriter.save_packed, which patches its
        host mirror of the map and the file): at 2 M voxels a full copy costs 0.3-0.5 s on this thread, as much as the feature
        extractor needs for the 100 frames between two checkpoints."""
        from ..utils import h5lite
        prev = getattr(self, "_save_thread", None)
        if background and self.skip_busy_checkpoints and prev is not None and prev.is_alive():
            self.checkpoints_skipped = getattr(self, "checkpoints_skipped", 0) + 1
            return
        import time
        t_join = time.perf_counter()
        self._join_save()
        t_join = time.perf_counter() - t_join
        writer = getattr(self, "_map_writer", None)
        lean_ok = (self.incremental_checkpoints and h5lite.available() and writer is not None and writer.n_saved is not None
                   and writer.mirror is not None and writer.path == Path(self.map_save_path) and writer.path.exists())
        import time
        t0 = time.perf_counter()
        log = self.build_times.setdefault("checkpoint_log", []) if hasattr(self, "build_times") else []
        if not lean_ok:
            if background and self.prefetch_frames:
                # the first (full) checkpoint of a build: finalise on the device (ms) and let the WRITER thread bring the map to the
                # host -- a fresh multi-hundred-MB host array is bound by its first-touch page faults (0.1-0.3 s per GB), and on a
                # cold process the HDF5 library, the writer and the staging buffers are created here too: none of it on the
                # fusing thread (0.95 s of a 1.35 s build on the driver's box)
                from .. import _lib
                dev_arrays = acc.finalize(as_numpy=False, want_dirty=self.incremental_checkpoints)
                dev = _lib.current_device()

                def to_host():
                    if dev is not None:
                        _lib.set_device(dev)
                    return {k: (v.numpy() if v is not None else None) for k, v in dev_arrays.items()}
                self._save_3d_map(to_host, mapped_iter_set, background=True)
                log.append(("full, host copy on the writer thread", time.perf_counter() - t0))
                return
            self._save_3d_map(acc.finalize(want_dirty=self.incremental_checkpoints), mapped_iter_set, background=background)
            log.append(("full", time.perf_counter() - t0))
            return
        lean = acc.finalize_rows(writer.n_saved)
        log.append(("changed rows", time.perf_counter() - t0))
        iters = list(mapped_iter_set)

        def write():
            try:
                writer.save_packed(lean, iters)
            except BaseException as e:
                self._save_error = e
        if background and self.prefetch_frames:
            import threading
            self._save_error = None
            self._save_thread = threading.Thread(target=write, name="avl-save", daemon=False)
            self._save_thread.start()
        else:
            t_w = time.perf_counter()
            write()
            self._join_save()
            self.last_map = writer.current_map()
            if hasattr(self, "build_times"):      # the final save of a build, in parts: waiting for the previous checkpoint's writer, rows to the host, file
                st = writer.stats[-1] if writer.stats else {}
                self.build_times["final_save_parts"] = dict(wait_for_writer_s=t_join, changed_rows_to_host_s=log[-1][1], file_s=time.perf_counter() - t_w,
                                                            rows_written=st.get("rows_written"), rows_total=st.get("rows_total"),
                                                            mirror_s=st.get("mirror_seconds"))

    def _join_save(self) -> None:
        prev = getattr(self, "_save_thread", None)
        if prev is not None:
            prev.join()
            self._save_thread = None
        if getattr(self, "_save_error", None) is not None:
            err, self._save_error = self._save_error, None
            raise err

    def _save_3d_map(self, arrays, mapped_iter_set, background: bool = False) -> None:
        """Reference: vlmap_builder.py:313-327 -> mapping_utils.save_3d_map.  The periodic checkpoints (every
        `save_every` frames upstream rewrites the whole file) are written by a host thread while fusion continues; the
        final save, and any save that follows an unfinished one, waits."""
        self._join_save()
        iters = list(mapped_iter_set)
        thunk = arrays if callable(arrays) else None      # multi-rank checkpoints: the device-to-host copy runs on the writer thread
        if thunk is None:
            self.last_map = arrays

        writer = getattr(self, "_map_writer", None)
        if writer is None or writer.path != Path(self.map_save_path):
            writer = self._map_writer = MapFileWriter(self.map_save_path)

        def write():
            try:
                a = arrays
                if thunk is not None:
                    a = thunk()
                    self.last_map = a
                if self.incremental_checkpoints:
                    writer.save(a, iters, a.get("row_dirty"))
                else:
                    save_3d_map(self.map_save_path, a["grid_feat"], a["grid_pos"], a["weight"], a["occupied_ids"], iters, a["grid_rgb"])
            except BaseException as e:
                self._save_error = e
                if not background:
                    raise
        if background and self.prefetch_frames:
            import threading
            self._save_error = None
            self._save_thread = threading.Thread(target=write, name="avl-save", daemon=False)
            self._save_thread.start()
        else:
            write()


import threading as _threading

_SCRATCH = _threading.local()


def _storage_range(feat):
    """(device, first byte, one past the last byte) of the memory a feature tensor occupies, None for anything the builder stages
    itself (NumPy arrays are copied into fresh device buffers)"""
    if not hasattr(feat, "untyped_storage"):
        return None
    try:
        a = int(feat.data_ptr())
        return (str(feat.device), a, a + int(feat.numel()) * int(feat.element_size()))
    except Exception:
        return None


def _ranges_overlap(a, b) -> bool:
    return a is not None and b is not None and a[0] == b[0] and a[1] < b[2] and b[1] < a[2]


def _storage_key(feat):
    """identity of the memory a feature tensor lives in: (device, start of its untyped storage) for torch tensors, None for
    anything the builder stages itself (NumPy arrays are copied into fresh device buffers)"""
    st = getattr(feat, "untyped_storage", None)
    if st is None:
        return None
    try:
        return (str(feat.device), int(st().data_ptr()))
    except Exception:
        return (str(getattr(feat, "device", "")), int(feat.data_ptr()))


class _RanksAborted(RuntimeError):
    """raised on every rank of a multi-rank build when one of them reported a failure"""


def _dist_rank_ws():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except Exception:
        pass
    return 0, 1
