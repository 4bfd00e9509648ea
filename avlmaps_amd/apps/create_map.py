"""Build a VLMap for one scene directory.  Counterpart of the reference's application/create_map.py:7-17.

    python -m avlmaps_amd.apps.create_map --data-dir <scene> [--config cfg.yaml] [--features lseg|hash] [--seed N]

<scene>/ holds rgb/*.png, depth/*.npy (float32 metres) and poses.txt (x y z qx qy qz qw per line), the layout of the
reference's dataset/README.md:76-93; the map goes to <scene>/vlmap/vlmaps.h5df (a real HDF5 file: through h5py, or through
the HDF5 C library where h5py is missing).
Multi-GPU: launch with torchrun; frames are sharded over ranks and merged with one row-sharded RCCL exchange (every --save-every
frames per rank as a checkpoint, and at the end); an interrupted run is continued with --resume; with --seed the N-rank map
equals the single-process map (every rank replays the RNG draws of the frames before its shard)."""
from __future__ import annotations

import argparse
import time

import numpy as np


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--data-dir", required=True)
    ap.add_argument("--config", default=None, help="YAML with params / map_config overrides")
    ap.add_argument("--features", choices=["lseg", "hash"], default="lseg",
                    help="lseg = upstream LSegEncNet + demo_e200.ckpt on PyTorch-ROCm; hash = model-free stand-in for smoke runs")
    ap.add_argument("--feat-dim", type=int, default=512)
    ap.add_argument("--seed", type=int, default=None, help="seed of the global NumPy RNG that orders the pixel sampling")
    ap.add_argument("--capacity", type=int, default=None, help="voxel capacity (default gs*gs)")
    ap.add_argument("--prefetch", type=int, default=None, help="frames decoded ahead on host threads (default 4, 0 = inline)")
    ap.add_argument("--batch-frames", type=int, default=None, help="frames fused per launch pair (default 1)")
    ap.add_argument("--deferred-fuse", action="store_true",
                    help="frame-by-frame fusion in one launch per frame (the extractor must return a new tensor per frame)")
    ap.add_argument("--pixel-sampling", choices=["reference", "uniform"], default=None,
                    help="reference = np.random.shuffle on the global RNG like upstream (6 ms per 720x1080 frame, serial); uniform = "
                         "the same distribution from per-frame generators (0.25 ms, other pixels than a seeded upstream run)")
    ap.add_argument("--shard-sampling", choices=["replay", "independent"], default=None,
                    help="several ranks: replay = sample the pixels of the single-process run (default); independent = do not "
                         "fast-forward the RNG past the other ranks' frames (unseeded runs)")
    ap.add_argument("--merge-mode", choices=["sharded", "reduce"], default=None,
                    help="several ranks: sharded = all_to_all of every rank's own voxel rows (default); reduce = one dense sum-reduce")
    ap.add_argument("--save-every", type=int, default=None, help="checkpoint every N frames (per rank); default 100 like upstream")
    ap.add_argument("--resume", action="store_true",
                    help="continue from an existing vlmaps.h5df and skip the frames it lists (upstream re-fuses every frame)")
    args = ap.parse_args(argv)

    from avlmaps_amd import parallel
    from avlmaps_amd.apps.common import HashFeatureExtractor, load_config
    from avlmaps_amd.map import AVLMap
    rank, ws, _ = parallel.init_distributed()
    cfg = load_config(args.config)
    if args.seed is not None:
        np.random.seed(args.seed)
    extractor = HashFeatureExtractor(args.feat_dim) if args.features == "hash" else None
    avlmap = AVLMap(cfg, data_dir=args.data_dir)
    if (args.capacity or args.prefetch is not None or args.batch_frames or args.shard_sampling or args.deferred_fuse or args.pixel_sampling
            or args.merge_mode or args.save_every is not None or args.resume):
        import avlmaps_amd.map.vlmap_builder as vb
        orig = vb.VLMapBuilder.__init__

        def patched(self, *a, **k):
            orig(self, *a, **k)
            if args.capacity:
                self.capacity = args.capacity
            if args.prefetch is not None:
                self.prefetch_frames = args.prefetch
            if args.batch_frames:
                self.batch_frames = args.batch_frames
            if args.shard_sampling:
                self.shard_sampling = args.shard_sampling
            if args.deferred_fuse:
                self.deferred_fuse = True
            if args.pixel_sampling:
                self.pixel_sampling = args.pixel_sampling
            if args.merge_mode:
                self.merge_mode = args.merge_mode
            if args.save_every is not None:
                self.save_every = args.save_every
            if args.resume:
                self.skip_mapped_frames = True
        vb.VLMapBuilder.__init__ = patched
    t0 = time.perf_counter()
    avlmap.create_map(args.data_dir, feat_extractor=extractor)
    if rank == 0:
        n = len(avlmap.vlmap.map_builder.last_map["grid_pos"]) if hasattr(avlmap.vlmap.map_builder, "last_map") else -1
        print(f"map with {n} voxels written to {avlmap.vlmap.map_builder.map_save_path} in {time.perf_counter() - t0:.2f} s")


if __name__ == "__main__":
    main()
