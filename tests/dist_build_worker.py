"""One rank of a multi-process VLMapBuilder run (launched by tests/test_api_gpu.py through torch.distributed.run).

Every rank seeds the global NumPy RNG like the reference run that produced the golden map, builds its contiguous frame
shard from the golden frames and joins the merge; rank 0 writes <out>/vlmap/vlmaps.h5df (+ merge timings as JSON).

    python -m torch.distributed.run --nproc-per-node 2 tests/dist_build_worker.py <golden.npz> <out_dir> <n_frames> [sampling [seed]]
"""
import json
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def main():
    golden, out_dir, n_frames = sys.argv[1], Path(sys.argv[2]), int(sys.argv[3])
    sampling = sys.argv[4] if len(sys.argv) > 4 else "replay"
    seed = int(sys.argv[5]) if len(sys.argv) > 5 else 1234
    from test_host_mirror import make_cfg
    from avlmaps_amd import parallel
    from avlmaps_amd.map.map import Map
    from avlmaps_amd.map.vlmap_builder import VLMapBuilder
    rank, ws, local = parallel.init_distributed()
    import torch
    torch.cuda.set_device(local)
    g = np.load(golden, allow_pickle=False)
    cfg = make_cfg(g)
    m = Map(cfg)
    out_dir.mkdir(parents=True, exist_ok=True)
    pose_path = out_dir / f"poses_rank{rank}.txt"
    np.savetxt(pose_path, g["poses"][:n_frames])

    loaded = []                                               # frame indices in the order the builder asked for them

    def extractor(rgb):
        return g["feats"][loaded[-1]][None]                   # reference layout (1, D, Hf, Wf)

    b = VLMapBuilder(out_dir, cfg, pose_path, [None] * n_frames, [None] * n_frames, m.base2cam_tf, m.base_transform,
                     feat_extractor=extractor)
    stop_after = int(os.environ.get("AVL_TEST_STOP_AFTER", "0"))       # simulated interruption after that many local frames

    fail_rank = int(os.environ.get("AVL_TEST_FAIL_RANK", "-1"))             # that rank cannot read its second frame

    def load_frame(i):
        if rank == fail_rank and len(loaded) >= 1:
            raise OSError(f"frame {i}: disk gone (simulated on rank {rank})")
        if stop_after and len(loaded) >= stop_after:
            b._join_save()                                    # the checkpoint that was being written completes, then the run dies
            import torch.distributed as dist
            if dist.is_initialized():
                dist.barrier()
            os._exit(0)
        loaded.append(i)
        return g["rgbs"][i], g["depths"][i]

    b.load_frame = load_frame
    b.prefetch_frames = 0                                     # inline loading: extractor calls follow load_frame calls in order
    b.merge_mode = os.environ.get("AVL_TEST_MERGE_MODE", b.merge_mode)
    b.save_every = int(os.environ.get("AVL_TEST_SAVE_EVERY", b.save_every))
    b.skip_mapped_frames = os.environ.get("AVL_TEST_SKIP_MAPPED", "0") == "1"
    b.capacity = 64                                           # forces the accumulators to double a few times
    if sampling == "uniform":
        b.pixel_sampling = "uniform"                          # per-frame generators: the sharding does not matter
    else:
        b.pixel_sampling = "reference"                        # (the default): the seeded reference run's pixels
        b.shard_sampling = sampling
    np.random.seed(seed)                                      # the state the reference run started from, on EVERY rank
    b.create_mobile_base_map()
    if os.environ.get("AVL_TEST_ADOPT") == "1":
        # straight from the build to the index: every rank adopts its block of the merged map (no upload), the sharded VLMap
        # answers like a single process that loaded the file
        from avlmaps_amd.map.vlmap import VLMap
        import avlmaps_amd.map.vlmap as vlmap_mod
        D = int(b.clip_feat_dim)
        qrng = np.random.default_rng(5)
        table = {"sofa": qrng.standard_normal((2, D)).astype(np.float32)}
        vlmap_mod.landmark_text_feats = lambda m, names, d, use_multiple_templates=False, add_other=True: (table[names[0]], list(names))
        vm = VLMap(cfg)
        vm.map_builder = b
        shard_rows = tuple(b.map_shard["rows"]) if b.map_shard is not None else None
        uploads = []
        vm._start_device_prefetch = lambda: uploads.append(1)
        assert vm.load_map(out_dir)
        vm.clip_model, vm.clip_feat_dim = None, D
        adopted = bool(ws > 1 and not uploads and vm._dev_feat is not None and vm._rows == shard_rows)
        mask = vm.index_map("sofa", with_init_cat=False)
        if rank == 0:
            want = np.argmax(vm.grid_feat.astype(np.float64) @ table["sofa"].astype(np.float64).T, axis=1) == 0
            np.savez(out_dir / "adopt.npz", mask=mask, want=want, adopted=np.array(adopted), rows=np.array(vm._rows))
    tim = dict(getattr(b, "merge_timings", {}))
    if b.map_shard is not None:
        tim["shard_rows"] = list(b.map_shard["rows"])
        tim["shard_feat_shape"] = list(b.map_shard["grid_feat"].shape)
    (out_dir / f"merge_timings_rank{rank}.json").write_text(json.dumps(tim))
    if rank == 0:
        (out_dir / "merge_timings.json").write_text(json.dumps(tim))
    import torch.distributed as dist
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
