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

    def load_frame(i):
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
        b.shard_sampling = sampling
    np.random.seed(seed)                                      # the state the reference run started from, on EVERY rank
    b.create_mobile_base_map()
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
