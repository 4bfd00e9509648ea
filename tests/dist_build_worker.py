"""One rank of a multi-process VLMapBuilder run (launched by tests/test_api_gpu.py through torch.distributed.run).

Every rank seeds the global NumPy RNG like the reference run that produced the golden map, builds its contiguous frame
shard from the golden frames and joins the merge; rank 0 writes <out>/vlmap/vlmaps.h5df (+ merge timings as JSON).

    python -m torch.distributed.run --nproc-per-node 2 tests/dist_build_worker.py <golden.npz> <out_dir> <n_frames> [sampling [seed]]
"""
import json
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

    frame_of_call = []

    def extractor(rgb):
        return g["feats"][frame_of_call[-1]][None]           # reference layout (1, D, Hf, Wf)

    b = VLMapBuilder(out_dir, cfg, pose_path, [None] * n_frames, [None] * n_frames, m.base2cam_tf, m.base_transform,
                     feat_extractor=extractor)

    def load_frame(i):
        return g["rgbs"][i], g["depths"][i]

    b.load_frame = load_frame
    b.prefetch_frames = 0                                     # inline loading: extractor calls follow load_frame calls in order
    orig = b._features_hwc

    def feats(rgb):
        return orig(rgb)
    # frame index of the extractor call = order of the frames this rank streams
    lo, hi = parallel.shard_frames(n_frames, rank, ws)
    it = iter(range(lo, hi))

    def features_hwc(rgb):
        frame_of_call.append(next(it))
        return feats(rgb)
    b._features_hwc = features_hwc
    b.capacity = 64                                           # forces the accumulators to double a few times
    if sampling == "uniform":
        b.pixel_sampling = "uniform"                          # per-frame generators: the sharding does not matter
    else:
        b.shard_sampling = sampling
    np.random.seed(seed)                                      # the state the reference run started from, on EVERY rank
    b.create_mobile_base_map()
    if rank == 0:
        (out_dir / "merge_timings.json").write_text(json.dumps(getattr(b, "merge_timings", {})))
    import torch.distributed as dist
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
