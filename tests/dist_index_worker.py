"""One rank of a row-sharded VLMap indexing run (launched by tests/test_api_gpu.py through torch.distributed.run): every rank
uploads and scores only its block of voxel rows, results are all-gathered; rank 0 stores what VLMap returned.

    python -m torch.distributed.run --nproc-per-node 2 tests/dist_index_worker.py <g3_similarity.npz> <out.npz>
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def main():
    golden, out = sys.argv[1], Path(sys.argv[2])
    from test_host_mirror import Cfg
    from avlmaps_amd import parallel
    from avlmaps_amd.map.vlmap import VLMap
    import avlmaps_amd.map.vlmap as vlmap_mod
    rank, ws, local = parallel.init_distributed()
    import torch
    torch.cuda.set_device(local)
    g = np.load(golden, allow_pickle=False)
    feats = {"sofa": g["q1_mean_feats"], "cats": g["q40_other_last_mean_feats"]}

    def fake(clip_model, landmarks, clip_feat_dim, use_multiple_templates=False, add_other=True):
        return (feats["sofa"] if list(landmarks) == ["sofa"] else feats["cats"]).astype(np.float32), list(landmarks)
    vlmap_mod.landmark_text_feats = fake
    cfg = Cfg(map_type="vlmap", grid_size=1000, cell_size=0.05,
              pose_info=Cfg(camera_height=1.5, base2cam_rot=[1, 0, 0, 0, -1, 0, 0, 0, -1], base_forward_axis=[0, 0, -1],
                            base_left_axis=[-1, 0, 0], base_up_axis=[0, 1, 0]))
    vm = VLMap(cfg)
    vm.grid_feat = g["feat"]
    vm.clip_model, vm.clip_feat_dim = None, 512
    mask = vm.index_map("sofa", with_init_cat=False)
    sm = vm.init_categories([f"cat{i}" for i in range(39)] + ["other"])
    rows = vm._rows
    assert (rows[1] - rows[0]) * ws >= len(g["feat"]) and (ws == 1 or rows[1] - rows[0] < len(g["feat"]))     # only a block is resident
    assert vm._dev_feat.shape[0] == rows[1] - rows[0]
    m7 = vm.index_map("cat7", with_init_cat=True)
    if rank == 0:
        np.savez(out, mask=mask, scores=sm, m7=m7, rows=np.array(rows))
    import torch.distributed as dist
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
