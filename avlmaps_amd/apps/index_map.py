"""Index a saved VLMap with a text query.  Counterpart of branch "1. object" of the reference's
application/index_map.py:23-38 (the Open3D viewer and the habitat branches are not part of the hot path).

    python -m avlmaps_amd.apps.index_map --data-dir <scene> --query sofa [--decay-rate 0.01] [--text-model clip|hash]

Prints the number of voxels assigned to the query, the heat statistics and the voxel the navigator would go to
(argmax of the heat, habitat_lang_robot.py:427-430); --save writes the (N,) heat vector as .npy."""
from __future__ import annotations

import argparse

import numpy as np


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--data-dir", required=True)
    ap.add_argument("--query", required=True)
    ap.add_argument("--config", default=None)
    ap.add_argument("--decay-rate", type=float, default=0.01)
    ap.add_argument("--text-model", choices=["clip", "hash"], default="clip",
                    help="clip = OpenAI CLIP ViT-B/32 on PyTorch-ROCm (as upstream); hash = model-free stand-in for smoke runs")
    ap.add_argument("--categories", default=None, help="comma separated list: preload scores_mat (VLMap.init_categories)")
    ap.add_argument("--save", default=None)
    args = ap.parse_args(argv)

    from avlmaps_amd import ops
    from avlmaps_amd.apps.common import HashClip, load_config
    from avlmaps_amd.map import AVLMap
    cfg = load_config(args.config)
    avlmap = AVLMap(cfg, data_dir=args.data_dir)
    if not avlmap.load_map(args.data_dir):
        raise SystemExit(1)
    vm = avlmap.vlmap
    if args.text_model == "hash":
        vm.clip_feat_dim = vm.grid_feat.shape[1]
        vm.clip_model = HashClip(vm.clip_feat_dim)
    else:
        vm._init_clip()
    cats = None
    if args.categories:
        cats = ["void"] + [c.strip() for c in args.categories.split(",")] + ["void"]   # upstream passes categories[1:-1]
    heat = avlmap.index_object(args.query, init_categories=cats, decay_rate=args.decay_rate)
    idx, val = ops.argmax_f32(heat)
    print(f"{int((heat == 1.0).sum())} of {len(heat)} voxels match {args.query!r}; heat>0 on {int((heat > 0).sum())}; "
          f"goal voxel id {idx} at grid_pos {vm.grid_pos[idx].tolist()} (heat {val:.3f})")
    if args.save:
        np.save(args.save, heat)
    return heat


if __name__ == "__main__":
    main()
