#!/usr/bin/env python3
"""one line per bench.py --workload build record: merge phases (rank 0) and every rank's compute / local voxels"""
import json
import sys

args = [a for a in sys.argv[1:] if not a.startswith("--")]
# --json=<file>: the per-rank record bench.py's projected_speedup_8gpu reads (profiles/r06_merge_rehearsal_8ranks.json)
json_out = next((a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--json=")), None)
for path in args:
    d = json.loads([l for l in open(path) if l.startswith("{")][-1])
    e = d["extra"] if "voxels_merged" in d.get("extra", {}) else d["extra"]["map_build_strong"]
    mb = e["merge_breakdown"]
    r = lambda x: round(x * 1e3, 2)
    print(path, f"frames/s {e['frames_per_s']:.0f}  total {e['seconds']:.3f} s  fuse {e['fuse_seconds_max_rank']:.3f} s  M {e['voxels_merged']}")
    print("   rank0 wall ms   ", {k: r(v) for k, v in mb["wall_s"].items()})
    print("   rank0 compute ms", {k: r(v) for k, v in mb["compute_s"].items()}, "wait for shared GPU s", round(mb.get("shared_gpu_wait_s") or 0, 3))
    print("   per rank: compute ms", [r(p["compute_total_s"]) for p in mb["per_rank"]], "local voxels", [p["local_voxels"] for p in mb["per_rank"]],
          "single-rank voxels", [p.get("single_rank_voxels") for p in mb["per_rank"]], "null launch us", [round(p.get("null_launch_us") or 0, 1) for p in mb["per_rank"]])
    sm = mb.get("second_merge")
    if sm:
        print("   second merge (buffers in place): per rank compute ms", [r(p["compute_total_s"]) for p in sm["per_rank"]])
    print("   payload MB sent", [round(p["payload_bytes_sent"] / 1e6) for p in mb["per_rank"]], "fp64 form", [round((p.get("payload_bytes_fp64_form") or 0) / 1e6) for p in mb["per_rank"]])
    if json_out:
        import os
        rec = dict(what="merge of a 10 000-frame build sharded over 8 ranks that take turns on ONE MI355X (gloo, AVLMAPS_SHARED_GPU_LOCK): per-rank "
                        "compute = wall time of the merge minus time inside collectives and waiting for the shared GPU.  Eight processes on one device "
                        "disturb each other at random (one process freeing or allocating device memory stalls the kernels of the others for "
                        "milliseconds): every launch of the rehearsal shows one to three ranks at 5-8 ms, different ranks each time; "
                        "per_rank_compute_ms is the per-rank minimum over the launches listed in `launches`",
                   world_size=mb.get("world_size"), merged_voxels=mb.get("merged_voxels"), plan=mb.get("plan"), trajectory=e.get("trajectory"),
                   per_rank_local_voxels=[p["local_voxels"] for p in mb["per_rank"]],
                   per_rank_payload_MB=[round(p["payload_bytes_sent"] / 1e6) for p in mb["per_rank"]], launches=[])
        if os.path.exists(json_out):
            old = json.loads(open(json_out).read())
            if old.get("merged_voxels") == rec["merged_voxels"] and old.get("plan") == rec["plan"]:
                rec["launches"] = old.get("launches", [])
        rec["launches"].append(dict(source=path, per_rank_compute_ms=[r(p["compute_total_s"]) for p in mb["per_rank"]],
                                    per_rank_compute_phases_ms=[{k: r(v) for k, v in p["compute_s"].items()} for p in mb["per_rank"]],
                                    merge_first_call_s=e.get("merge_first_call_s")))
        rec["per_rank_compute_ms"] = [min(l["per_rank_compute_ms"][k] for l in rec["launches"]) for k in range(len(mb["per_rank"]))]
        open(json_out, "w").write(json.dumps(rec, indent=1))
