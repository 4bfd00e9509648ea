"""ctypes binding of libavlmaps_hip.so (include/avlmaps_hip.h).  Fails loudly when the library is missing."""
from __future__ import annotations

import ctypes as C
import os
import sys
import threading
from pathlib import Path

_PKG = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("AVLMAPS_HIP_LIB") or _PKG / "lib" / "libavlmaps_hip.so")       # (an empty variable = the stock library)

AVL_OK = 0
SIM_AUTO, SIM_EXACT, SIM_SPLIT_F16, SIM_EXACT_VALU, SIM_PREPARED, SIM_PREPARED24 = 0, 1, 2, 3, 4, 5


class AvlError(RuntimeError):
    pass


_lib = None

_vp, _i32, _i64, _f64, _sz = C.c_void_p, C.c_int32, C.c_int64, C.c_double, C.c_size_t
_SIGS = {
    "avl_last_error": (C.c_char_p, []),
    "avl_version": (C.c_int, []),
    "avl_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "avl_set_device": (C.c_int, [C.c_int]),
    "avl_get_device": (C.c_int, [C.POINTER(C.c_int)]),
    "avl_device_name": (C.c_int, [C.c_int, C.c_char_p, _sz]),
    "avl_device_sync": (C.c_int, []),
    "avl_stream_create": (C.c_int, [C.POINTER(_vp)]),
    "avl_stream_destroy": (C.c_int, [_vp]),
    "avl_stream_sync": (C.c_int, [_vp]),
    "avl_malloc": (C.c_int, [C.POINTER(_vp), _sz]),
    "avl_free": (C.c_int, [_vp]),
    "avl_mt19937_skip_shuffles": (C.c_int, [_vp, C.POINTER(C.c_int), _i64, _i64]),
    "avl_mt19937_shuffle_sample": (C.c_int, [_vp, C.POINTER(C.c_int), _i64, _i64, _vp, _vp]),
    "avl_host_alloc": (C.c_int, [C.POINTER(_vp), _sz]),
    "avl_host_free": (C.c_int, [_vp]),
    "avl_memset": (C.c_int, [_vp, C.c_int, _sz, _vp]),
    "avl_memcpy_h2d": (C.c_int, [_vp, _vp, _sz, _vp]),
    "avl_memcpy_d2h": (C.c_int, [_vp, _vp, _sz, _vp]),
    "avl_memcpy_d2d": (C.c_int, [_vp, _vp, _sz, _vp]),
    "avl_gather_rows": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp]),
    "avl_scatter_rows": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp]),
    "avl_rows_div_f32": (C.c_int, [_i64, C.c_int, _vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    "avl_merge_work_bytes": (C.c_int, [_i64, C.POINTER(_sz)]),
    "avl_merge_rows_work_bytes": (C.c_int, [_i64, C.POINTER(_sz)]),
    "avl_merge_rows_new": (C.c_int, [_i64, _i64, _vp, _vp, C.c_int, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _sz, _vp]),
    "avl_merge_dir_rows": (C.c_int, [_i64, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _sz, _vp]),
    "avl_merge_rows_other": (C.c_int, [_i64, _vp, _vp, _vp, _i64, _vp, _vp, _sz, _vp]),
    "avl_merge_partition": (C.c_int, [_i64, _vp, _vp, C.c_int, _vp, _vp, _vp, _vp, _sz, _vp]),
    "avl_merge_dir_scan": (C.c_int, [_i64, _vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "avl_merge_classify": (C.c_int, [_i64, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "avl_merge_side_pack": (C.c_int, [_i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "avl_merge_side_unpack": (C.c_int, [_i64, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp]),
    "avl_merge2_load": (C.c_int, []),
    "avl_merge2_prepare_work_bytes": (C.c_int, [_i64, C.POINTER(_sz)]),
    "avl_merge2_prepare": (C.c_int, [_i64, _vp, _vp, C.c_int, _i64, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "avl_merge2_max_chunks": (C.c_int, [C.c_int, C.POINTER(C.c_int)]),
    "avl_merge2_work_bytes": (C.c_int, [_i64, _i64, C.c_int, C.c_int, C.POINTER(_sz)]),
    "avl_merge2_plan": (C.c_int, [C.c_int, C.c_int, _vp, _i64, _vp, _vp, C.c_int, _i64, C.c_int, _i64, C.c_int, _vp, _sz, _vp, _vp, _vp]),
    "avl_builder_m2_pack": (C.c_int, [_vp, _i64, C.c_int, C.c_int, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "avl_merge2_side_state": (C.c_int, [_i64, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "avl_merge2_state_gather": (C.c_int, [_i64, _vp, _vp, _vp, _vp]),
    "avl_merge2_state_scatter": (C.c_int, [_i64, _vp, _vp, _vp, _vp]),
    "avl_merge2_fold": (C.c_int, [_i64, _i64, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    "avl_merge2_fold_work_bytes": (C.c_int, [_i64, C.c_int, C.POINTER(_sz)]),
    "avl_argsort_bits_work_bytes": (C.c_int, [_i64, C.c_int, C.c_int, C.POINTER(_sz)]),
    "avl_argsort_bits": (C.c_int, [_i64, _vp, C.c_int, C.c_int, _vp, _vp, _sz, _vp]),
    "avl_hbm_read_probe": (C.c_int, [_vp, _i64, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), _vp]),
    "avl_event_create": (C.c_int, [C.POINTER(_vp)]),
    "avl_event_destroy": (C.c_int, [_vp]),
    "avl_event_record": (C.c_int, [_vp, _vp]),
    "avl_event_sync": (C.c_int, [_vp]),
    "avl_stream_wait_event": (C.c_int, [_vp, _vp]),
    "avl_event_elapsed_ms": (C.c_int, [_vp, _vp, C.POINTER(C.c_float)]),
    "avl_sim_prepare_map": (C.c_int, [_vp, _i64, C.c_int, _i64, _vp, _vp]),
    "avl_sim_prepare_map24": (C.c_int, [_vp, _i64, C.c_int, _i64, _vp, _vp, _vp]),
    "avl_sim_scores_prepared24": (C.c_int, [_vp, _vp, _i64, C.c_int, _vp, C.c_int, _i64, _vp, _vp, _vp, _vp, _sz, _vp]),
    "avl_sim_workspace_bytes_n": (C.c_int, [_i64, C.c_int, C.c_int, C.POINTER(_sz)]),
    "avl_sim_scores_blocks": (C.c_int, [_vp, _vp, _i64, C.c_int, _i64, _vp, C.c_int, _i64, _vp, _vp, _vp, _vp, _vp, C.c_int, _vp, _sz, _vp]),
    "avl_sim_scores_prepared": (C.c_int, [_vp, _vp, _i64, C.c_int, _i64, _vp, C.c_int, _i64, _vp, _vp, _vp, _vp, _sz, _vp]),
    "avl_sim_scores": (C.c_int, [_vp, _i64, C.c_int, _i64, _vp, C.c_int, _i64, _vp, _vp, _vp, C.c_int, _vp]),
    "avl_sim_workspace_bytes": (C.c_int, [C.c_int, C.c_int, C.POINTER(_sz)]),
    "avl_sim_scores_ws": (C.c_int, [_vp, _i64, C.c_int, _i64, _vp, C.c_int, _i64, _vp, _vp, _vp, C.c_int, _vp, _sz, _vp]),
    "avl_sim_scores_host": (C.c_int, [_vp, _i64, C.c_int, _vp, C.c_int, _vp, _vp, _vp, C.c_int]),
    "avl_mask_from_argmax": (C.c_int, [_vp, _i64, _i32, _vp, _vp]),
    "avl_mask_bits_from_argmax": (C.c_int, [_vp, _i64, _i32, _vp, _vp]),
    "avl_argmax_f32": (C.c_int, [_vp, _i64, C.POINTER(_i64), C.POINTER(C.c_float), _vp]),
    "avl_topk_f32": (C.c_int, [_vp, _i64, C.c_int, _vp, _vp, _vp]),
    "avl_builder_create": (C.c_int, [C.POINTER(_vp), C.c_int, _f64, C.c_int, C.c_int, _i64]),
    "avl_builder_create_grid": (C.c_int, [C.POINTER(_vp), C.c_int, C.c_int, C.c_int, _f64, C.c_int, _i64]),
    "avl_builder_destroy": (C.c_int, [_vp]),
    "avl_builder_reset": (C.c_int, [_vp, _vp]),
    "avl_builder_enable_replay_log": (C.c_int, [_vp, _i64]),
    "avl_builder_integrate_frame": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp, C.c_int, _vp, C.c_int,
                                              C.c_int, _vp, _i64, _f64, _f64, _f64, _vp]),
    "avl_builder_integrate_batch": (C.c_int, [_vp, C.c_int, _vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp, C.c_int, _vp, C.c_int, C.c_int,
                                              _vp, _i64, _f64, _f64, _f64, _vp]),
    "avl_builder_integrate_frames": (C.c_int, [_vp, C.c_int, _vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp, C.c_int, _vp, C.c_int, C.c_int,
                                               _vp, _i64, _f64, _f64, _f64, _vp]),
    "avl_builder_integrate_frame_global": (C.c_int, [_vp, _vp, C.c_int, _f64, C.c_int, C.c_int, _vp, _vp, _vp, _vp, C.c_int, _vp,
                                                     C.c_int, C.c_int, _vp, _i64, _f64, _f64, _f64, _vp, _vp]),
    "avl_points_bbox": (C.c_int, [_vp, C.c_int, _f64, C.c_int, C.c_int, _vp, _vp, _vp, C.c_int, _f64, _f64, _vp, _vp]),
    "avl_builder_import_map": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "avl_builder_num_voxels": (C.c_int, [_vp, C.POINTER(_i64), _vp]),
    "avl_builder_num_points": (C.c_int, [_vp, C.POINTER(_i64), _vp]),
    "avl_builder_num_groups": (C.c_int, [_vp, C.POINTER(_i64), _vp]),
    "avl_builder_finalize": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "avl_builder_finalize_ex": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, _vp]),
    "avl_builder_export_raw": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "avl_finalize_raw": (C.c_int, [_i64, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "avl_builder_set_max_capacity": (C.c_int, [_vp, _i64]),
    "avl_builder_capacity": (C.c_int, [_vp, C.POINTER(_i64)]),
    "avl_builder_set_deferred_fuse": (C.c_int, [_vp, C.c_int, _vp]),
    "avl_builder_flush": (C.c_int, [_vp, _vp]),
    "avl_builder_release_scratch": (C.c_int, [_vp, _i64, _vp]),
    "avl_builder_scatter_merge": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _i64, _vp]),
    "avl_finalize_merged": (C.c_int, [_i64, _i64, C.c_int, C.c_int, C.c_int, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "avl_builder_replay_chain": (C.c_int, [_vp, _i64, _vp, C.c_uint64, _vp, _vp]),
    "avl_builder_drop_replay_cache": (C.c_int, [_vp, _vp]),
    "avl_builder_replay_prepare": (C.c_int, [_vp, _i64, _vp]),
    "avl_replay_state_apply": (C.c_int, [_i64, _vp, _vp, _vp, _vp]),
    "avl_rows_add_f64": (C.c_int, [_i64, C.c_int, _vp, _i64, _i64, _vp, _i64, _vp, _i64, _vp]),
    "avl_rows_add_f64_async": (C.c_int, [_i64, C.c_int, _vp, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _vp]),
    "avl_builder_export_rows_f32": (C.c_int, [_vp, _i64, _vp, _vp, _i64, _vp]),
    "avl_builder_export_rows_f64": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _i64, _vp]),
    "avl_finalize_side": (C.c_int, [_i64, _i64, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "avl_pool_label_2d": (C.c_int, [_vp, _vp, _i64, C.c_int, _vp, _vp]),
    "avl_rgb_topdown": (C.c_int, [_vp, _vp, _i64, C.c_int, _vp, _vp]),
    "avl_obstacle_map": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "avl_obstacle_scatter": (C.c_int, [_vp, _vp, _i64, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp]),
    "avl_heatmap_from_mask": (C.c_int, [_vp, _vp, _i64, _f64, _f64, _vp, _vp]),
    "avl_heat_plan_create": (C.c_int, [C.POINTER(_vp), _vp, _i64, _vp]),
    "avl_heat_plan_destroy": (C.c_int, [_vp]),
    "avl_heatmap_from_mask_planned": (C.c_int, [_vp, _vp, _f64, _f64, _vp, _vp]),
    "avl_lseg_merge_windows": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_int, C.c_int, _vp, _vp]),
}

EXPORTED_SYMBOLS = tuple(_SIGS)


def load():
    """dlopen the HIP library and declare signatures.  Raises AvlError if it has not been built."""
    global _lib
    if _lib is None:
        # torch bundles its own HIP runtime; it must be the first one loaded into the process so that this
        # library binds to the same runtime instance (device pointers and streams are shared with torch).
        if os.environ.get("AVLMAPS_NO_TORCH") != "1":
            try:
                import torch  # noqa: F401
            except Exception:
                pass
        if not LIB_PATH.exists():
            raise AvlError(
                f"{LIB_PATH} not found: build it with `python -m avlmaps_amd.build` (needs hipcc, gfx950). "
                "There is no CPU fallback for the avlmaps_amd compute path.")
        lib = C.CDLL(str(LIB_PATH))
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc: int, what: str = ""):
    if rc != AVL_OK:
        msg = load().avl_last_error().decode(errors="replace")
        raise AvlError(f"{what or 'libavlmaps_hip'} failed (status {rc}): {msg}")


_tls = threading.local()


def set_device(device: int) -> None:
    """avl_set_device for the calling thread, remembered so that helpers started from this thread (VLMap's map-upload thread) can
    bind to the same GPU without having to start the HIP runtime just to ask"""
    check(load().avl_set_device(int(device)), "avl_set_device")
    _tls.device = int(device)


def current_device(query_runtime: bool = True):
    """the calling thread's current GPU.  Once this thread has called set_device the HIP runtime is running, so HIP itself is asked
    (a torch.cuda.set_device or hipSetDevice made elsewhere in between is then seen: the remembered value alone went stale, ADVICE
    r3); before that: torch's current device if torch has initialised the GPU, else (query_runtime) what HIP reports -- which
    starts the runtime (~0.8 s the first time) -- else None"""
    if getattr(_tls, "device", None) is not None:
        n = C.c_int(0)
        if load().avl_get_device(C.byref(n)) == AVL_OK:
            _tls.device = n.value
        return _tls.device
    t = sys.modules.get("torch")
    try:
        if t is not None and t.cuda.is_initialized():
            return int(t.cuda.current_device())
    except Exception:
        pass
    if not query_runtime:
        return None
    n = C.c_int(0)
    return n.value if load().avl_get_device(C.byref(n)) == AVL_OK else 0


def device_count() -> int:
    n = C.c_int(0)
    rc = load().avl_device_count(C.byref(n))
    return n.value if rc == AVL_OK else 0


def require_gpu():
    if device_count() < 1:
        raise AvlError("no HIP device visible: avlmaps_amd has no CPU fallback (the reference CPU path lives upstream)")
