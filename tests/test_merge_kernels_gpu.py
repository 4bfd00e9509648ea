"""GPU checks of the round-4 C-ABI entry points behind the multi-GPU merge and the pinned frame staging, each against NumPy on
the accumulators the library itself exports (avl_builder_export_raw)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    from avlmaps_amd import _lib, ops
    from avlmaps_amd.device import DeviceArray
    lib = _lib.load()
    _lib.require_gpu()
    return _lib, lib, ops, DeviceArray


def small_build(ops, golden, D=None):
    from oracle import avl_oracle as O                      # (the checker's pose chain, as in tests/test_builder_gpu.py)
    from test_builder_gpu import run_gpu_builder
    g = golden("g2a_builder_small.npz")
    Ts = O.pc_transforms(g["poses_rt"], g["base_transform"], g["base2cam_tf"])
    acc = run_gpu_builder(ops, int(g["gs"]), float(g["cs"]), float(g["camera_height"]), g["calib"], Ts, g["depths"], g["rgbs"],
                          g["feats"], g["samples"], capacity=4000, replay=True)
    return g, acc


def test_export_rows_match_the_single_process_finalisation(env, golden):
    """avl_builder_export_rows_f32 = the finished float32 row of a voxel one rank touched alone: bit for bit what finalize() emits
    for that voxel; avl_builder_export_rows_f64 = sum_feat minus the first-touch term where the caller owns it"""
    _lib, lib, ops, DeviceArray = env
    g, acc = small_build(ops, golden)
    n, D = acc.num_voxels(), acc.D
    raw = acc.export_raw()
    fin = acc.finalize()
    order = np.argsort(raw["first_key"].astype(np.uint64), kind="stable")         # finalize emits voxels in first-touch-key order
    rng = np.random.default_rng(0)
    slots = rng.permutation(n)[: max(1, n // 2)].astype(np.int32)
    d_slots = DeviceArray.from_numpy(slots)
    out32 = DeviceArray((len(slots), D + 8), np.float32)                           # a padded row stride
    _lib.check(lib.avl_builder_export_rows_f32(acc._h, len(slots), d_slots.ptr, out32.ptr, D + 8, None), "export_rows_f32")
    rows32 = out32.numpy()[:, :D]
    row_of_slot = np.empty(n, np.int64)
    row_of_slot[order] = np.arange(n)
    assert np.array_equal(rows32, fin["grid_feat"][row_of_slot[slots]])           # bit-identical to the single-process map
    a1 = raw["first_alpha"][slots]
    want32 = (((a1 * a1)[:, None] * raw["first_feat"][slots].astype(np.float64) + raw["sum_feat"][slots]) / raw["sum_w4"][slots, :1])
    assert np.array_equal(rows32, want32.astype(np.float32))
    own = (rng.random(len(slots)) < 0.5).astype(np.uint8)
    out64 = DeviceArray((len(slots), D), np.float64)
    _lib.check(lib.avl_builder_export_rows_f64(acc._h, len(slots), d_slots.ptr, DeviceArray.from_numpy(own).ptr, out64.ptr, D, None), "export_rows_f64")
    want64 = np.where(own != 0, a1 * a1, a1)[:, None] * raw["first_feat"][slots].astype(np.float64) + raw["sum_feat"][slots]
    assert np.array_equal(out64.numpy(), want64)
    # argument checks: more slots than voxels, a row stride below D
    assert lib.avl_builder_export_rows_f32(acc._h, n + 1, d_slots.ptr, out32.ptr, D + 8, None) != 0
    assert lib.avl_builder_export_rows_f32(acc._h, 1, d_slots.ptr, out32.ptr, D - 1, None) != 0
    acc.close()


def test_finalize_side_is_the_side_part_of_finalize_merged(env, golden):
    """avl_finalize_side (grid_pos / weight / grid_rgb / occupied_ids from cells + [sum alpha, sum alpha rgb]) against
    avl_finalize_merged on the same rows"""
    _lib, lib, ops, DeviceArray = env
    g, acc = small_build(ops, golden)
    n, D = acc.num_voxels(), acc.D
    raw = acc.export_raw()
    gs, vh = acc.gs, acc.vh
    accbuf = np.concatenate([raw["sum_feat"], raw["sum_w4"]], axis=1)
    d_cell, d_acc = DeviceArray.from_numpy(raw["cell"]), DeviceArray.from_numpy(accbuf)
    ref = {k: DeviceArray(s, t) for k, s, t in (("pos", (n, 3), np.int32), ("w", (n,), np.float32), ("rgb", (n, 3), np.uint8), ("occ", (gs, gs, vh), np.int32))}
    got = {k: DeviceArray(v.shape, v.dtype) for k, v in ref.items()}
    for d in (ref["occ"], got["occ"]):
        _lib.check(lib.avl_memset(d.ptr, 0xFF, d.nbytes, None))
    feat = DeviceArray((n, D), np.float32)
    _lib.check(lib.avl_finalize_merged(n, 5, D, gs, vh, d_cell.ptr, d_acc.ptr, D + 4, feat.ptr, ref["pos"].ptr, ref["w"].ptr, ref["rgb"].ptr, ref["occ"].ptr, None))
    d_w4 = DeviceArray.from_numpy(np.ascontiguousarray(raw["sum_w4"]))
    _lib.check(lib.avl_finalize_side(n, 5, gs, vh, d_cell.ptr, d_w4.ptr, got["pos"].ptr, got["w"].ptr, got["rgb"].ptr, got["occ"].ptr, None), "finalize_side")
    for k in ref:
        assert np.array_equal(ref[k].numpy(), got[k].numpy()), k
    assert lib.avl_finalize_side(n, (1 << 31) - 2, gs, vh, d_cell.ptr, d_w4.ptr, None, None, None, None, None) != 0        # ids beyond int32
    acc.close()


def test_replay_chain_in_two_calls_equals_one(env, golden):
    """avl_builder_replay_chain with a negative index per skipped slot: replaying an arbitrary half of the voxels first and the rest
    in a second call (the round-4 merge: voxels without a predecessor, then the ones whose state arrived) is the single call"""
    _lib, lib, ops, DeviceArray = env
    g, acc = small_build(ops, golden)
    n = acc.num_voxels()
    ar = np.arange(n, dtype=np.int64)
    one = DeviceArray((n, 3), np.int64).zero_()
    _lib.check(lib.avl_builder_replay_chain(acc._h, n, DeviceArray.from_numpy(ar).ptr, C.c_uint64((1 << 64) - 1), one.ptr, None), "replay_chain")
    sel = np.random.default_rng(1).random(n) < 0.5
    two = DeviceArray((n, 3), np.int64).zero_()
    for mask in (sel, ~sel):
        idx = np.where(mask, ar, -1)
        _lib.check(lib.avl_builder_replay_chain(acc._h, n, DeviceArray.from_numpy(idx).ptr, C.c_uint64((1 << 64) - 1), two.ptr, None), "replay_chain")
    assert np.array_equal(one.numpy(), two.numpy())
    # ... and the final states are the weights / colours of the exact single-process finalisation
    fin = acc.finalize()
    order = np.argsort(acc.export_raw()["first_key"].astype(np.uint64), kind="stable")
    w = DeviceArray((n,), np.float32).zero_()
    rgb = DeviceArray((n, 3), np.uint8).zero_()
    st = DeviceArray.from_numpy(np.ascontiguousarray(one.numpy()[order]))
    _lib.check(lib.avl_replay_state_apply(n, st.ptr, w.ptr, rgb.ptr, None))
    assert np.array_equal(w.numpy(), fin["weight"]) and np.array_equal(rgb.numpy(), fin["grid_rgb"])
    acc.close()


def test_rows_add_async_flags_an_out_of_range_row(env):
    _lib, lib, ops, DeviceArray = env
    rng = np.random.default_rng(2)
    src = rng.standard_normal((6, 5))
    dst0 = rng.standard_normal((4, 7))
    rows = np.array([3, 0, 2, 9, 1, -1], np.int64)                                # 9 and -1 lie outside the 4-row block
    d_dst, flag = DeviceArray.from_numpy(dst0.copy()), DeviceArray((1,), np.int32).zero_()
    _lib.check(lib.avl_rows_add_f64_async(6, 5, DeviceArray.from_numpy(rows).ptr, 0, 4, DeviceArray.from_numpy(src).ptr, 5, d_dst.ptr, 7, flag.ptr, None),
               "rows_add_async")
    want = dst0.copy()
    for i, r in enumerate(rows):
        if 0 <= r < 4:
            want[r, :5] += src[i]
    assert np.array_equal(d_dst.numpy(), want) and int(flag.numpy()[0]) == 1


def test_frame_stager_hands_over_exact_copies_and_follows_shape_changes(env):
    """device.FrameStager: what the fusing stream reads after avl_stream_wait_event is the frame that was staged, slot reuse waits
    for the release event, a change of the frame shape re-sizes the ring"""
    _lib, lib, ops, DeviceArray = env
    from avlmaps_amd.device import FrameStager
    st = FrameStager(3)
    rng = np.random.default_rng(3)
    held = []                                                     # frames staged but not yet consumed (a queue between the threads)
    try:
        for k, (H, W, P) in enumerate([(12, 16, 40), (12, 16, 33), (12, 16, 40), (12, 16, 7), (12, 16, 40), (20, 8, 50), (20, 8, 50), (20, 8, 9)]):
            depth = rng.random((H, W)).astype(np.float32)
            rgb = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
            samples = rng.integers(0, H * W, P).astype(np.int32)
            held.append((st.stage(depth, rgb, samples), depth, rgb, samples))
            while len(held) > (2 if k != 5 else 0):               # consume with a lag of two frames -- also across the change of shape
                sf, d0, r0, s0 = held.pop(0)                      # at k == 5 (old-ring frames still held when the new ring starts)
                st.acquire(sf)                                    # the (null) stream waits for the copies on the device
                back = [np.empty(v.shape, v.dtype) for v in (sf.depth, sf.rgb, sf.samples)]
                for b, v in zip(back, (sf.depth, sf.rgb, sf.samples)):
                    _lib.check(lib.avl_memcpy_d2h(b.ctypes.data, v.ptr, v.nbytes, None))
                st.release(sf)
                assert np.array_equal(back[0], d0) and np.array_equal(back[1], r0) and np.array_equal(back[2], s0), k
        assert len(st.retired) == 1                               # the 12 x 16 ring was retired, not freed under the queued frames
    finally:
        st.close()
    assert lib.avl_stream_wait_event(None, None) != 0                 # a null event is an error, not a crash


def test_argsort_bits_is_torchs_stable_argsort(env):
    """avl_argsort_bits (the merge plan's narrow-key sorts, parallel._argsort_bits): the permutation of torch.argsort(stable=True)
    for int64 and int32 keys, few and many bits, ties in position order; and the plan built with it equals the plan built with
    torch's own sorts"""
    import torch
    from avlmaps_amd import parallel
    g = torch.Generator(device="cpu").manual_seed(11)
    for dtype, bits, n in ((torch.int64, 3, 100_001), (torch.int64, 22, 300_000), (torch.int32, 25, 250_000), (torch.int64, 1, 50_000),
                           (torch.int64, 40, 65_537)):
        t = torch.randint(0, 1 << min(bits, 62), (n,), generator=g, dtype=torch.int64).to(dtype).cuda()
        got = parallel._argsort_bits(t, bits)
        want = torch.argsort(t, stable=True)
        assert got.dtype == torch.int64 and torch.equal(got, want), (dtype, bits, n)
    n = 200_000
    cell = torch.randperm(4 * n, generator=g)[:n].to(torch.int32).cuda()
    key = torch.randperm(n, generator=g).cuda()
    a = parallel.plan_merge_directory(cell, key, local=True)
    small = parallel._argsort_bits
    try:
        parallel._argsort_bits = lambda t, bits: torch.argsort(t, stable=True)
        b = parallel.plan_merge_directory(cell, key, local=True)
    finally:
        parallel._argsort_bits = small
    assert a.M == b.M == n and torch.equal(a.row_of_slot, b.row_of_slot) and torch.equal(a.prev, b.prev) and torch.equal(a.next, b.next)
    assert torch.equal(torch.sort(a.row_of_slot).values, torch.arange(n, device="cuda"))
    assert torch.equal(a.row_of_slot[torch.argsort(key)], torch.arange(n, device="cuda"))       # one rank: rows in key order


def test_merge_plan_kernels_equal_the_tensor_code(env, monkeypatch):
    """csrc/avl_merge.hip (avl_merge_partition / _dir_scan / _classify) against the tensor code of plan_merge_directory on the same
    device tensors: every output of the plan identical -- one rank here; the N-rank builds of test_api_gpu run the kernels with
    real neighbours and compare the merged map with the single-rank one"""
    import torch
    from avlmaps_amd import parallel
    g = torch.Generator(device="cpu").manual_seed(3)
    for n in (0, 1, 777, 200_003):
        cell = torch.randperm(max(4 * n, 8), generator=g)[:n].to(torch.int32).cuda()
        key = ((torch.randint(0, 50, (n,), generator=g) << 32) | torch.randperm(max(n, 1), generator=g)[:n]).cuda()
        monkeypatch.setenv("AVLMAPS_MERGE_KERNELS", "1")
        a = parallel.plan_merge_directory(cell, key, local=True, grow_row=n // 2, aux=(1, 0))
        monkeypatch.setenv("AVLMAPS_MERGE_KERNELS", "0")
        b = parallel.plan_merge_directory(cell, key, local=True, grow_row=n // 2, aux=(1, 0))
        assert a.M == b.M == n and a.grow_key == b.grow_key and a.dir_entries == b.dir_entries and a.counts == b.counts
        assert list(a.n_prev) == list(b.n_prev) and list(a.n_next) == list(b.n_next)
        for f in ("row_of_slot", "is_new", "prev", "next"):
            assert torch.equal(getattr(a, f), getattr(b, f)), (n, f)
        assert np.array_equal(a.aux_all, b.aux_all)


def test_side_records_pack_and_unpack_equal_the_tensor_code(env):
    """avl_merge_side_pack / avl_merge_side_unpack against the tensor expressions of merge_raw_sharded / _fold_mixed (the CPU twin
    of the device merge): the 64-byte records, and what an owner reads out of them; plus the narrow rows_add on a column window"""
    import torch
    _lib, lib, ops, DeviceArray = env
    i64 = torch.int64
    g = torch.Generator(device="cpu").manual_seed(11)
    for n in (1, 5, 1000, 70_001):
        order = torch.randperm(n, generator=g).cuda()
        rows_sorted = (torch.sort(torch.randperm(3 * n, generator=g)[:n]).values + 40).cuda()
        single = (torch.rand(n, generator=g) < 0.6).cuda()
        cell = torch.randint(0, 2**31 - 1, (n,), generator=g, dtype=torch.int32).cuda()
        w4 = torch.randn((n, 4), generator=g, dtype=torch.float64).cuda()
        state = torch.randint(-2**62, 2**62, (n, 3), generator=g, dtype=i64).cuda()
        state[:, 2] |= 1 << 32                                      # started
        nxt = torch.where(torch.rand(n, generator=g) < 0.5, -1, 3).to(i64).cuda()
        for have_state, have_next in ((True, True), (True, False), (False, True)):
            side = torch.full((n, 8), -7, dtype=i64, device="cuda")
            _lib.check(lib.avl_merge_side_pack(n, order.data_ptr(), rows_sorted.data_ptr(), single.view(torch.uint8).data_ptr(), cell.data_ptr(),
                                               w4.data_ptr(), state.data_ptr() if have_state else None, nxt.data_ptr() if have_next else None,
                                               side.data_ptr(), None), "side_pack")
            st = state if have_state else torch.zeros_like(state)
            if have_next:
                st = torch.where((nxt < 0)[:, None], st, torch.zeros_like(st))
            want = torch.empty((n, 8), dtype=i64, device="cuda")
            want[:, 0] = rows_sorted | (cell[order].to(i64) << 32) | (single.to(i64) << 63)
            want[:, 1:5] = w4[order].view(i64)
            want[:, 5:8] = st[order]
            assert torch.equal(side, want), (n, have_state, have_next)
        # owner side: a block [r0, r0 + n_own) that holds every row but the last two records' (those must raise the flag)
        r0, n_own = 40, int(rows_sorted[max(n - 3, 0)].item()) - 40 + 1 if n > 2 else 3 * n + 1
        rows = torch.full((n,), -5, dtype=i64, device="cuda")
        own_cell = torch.zeros(n_own, dtype=torch.int32, device="cuda")
        own_state = torch.zeros((n_own, 3), dtype=i64, device="cuda")
        flag = torch.zeros(1, dtype=torch.int32, device="cuda")
        _lib.check(lib.avl_merge_side_unpack(n, side.data_ptr(), r0, n_own, rows.data_ptr(), own_cell.data_ptr(), own_state.data_ptr(),
                                             flag.data_ptr(), None), "side_unpack")
        word = side[:, 0]
        rel = (word & 0xFFFFFFFF) - r0
        ok = (rel >= 0) & (rel < n_own)
        assert int(flag.item()) == int((~ok).any().item())
        assert torch.equal(rows, rel.clamp(0, n_own - 1))
        want_cell = torch.zeros_like(own_cell)
        want_cell[rel[ok]] = ((word[ok] >> 32) & 0x7FFFFFFF).to(torch.int32)
        want_state = torch.zeros_like(own_state)
        fin = ok & ((side[:, 7] >> 32) != 0)
        want_state[rel[fin]] = side[fin][:, 5:8]
        assert torch.equal(own_cell, want_cell) and torch.equal(own_state, want_state)
        # the four sums of the records, added straight out of the (n, 8) buffer (ld = 8)
        dst = torch.zeros((n_own, 4), dtype=torch.float64, device="cuda")
        k = int(ok.sum().item())
        src = side[:, 1:5].view(torch.float64)
        _lib.check(lib.avl_rows_add_f64_async(k, 4, rows.data_ptr(), 0, n_own, src.data_ptr(), 8, dst.data_ptr(), 4, flag.data_ptr(), None), "rows_add")
        want = torch.zeros_like(dst)
        want[rel[:k]] = src[:k]
        assert torch.equal(dst, want)
