"""g1's knife-edge geometry vectors THROUGH THE HIP KERNEL (K1 = bp_voxelize_body of avl_builder.hip), via the C ABI.

g1_geometry.npz holds what the reference's own helpers returned (tools/gen_golden.py:gen_g1) for inputs chosen to sit on
the edges that decide a voxel index: coordinates that are exact multiples of the cell size and their nextafter neighbours,
points in (-cs, 0) (Python int() truncates toward zero: cell 0, in range), out-of-range points on every side, depths of
exactly 0.1 and 6.0 (strict bounds), and 1 200 camera-frame points through the two truncating projections.

How an exact float64 vector reaches the kernel: avl_builder_integrate_frame takes inv(calib) and the 4x4 pc_transform as
host float64 arguments.  With inv(calib) = [[0,0,x],[0,0,y],[0,0,z]], a unit depth and pixel 0 sampled, K1's
`p_local = (Kinv @ (u+.5, v+.5, 1)) * depth` IS (x, y, z) bit for bit (0*a = 0, fma(x, 1, 0) = x, x * 1.0 = x); with a zero
rotation block and translation t, `p_global = T @ [p_local; 1]` IS t.  Everything downstream of that -- true fp64 divide,
truncation toward zero, range test, projections, alpha -- is the kernel's own arithmetic on the golden inputs.

Reference lines: avlmaps/utils/mapping_utils.py:226-251 (depth2pc), :305-315 (transform_pc), :345-349 (base_pos2grid_id_3d),
:599-605 (project_point); avlmaps/map/vlmap_builder.py:129 (depth mask), :138 (_out_of_range), :141-143, :156-162."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GS, CS, VH = 1000, 0.05, 30


@pytest.fixture(scope="module")
def ops():
    from avlmaps_amd import _lib, ops
    _lib.load()
    _lib.require_gpu()
    return ops


def first_touch_unique(ids):
    """rows of `ids` in order of first appearance = the reference's voxel-id order (max_id increments on insert)"""
    seen, out = set(), []
    for r in map(tuple, ids.tolist()):
        if r not in seen:
            seen.add(r)
            out.append(r)
    return np.array(out, dtype=np.int32).reshape(-1, 3)


def inject_transform(p):
    T = np.zeros((4, 4))
    T[:3, 3] = p
    T[3, 3] = 1.0
    return T


UNIT_KINV = np.array([[0.0, 0, 0], [0, 0, 0], [0, 0, 1]])            # p_local = (0, 0, 1) for pixel 0 at depth 1
UNIT_K = np.array([[1.0, 0, 0.5], [0, 1, 0.5], [0, 0, 1]])            # projects it to pixel (0, 0) of a 1 x 1 image


def expected_voxels(g):
    ids = g["vox_ids"]
    inside = ((ids[:, 0] >= 0) & (ids[:, 0] < GS) & (ids[:, 1] >= 0) & (ids[:, 1] < GS) & (ids[:, 2] >= 0) & (ids[:, 2] < VH))
    return ids, inside, first_touch_unique(ids[inside])


@pytest.mark.parametrize("mode", ["frame_by_frame", "deferred", "batched", "c_frame_loop", "c_frame_loop_deferred"])
def test_g1_voxel_ids_through_the_kernel(ops, golden, mode):
    """every golden point of base_pos2grid_id_3d, one single-sample frame each: grid_pos == the reference's ids of the in-range
    points in first-touch order, out-of-range points (vlmap_builder.py:283-284) create nothing.  The modes are the
    instantiations of K1 (bp_voxelize_body): <0> inside voxelize_link_kernel / pipe_kernel (frame_by_frame / deferred / batched),
    and the PreGather pair of the frame loop in C -- <2>, the stateless half run one frame ahead, + <1>, the resume from the
    24-byte record (c_frame_loop*: avl_builder_integrate_frames over chunks of single-sample frames, per-frame T)"""
    from avlmaps_amd.device import DeviceArray
    g = golden("g1_geometry.npz")
    assert int(g["vox_gs"]) == GS and float(g["vox_cs"]) == CS
    pts = g["vox_pts"]
    ids, inside, want = expected_voxels(g)
    # the fixture does hold the edge cases this test exists for
    assert (~inside).sum() > 300 and inside.sum() > 500
    on_edge = np.isclose(pts / CS, np.round(pts / CS), rtol=0, atol=1e-9).any(axis=1)
    assert (on_edge & inside).sum() >= 15 and on_edge.sum() >= 900
    assert ((pts > -CS) & (pts < 0)).all(axis=1).sum() >= 100
    D = 4
    depth = DeviceArray.from_numpy(np.ones((1, 1), np.float32))
    rgb = DeviceArray.from_numpy(np.array([[[7, 8, 9]]], np.uint8))
    feat = DeviceArray.from_numpy(np.ones((1, 1, D), np.float32))
    idx = DeviceArray.from_numpy(np.zeros(1, np.int32))
    acc = ops.VoxelAccumulator(GS, CS, VH, D, capacity=4096, deferred_fuse=mode.endswith("deferred"))
    if mode.startswith("c_frame_loop"):
        B = 61                                   # chunk boundaries: the first frame of a call has nothing prepared for it
        for lo in range(0, len(pts), B):
            chunk = pts[lo:lo + B]
            n = len(chunk)
            plan = acc.make_batch_plan([depth] * n, [idx] * n, [feat] * n, [rgb] * n)
            acc.integrate_frames(plan, UNIT_K, np.stack([inject_transform(p) for p in chunk]), frame_idx0=lo, calib_inv=UNIT_KINV)
    elif mode == "batched":
        B = 64
        for lo in range(0, len(pts), B):
            chunk = pts[lo:lo + B]
            n = len(chunk)
            acc.integrate_batch([depth] * n, UNIT_K, np.stack([inject_transform(p) for p in chunk]), [idx] * n, [feat] * n, [rgb] * n,
                                frame_idx0=lo, calib_inv=UNIT_KINV)
    else:
        for i, p in enumerate(pts):
            acc.integrate_frame(depth, UNIT_K, inject_transform(p), idx, feat, rgb, frame_idx=i, calib_inv=UNIT_KINV)
    assert acc.num_voxels() == len(want)
    out = acc.finalize()
    assert np.array_equal(out["grid_pos"], want)                       # bit-exact ids, reference id order, nothing out of range
    occ = out["occupied_ids"]
    assert (occ >= 0).sum() == len(want)
    assert np.array_equal(occ[want[:, 0], want[:, 1], want[:, 2]], np.arange(len(want)))
    # p_local = (0, 0, 1) for every sample: alpha = exp(-1 / 1.2); a voxel hit k times weighs k * alpha (float32 running sum)
    cnt = np.zeros(len(want))
    rowof = {tuple(r): i for i, r in enumerate(want.tolist())}
    for r in ids[inside].tolist():
        cnt[rowof[tuple(r)]] += 1
    np.testing.assert_allclose(out["weight"], cnt * np.exp(-1.0 / 1.2), rtol=2e-6)
    # every sample has the colour (7, 8, 9): the once-truncated weighted mean of k equal colours is that colour (finalize_kernel
    # keeps (sum alpha c) / (sum alpha) = c - 1 ulp from truncating to c - 1)
    assert np.all(out["grid_rgb"] == np.array([7, 8, 9], np.uint8))


@pytest.mark.parametrize("replay_log", [False, True])
def test_g1_projections_and_alpha_through_the_kernel(ops, golden, replay_log):
    """project_point with the calibration matrix (rgb pixel, NumPy's negative-index wrap) and with get_sim_cam_mat(347, 520)
    (feature pixel, bounds test of vlmap_builder.py:161), int() truncation of x/z - 0.5 included, for g1's 1 200 camera-frame
    points; each point is parked in a voxel of its own so that the voxel's colour / feature say which pixels K1 read."""
    from avlmaps_amd.device import DeviceArray
    g = golden("g1_geometry.npz")
    P = g["proj_pts"]
    K = g["proj_calib"]
    H, W, Hf, Wf, D = 720, 1080, 347, 520, 4
    assert np.array_equal(g["simcam_347_520"], np.array([[260.0, 0, 260.0], [0, 260.0, 173.5], [0, 0, 1]]))
    rc, rs = g["proj_calib_xyz"], g["proj_sim_xyz"]
    pxr, pyr = rc[:, 0].astype(np.int64), rc[:, 1].astype(np.int64)
    pxs, pys = rs[:, 0].astype(np.int64), rs[:, 1].astype(np.int64)
    in_feat = (pxs >= 0) & (pxs < Wf) & (pys >= 0) & (pys < Hf)                     # vlmap_builder.py:161
    pxw, pyw = np.where(pxr < 0, pxr + W, pxr), np.where(pyr < 0, pyr + H, pyr)    # rgb[py, px] with negative wrap
    in_rgb = (pxw >= 0) & (pxw < W) & (pyw >= 0) & (pyw < H)                        # outside: the reference raises IndexError
    assert (in_feat & in_rgb).sum() > 400 and (~in_feat & in_rgb).sum() > 100 and (~in_rgb).sum() > 100
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    rgb_img = np.stack([xx & 255, yy & 255, (xx >> 8) | ((yy >> 8) << 4)], axis=-1).astype(np.uint8)
    fy, fx = np.meshgrid(np.arange(Hf), np.arange(Wf), indexing="ij")
    feat_img = np.stack([fx + 1.0, fy + 1.0, np.ones_like(fx, dtype=np.float64), np.zeros_like(fx, dtype=np.float64)], -1).astype(np.float32)
    depth = DeviceArray.from_numpy(np.ones((H, W), np.float32))
    rgb = DeviceArray.from_numpy(rgb_img)
    feat = DeviceArray.from_numpy(feat_img)
    idx = DeviceArray.from_numpy(np.zeros(1, np.int32))
    cells = np.stack([100 + np.arange(len(P)) % 40, 100 + np.arange(len(P)) // 40, np.full(len(P), 3)], axis=1)

    def park(acc, i):
        # zero rotation + the centre of cell i as translation: the point becomes voxel cells[i] whatever p_local is
        t = np.array([(GS / 2 - cells[i, 0]) * CS + CS / 2, (GS / 2 - cells[i, 1]) * CS + CS / 2, cells[i, 2] * CS + CS / 2])
        kinv = np.zeros((3, 3))
        kinv[:, 2] = P[i]                        # p_local = P[i] exactly (module docstring)
        acc.integrate_frame(depth, K, inject_transform(t), idx, feat, rgb, frame_idx=i, calib_inv=kinv)

    acc = ops.VoxelAccumulator(GS, CS, VH, D, capacity=2048)
    if replay_log:                               # VLMapBuilder's default: sequential weight / uint8 colour replayed at finalisation
        acc.enable_replay_log(int(in_rgb.sum()))
    for i in np.nonzero(in_rgb)[0]:              # the other points make the reference raise IndexError at rgb[py, px]: see below
        park(acc, int(i))
    out = acc.finalize()
    keep = in_feat & in_rgb
    assert np.array_equal(out["grid_pos"], cells[keep].astype(np.int32))           # exactly the points the reference keeps
    got_rgb = out["grid_rgb"].astype(np.int64)
    gx = got_rgb[:, 0] | ((got_rgb[:, 2] & 15) << 8)
    gy = got_rgb[:, 1] | ((got_rgb[:, 2] >> 4) << 8)
    assert np.array_equal(gx, pxw[keep]) and np.array_equal(gy, pyw[keep])          # the rgb pixel the reference reads
    gf = out["grid_feat"].astype(np.float64)
    w = out["weight"].astype(np.float64)
    # a new voxel stores feat * alpha with weight alpha (vlmap_builder.py:166-168): feat = grid_feat / weight
    assert np.array_equal(np.rint(gf[:, 0] / gf[:, 2]).astype(np.int64) - 1, pxs[keep])
    assert np.array_equal(np.rint(gf[:, 1] / gf[:, 2]).astype(np.int64) - 1, pys[keep])
    alpha = np.exp(-np.sum(np.square(P[keep]), axis=1) / (2 * 0.6))                 # vlmap_builder.py:156-158
    np.testing.assert_allclose(w, alpha.astype(np.float32), rtol=3e-7, atol=1e-45)
    np.testing.assert_allclose(gf[:, 2], alpha, rtol=3e-7, atol=1e-45)
    # error parity: a point whose rgb projection leaves the image is the reference's IndexError (vlmap_builder.py:144); the
    # library drops it, flags it and reports AVL_ERR_INVALID at the next read of the map
    bad = ops.VoxelAccumulator(GS, CS, VH, D, capacity=64)
    park(bad, int(np.nonzero(~in_rgb)[0][0]))
    with pytest.raises(Exception, match="outside the RGB image"):
        bad.num_voxels()


def test_g1_depth_bounds_and_backprojection_through_the_kernel(ops, golden):
    """depth2pc + the strict depth mask + transform_pc on g1's 20 x 28 depth image (pixels of exactly 0.1, 6.0 and
    float32(0.1) + 1e-6 included): (a) per pixel, does the kernel keep what d2p_mask keeps; (b) the whole frame through
    tpc_T: voxel ids of the reference's transformed points (tpc_out), bit-exact and in the reference's order."""
    from avlmaps_amd.device import DeviceArray
    g = golden("g1_geometry.npz")
    depth_np, K, mask, tpc = g["d2p_depth"], g["d2p_K"], g["d2p_mask"], g["tpc_out"]
    H, W = depth_np.shape
    assert depth_np[0, 0] == np.float32(0.1) and depth_np[0, 1] == np.float32(6.0) and depth_np[0, 2] > np.float32(0.1)
    # float32(0.1) is 0.100000001490116 in float64: INSIDE the strict lower bound; 6.0 is exact: outside the strict upper bound
    assert mask[0] and not mask[1] and mask[2] and (~mask).sum() > 50
    D = 4
    depth = DeviceArray.from_numpy(depth_np)
    rgb = DeviceArray.from_numpy(np.zeros((H, W, 3), np.uint8))
    feat = DeviceArray.from_numpy(np.ones((H, W, D), np.float32))
    # (a) one pixel per frame, each parked in its own cell by a zero-rotation transform: a voxel appears iff the pixel survives
    acc = ops.VoxelAccumulator(GS, CS, VH, D, capacity=1024)
    cells = np.stack([200 + np.arange(H * W) % 28, 200 + np.arange(H * W) // 28, np.full(H * W, 5)], axis=1)
    for i in range(H * W):
        t = np.array([(GS / 2 - cells[i, 0]) * CS + CS / 2, (GS / 2 - cells[i, 1]) * CS + CS / 2, cells[i, 2] * CS + CS / 2])
        acc.integrate_frame(depth, K, inject_transform(t), DeviceArray.from_numpy(np.array([i], np.int32)), feat, rgb, frame_idx=i)
    out = acc.finalize()
    assert np.array_equal(out["grid_pos"], cells[mask].astype(np.int32))
    # alpha of pixel i = exp(-|pc_i|^2 / 1.2) with the reference's back-projected point
    pc = g["d2p_pc"][:, mask]
    np.testing.assert_allclose(out["weight"], np.exp(-np.sum(pc * pc, axis=0) / 1.2).astype(np.float32), rtol=3e-7, atol=1e-45)
    # (b) all pixels of the frame in one launch through the golden pose; coarse cells so that most points are inside the grid
    gs, cs, vh = 200, 0.25, 40
    x, y, z = tpc                                                                  # transform_pc's output for the masked points
    ids = np.array([[int(gs / 2 - int(a / cs)), int(gs / 2 - int(b / cs)), int(c / cs)] for a, b, c in zip(x, y, z)])
    # tpc_out holds every pixel's point (depth2pc returns all of them and the mask separately)
    assert tpc.shape[1] == H * W
    inside = mask & (ids[:, 0] >= 0) & (ids[:, 0] < gs) & (ids[:, 1] >= 0) & (ids[:, 1] < gs) & (ids[:, 2] >= 0) & (ids[:, 2] < vh)
    assert inside.sum() > 50
    acc2 = ops.VoxelAccumulator(gs, cs, vh, D, capacity=1024)
    acc2.integrate_frame(depth, K, g["tpc_T"], np.arange(H * W, dtype=np.int32), feat, rgb, frame_idx=0)
    out2 = acc2.finalize()
    assert np.array_equal(out2["grid_pos"], first_touch_unique(ids[inside]))
