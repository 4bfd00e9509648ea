"""CPU tests of the host-side mirror of the reference interface (no GPU compute)."""
import hashlib
import json
from pathlib import Path

import numpy as np
import pytest

from avlmaps_amd.map.map import Map
from avlmaps_amd.map.vlmap_builder import VLMapBuilder
from avlmaps_amd.utils import mapping_utils as mu
from avlmaps_amd.utils.clip_utils import multiple_templates

GOLDEN = Path(__file__).resolve().parent / "golden"


class Cfg(dict):
    __getattr__ = dict.__getitem__


def make_cfg(g):
    return Cfg(map_type="vlmap", grid_size=int(g["gs"]), cell_size=float(g["cs"]), depth_sample_rate=int(g["rate"]),
               cam_calib_mat=[float(x) for x in g["calib"]],
               pose_info=Cfg(pose_type="mobile_base", camera_height=float(g["camera_height"]),
                             base2cam_rot=[float(x) for x in g["base2cam_rot"]],
                             base_forward_axis=list(g["base_axes"][0]), base_left_axis=list(g["base_axes"][1]),
                             base_up_axis=list(g["base_axes"][2])))


def test_templates_match_reference_hash():
    meta = json.loads((GOLDEN / "templates.json").read_text())
    assert len(multiple_templates) == meta["n_templates"] == 63
    assert hashlib.sha256("\n".join(multiple_templates).encode()).hexdigest() == meta["sha256_of_newline_joined"]


def test_geometry_helpers_match_reference(golden):
    g = golden("g1_geometry.npz")
    for v, tf in zip(g["posevecs"], g["pose_tfs"]):
        assert np.array_equal(mu.cvt_pose_vec2tf(v), tf)
    assert np.array_equal(mu.get_sim_cam_mat(347, 520), g["simcam_347_520"])
    gs, cs = int(g["vox_gs"]), float(g["vox_cs"])
    got = np.array([mu.base_pos2grid_id_3d(gs, cs, *p) for p in g["vox_pts"]])
    assert np.array_equal(got, g["vox_ids"])


@pytest.mark.parametrize("name", ["g2a_builder_small.npz", "g2b_builder_growth.npz"])
def test_transforms_and_pose_chain(golden, name):
    from oracle import avl_oracle as O
    g = golden(name)
    m = Map(make_cfg(g))
    assert np.array_equal(m.base2cam_tf, g["base2cam_tf"]) and np.array_equal(m.base_transform, g["base_transform"])
    assert m.gs == int(g["gs"]) and m.cs == float(g["cs"])
    b = VLMapBuilder(Path("/tmp"), make_cfg(g), None, [], [], m.base2cam_tf, m.base_transform)
    Ts = b.frame_transforms(g["poses_rt"])
    assert np.array_equal(np.stack(Ts), O.pc_transforms(g["poses_rt"], g["base_transform"], g["base2cam_tf"]))


def test_sampling_follows_global_numpy_rng(golden):
    g = golden("g2a_builder_small.npz")
    H, W = g["depths"].shape[1:]
    np.random.seed(1234)            # the seed tools/gen_golden.py gave the reference run
    for i in range(len(g["samples"])):
        assert np.array_equal(VLMapBuilder.sample_pixels(H * W, int(g["rate"])), g["samples"][i])


def test_save_load_roundtrip(tmp_path):
    rng = np.random.default_rng(0)
    n, D = 17, 8
    arrays = dict(grid_feat=rng.standard_normal((n, D)).astype(np.float32), grid_pos=rng.integers(0, 9, (n, 3)).astype(np.int32),
                  weight=rng.random(n).astype(np.float32), occupied_ids=-np.ones((9, 9, 3), np.int32),
                  grid_rgb=rng.integers(0, 255, (n, 3)).astype(np.uint8))
    p = tmp_path / "vlmap" / "vlmaps.h5df"
    p.parent.mkdir()
    assert not mu.map_file_exists(p)
    mu.save_3d_map(p, arrays["grid_feat"], arrays["grid_pos"], arrays["weight"], arrays["occupied_ids"], [2, 0, 1], arrays["grid_rgb"])
    assert mu.map_file_exists(p)
    it, gf, gp, w, occ, rgb = mu.load_3d_map(p)
    assert it == [0, 1, 2]
    for a, b in ((gf, "grid_feat"), (gp, "grid_pos"), (w, "weight"), (occ, "occupied_ids"), (rgb, "grid_rgb")):
        assert np.array_equal(a, arrays[b]) and a.dtype == arrays[b].dtype


def test_map_attribute_surface_and_obstacles():
    cfg = Cfg(map_type="vlmap", grid_size=20, cell_size=0.05,
              pose_info=Cfg(camera_height=1.5, base2cam_rot=[1, 0, 0, 0, -1, 0, 0, 0, -1], base_forward_axis=[0, 0, -1],
                            base_left_axis=[-1, 0, 0], base_up_axis=[0, 1, 0]))
    m = Map.create(cfg)
    for a in ("grid_feat", "grid_pos", "weight", "occupied_ids", "grid_rgb", "mapped_iter_list", "obstacles_map",
              "obstacles_cropped", "scores_mat", "categories"):
        assert getattr(m, a) is None
    # generate_cropped_obstacle_map is host logic (map.py:97-104); generate_obstacle_map itself runs on the GPU
    # (avl_obstacle_map) and is covered by tests/test_map2d_gpu.py against the reference's outputs
    obs = np.ones((20, 20), bool)
    obs[5, 6] = obs[9, 12] = False
    m.generate_cropped_obstacle_map(obs)
    assert (m.rmin, m.rmax, m.cmin, m.cmax) == (5, 9, 6, 12) and m.obstacles_cropped.shape == (5, 7)
    with pytest.raises(Exception, match="Categories are not preloaded"):
        m.index_map("sofa", with_init_cat=True)
    assert VLMapBuilder(Path("/tmp"), cfg, None, [], [], None, None).create_camera_map() is NotImplementedError


@pytest.fixture(params=["H5Dwrite", "H5Dwrite_chunk"])
def chunk_writes(request, monkeypatch):
    """large chunked datasets are written chunk by chunk with H5Dwrite_chunk (h5lite.H5File._write_whole_chunks); here: every one, or none"""
    from avlmaps_amd.utils import h5lite
    monkeypatch.setattr(h5lite.H5File, "DIRECT_CHUNK_BYTES", 0 if request.param == "H5Dwrite_chunk" else 1 << 62)
    return request.param


def test_map_file_is_the_reference_hdf5_layout(tmp_path, chunk_writes):
    """save_3d_map writes a real HDF5 file with the six dataset names / dtypes / shapes of the reference writer
    (mapping_utils.py:499-505) -- through h5py, or through the HDF5 C library where h5py is missing -- and load_3d_map reads
    it back as the reference reader does (:508-541); init_height_id is a 0-d int32 dataset like upstream's"""
    import shutil
    import subprocess
    backend = mu.hdf5_backend()
    if backend is None:
        pytest.skip("neither h5py nor libhdf5 on this machine")
    rng = np.random.default_rng(3)
    n, D, gs, vh = 23, 16, 12, 5
    arrays = dict(grid_feat=rng.standard_normal((n, D)).astype(np.float32), grid_pos=rng.integers(0, gs, (n, 3)).astype(np.int32),
                  weight=rng.random(n).astype(np.float32), occupied_ids=-np.ones((gs, gs, vh), np.int32),
                  grid_rgb=rng.integers(0, 255, (n, 3)).astype(np.uint8))
    p = tmp_path / "vlmaps.h5df"
    mu.save_3d_map(p, arrays["grid_feat"], arrays["grid_pos"], arrays["weight"], arrays["occupied_ids"], {4, 1, 3}, arrays["grid_rgb"],
                   init_height_id=7)
    assert p.read_bytes()[:8] == b"\x89HDF\r\n\x1a\n"
    assert not Path(str(p) + ".npz").exists()
    from avlmaps_amd.utils import h5lite
    if h5lite.available():
        with h5lite.H5File(p) as f:
            assert sorted(f.keys()) == sorted(mu.MAP_DATASETS + ("init_height_id",))
            want = dict(mapped_iter_list=((3,), np.int32), grid_feat=((n, D), np.float32), grid_pos=((n, 3), np.int32),
                        weight=((n,), np.float32), occupied_ids=((gs, gs, vh), np.int32), grid_rgb=((n, 3), np.uint8),
                        init_height_id=((), np.int32))
            for k, (shape, dt) in want.items():
                assert f.shape_dtype(k) == (shape, np.dtype(dt)), k
    out = mu.load_3d_map(p)
    assert out[0] == [1, 3, 4] and int(out[6]) == 7
    for a, b in zip(out[1:6], ("grid_feat", "grid_pos", "weight", "occupied_ids", "grid_rgb")):
        assert np.array_equal(a, arrays[b]) and a.dtype == arrays[b].dtype
    h5dump = shutil.which("h5dump") or ("/opt/conda/bin/h5dump" if Path("/opt/conda/bin/h5dump").exists() else None)
    if h5dump:   # the HDF5 project's own tool agrees on types and extents
        txt = subprocess.run([h5dump, "-H", str(p)], capture_output=True, text=True).stdout
        for frag in ('DATASET "grid_feat"', "H5T_IEEE_F32LE", f"( {n}, {D} )", 'DATASET "grid_rgb"', "H5T_STD_U8LE",
                     'DATASET "occupied_ids"', f"( {gs}, {gs}, {vh} )", "H5T_STD_I32LE"):
            assert frag in txt, frag
        # ... and on the CONTENT (h5dump -d), not only the header: every value of the small datasets, the features to the
        # precision h5dump prints
        import re

        def dumped(name):
            t = subprocess.run([h5dump, "-d", "/" + name, "-w", "0", str(p)], capture_output=True, text=True).stdout
            body = t[t.index("DATA {"):]
            vals = []
            for line in body.splitlines():
                m = re.match(r"\s*\(\d+(?:,\d+)*\):\s*(.*)", line)
                if m:
                    vals += [float(x) for x in m.group(1).replace(",", " ").split()]
            return np.array(vals)
        assert np.array_equal(dumped("mapped_iter_list"), [1, 3, 4])
        assert np.array_equal(dumped("grid_pos").reshape(n, 3), arrays["grid_pos"])
        assert np.array_equal(dumped("grid_rgb").reshape(n, 3), arrays["grid_rgb"])
        assert np.array_equal(dumped("occupied_ids"), arrays["occupied_ids"].ravel())
        np.testing.assert_allclose(dumped("weight"), arrays["weight"], rtol=1e-5)
        np.testing.assert_allclose(dumped("grid_feat").reshape(n, D), arrays["grid_feat"], rtol=1e-5, atol=1e-7)
    # the navigator reads the file with h5py (mapping_utils.py:508-541): the first box that has it proves the interop for real --
    # also for a file written through the HDF5 C library (h5lite), which is what this image's builder produces
    try:
        import h5py
    except ImportError:
        h5py = None
    if h5py is not None:
        with h5py.File(p, "r") as f:
            assert sorted(f.keys()) == sorted(mu.MAP_DATASETS + ("init_height_id",))
            assert f["mapped_iter_list"][:].tolist() == [1, 3, 4] and int(f["init_height_id"][()]) == 7
            for k in ("grid_feat", "grid_pos", "weight", "occupied_ids", "grid_rgb"):
                assert np.array_equal(f[k][:], arrays[k]) and f[k].dtype == arrays[k].dtype, k
        if h5lite.available():
            p2 = tmp_path / "via_h5lite.h5df"
            with h5lite.H5File(p2, "w") as g:
                for k, v in arrays.items():
                    g.create_dataset(k, v)
            with h5py.File(p2, "r") as f:
                for k, v in arrays.items():
                    assert np.array_equal(f[k][:], v), k


def test_h5lite_extendible_datasets_and_foreign_files(tmp_path):
    from avlmaps_amd.utils import h5lite
    if not h5lite.available():
        pytest.skip("libhdf5 not found")
    rng = np.random.default_rng(4)
    a = rng.standard_normal((9, 6)).astype(np.float32)
    occ = -np.ones((3, 4, 5), np.int32)
    with h5lite.H5File(tmp_path / "x.h5", "w") as f:
        f.create_dataset("a", data=a[:4], maxshape=(None, 6))
        f.create_dataset("occ", data=occ)
        f.create_dataset("empty", data=np.zeros((0, 3), np.uint8), maxshape=(None, 3))
    with h5lite.H5File(tmp_path / "x.h5", "r+") as f:        # a later checkpoint: append rows, overwrite a few, poke cells
        f.resize("a", 9)
        f.write_rows("a", 4, a[4:])
        f.write_scattered_rows("a", np.array([0, 2, 3, 8]), a[[0, 2, 3, 8]] + 1)
        f.write_points("occ", np.array([[0, 0, 0], [2, 3, 4]]), np.array([5, 6]))
        with pytest.raises(h5lite.H5Error):
            f.write_rows("a", 8, a[:2])                        # beyond the extent
    d = h5lite.read_datasets(tmp_path / "x.h5")
    want = a.copy()
    want[[0, 2, 3, 8]] += 1
    assert np.array_equal(d["a"], want) and d["empty"].shape == (0, 3)
    assert d["occ"][0, 0, 0] == 5 and d["occ"][2, 3, 4] == 6 and (d["occ"] == -1).sum() == occ.size - 2
    with pytest.raises(KeyError):
        h5lite.H5File(tmp_path / "x.h5").read("nope")
    with pytest.raises(h5lite.H5Error):
        h5lite.H5File(tmp_path / "missing.h5")
    # a file written by other software (MATLAB v7.3 = HDF5 with a user block), shipped with SciPy's tests
    import scipy.io
    mat = Path(scipy.io.__file__).parent / "matlab" / "tests" / "data" / "testhdf5_7.4_GLNX86.mat"
    if mat.exists():
        with h5lite.H5File(mat) as f:
            assert "testdouble" in f.keys() and f.read("testdouble").dtype == np.float64


def test_h5lite_row_runs_whole_chunks_or_hyperslabs(tmp_path, chunk_writes):
    """write_row_runs: runs made of whole chunks may go out with H5Dwrite_chunk (incl. the partial last chunk of the dataset and chunks
    that did not exist before a resize), anything else as hyperslabs -- the dataset read back is the array either way"""
    from avlmaps_amd.utils import h5lite
    if not h5lite.available():
        pytest.skip("libhdf5 not found")
    rng = np.random.default_rng(12)
    a = rng.standard_normal((203, 5)).astype(np.float32)
    p = tmp_path / "runs.h5"
    with h5lite.H5File(p, "w") as f:
        f.create_dataset("g", data=a[:100], maxshape=(None, 5), chunks=(8, 5))       # 12 full chunks + a partial one
        f.create_dataset("v", data=a[:100, 0].copy(), maxshape=(None,))              # 1-d: never chunk by chunk
    b = a.copy()
    b[:100] += 1
    with h5lite.H5File(p, "r+") as f:
        f.resize("g", 203)
        dset = f._open("g")
        try:
            assert f._chunk_rows(dset, (203, 5)) == 8
        finally:
            h5lite._lib().H5Dclose(dset)
        f.write_row_runs("g", [(0, 8), (16, 40), (96, 203)], b)                       # whole chunks; the last run ends in the partial chunk 200..202
        f.resize("v", 203)
        f.write_row_runs("v", [(0, 203)], b[:, 0].copy())
    want = a.copy()
    want[0:8], want[16:40], want[96:203] = b[0:8], b[16:40], b[96:203]
    with h5lite.H5File(p) as f:
        assert np.array_equal(f.read("g"), want) and np.array_equal(f.read("v"), b[:, 0])
    with h5lite.H5File(p, "r+") as f:
        f.write_row_runs("g", [(3, 11), (50, 51)], b)                                 # not chunk-aligned: hyperslabs
        with pytest.raises(h5lite.H5Error):
            f.write_row_runs("g", [(200, 204)], b)
    want[3:11], want[50:51] = b[3:11], b[50:51]
    with h5lite.H5File(p) as f:
        assert np.array_equal(f.read("g"), want)


def test_map_file_writer_incremental_checkpoints(tmp_path, chunk_writes):
    """MapFileWriter: after the first full save only dirty + new rows (and the new cells of occupied_ids) are written; the
    file read back is always the complete current map in the reference's layout"""
    from avlmaps_amd.utils import h5lite
    if not h5lite.available():
        pytest.skip("libhdf5 not found")
    rng = np.random.default_rng(9)
    gs, vh, D = 16, 6, 8
    cells = rng.permutation(gs * gs * vh)[:300]
    pos_all = np.stack([cells // (gs * vh), (cells // vh) % gs, cells % vh], 1).astype(np.int32)

    def state(n, version):
        r = np.random.default_rng(version)
        feat = r.standard_normal((n, D)).astype(np.float32)
        occ = -np.ones((gs, gs, vh), np.int32)
        occ[pos_all[:n, 0], pos_all[:n, 1], pos_all[:n, 2]] = np.arange(n, dtype=np.int32)
        return dict(grid_feat=feat, grid_pos=pos_all[:n].copy(), weight=r.random(n).astype(np.float32),
                    grid_rgb=r.integers(0, 255, (n, 3)).astype(np.uint8), occupied_ids=occ)

    w = mu.MapFileWriter(tmp_path / "vlmaps.h5df")
    w.FEAT_CHUNK_ROWS = 1                                   # row granularity first; chunk-aligned runs further down
    cur = state(100, 0)
    w.save(cur, {0, 1}, None)
    for step, n in enumerate((140, 140, 300), start=1):
        new = state(n, step)
        dirty = np.zeros(n, np.uint8)
        keep = np.random.default_rng(100 + step).random(len(cur["grid_pos"])) < 0.7      # 70 % of the old rows did not change
        for k in w.ROW_SETS:
            new[k][: len(keep)][keep] = cur[k][keep]
        dirty[: len(keep)][~keep] = 1
        dirty[len(keep):] = 1
        w.save(new, set(range(2 * step + 2)), dirty)
        it, gf, gp, wt, occ, rgb = mu.load_3d_map(tmp_path / "vlmaps.h5df")
        assert it == list(range(2 * step + 2))
        for a, k in ((gf, "grid_feat"), (gp, "grid_pos"), (wt, "weight"), (occ, "occupied_ids"), (rgb, "grid_rgb")):
            assert np.array_equal(a, new[k]) and a.dtype == new[k].dtype, (step, k)
        cur = new
    assert [s["mode"] for s in w.stats] == ["full", "incremental", "incremental", "incremental"]
    assert w.stats[1]["rows_written"] < 0.6 * w.stats[1]["rows_total"] and w.stats[2]["rows_written"] < 0.5 * 140
    # the lean form (VoxelAccumulator.finalize_rows): only the changed rows reach the writer; it patches its host mirror of the
    # map and writes the file from there in coalesced runs
    new = state(300, 7)
    keep = np.random.default_rng(77).random(300) < 0.8
    for k in w.ROW_SETS:
        new[k][keep] = cur[k][keep]
    idx = np.flatnonzero(~keep)
    w.MAX_RUNS, saved_max = 5, w.MAX_RUNS                                  # forces coalescing across small gaps
    w.save_packed(dict(n=300, n_saved=300, idx=idx, rows={k: new[k][idx] for k in w.ROW_SETS}), set(range(9)))
    w.MAX_RUNS = saved_max
    it, gf, gp, wt, occ, rgb = mu.load_3d_map(tmp_path / "vlmaps.h5df")
    assert it == list(range(9)) and w.stats[-1]["mode"] == "incremental" and w.stats[-1]["runs"] <= 5
    assert w.stats[-1]["rows_dirty"] == len(idx) <= w.stats[-1]["rows_written"] < 300
    for a, k in ((gf, "grid_feat"), (gp, "grid_pos"), (wt, "weight"), (occ, "occupied_ids"), (rgb, "grid_rgb")):
        assert np.array_equal(a, new[k]) and a.dtype == new[k].dtype, k
    # ... and with appended rows: the mirror grows, the new cells of occupied_ids come from the new grid_pos rows
    small = {k: v[:280].copy() for k, v in new.items() if k != "occupied_ids"}
    w.save(dict(small, occupied_ids=state(280, 0)["occupied_ids"]), {0}, None)                     # full rewrite at 280 rows
    idx = np.array([3, 4, 100] + list(range(280, 300)))
    w.save_packed(dict(n=300, n_saved=280, idx=idx, rows={k: new[k][idx] for k in w.ROW_SETS}), {0, 1})
    it, gf, gp, wt, occ, rgb = mu.load_3d_map(tmp_path / "vlmaps.h5df")
    for a, k in ((gf, "grid_feat"), (gp, "grid_pos"), (wt, "weight"), (occ, "occupied_ids"), (rgb, "grid_rgb")):
        assert np.array_equal(a, new[k]), k
    cm = w.current_map()
    for k in ("grid_feat", "grid_pos", "weight", "grid_rgb", "occupied_ids"):
        assert np.array_equal(cm[k], new[k]), k
    with pytest.raises(RuntimeError):
        w.save_packed(dict(n=300, n_saved=280, idx=idx, rows={k: new[k][idx] for k in w.ROW_SETS}), {0})   # stale n_saved
    # whole-chunk writes: with 8-row chunks a save only writes runs of complete chunks (the last one may be partial) that contain
    # a changed or a new row -- and the file is still the complete current map
    w8 = mu.MapFileWriter(tmp_path / "chunked.h5df")
    w8.FEAT_CHUNK_ROWS = 8
    base = state(203, 20)
    w8.save(base, {0}, None)
    nxt = state(230, 21)
    for k in w8.ROW_SETS:
        nxt[k][:203] = base[k]
    changed = np.array([5, 6, 90, 201])
    for k in w8.ROW_SETS:
        nxt[k][changed] = state(230, 22)[k][changed]
    w8.save_packed(dict(n=230, n_saved=203, idx=np.concatenate([changed, np.arange(203, 230)]),
                        rows={k: nxt[k][np.concatenate([changed, np.arange(203, 230)])] for k in w8.ROW_SETS}), {0, 1})
    st = w8.stats[-1]
    assert st["rows_written"] == 8 + 8 + (230 - 200) and st["runs"] == 3               # chunks 0, 11 and 25..28 (rows 200..229)
    got = mu.load_3d_map(tmp_path / "chunked.h5df")
    for a, k in zip(got[1:], ("grid_feat", "grid_pos", "weight", "occupied_ids", "grid_rgb")):
        assert np.array_equal(a, nxt[k]), k
    # a map that shrank (or no dirty information) falls back to a full rewrite
    w.save(state(50, 9), {0}, None)
    assert w.stats[-1]["mode"] == "full" and len(mu.load_3d_map(tmp_path / "vlmaps.h5df")[2]) == 50


def test_skip_pixel_shuffles_fast_forwards_the_global_rng_like_the_shuffles():
    """rank r of a sharded build advances np.random past the frames before its shard with the draws of the shuffles only (host C
    code in the library, no GPU): the state afterwards is the state after really shuffling -- every array length incl. the
    power-of-two boundaries of the masked-rejection rule, from a mid-block generator position"""
    from avlmaps_amd.map.vlmap_builder import VLMapBuilder
    for n_pix, k in ((0, 2), (1, 3), (2, 5), (3, 5), (4, 4), (5, 3), (8, 4), (9, 4), (1000, 3), (4096, 2), (4097, 2), (50000, 2)):
        np.random.seed(123)
        np.random.rand(5)
        a = np.arange(n_pix)
        for _ in range(k):
            np.random.shuffle(a)
        want = np.random.get_state()
        np.random.seed(123)
        np.random.rand(5)
        VLMapBuilder.skip_pixel_shuffles(k, n_pix)
        got = np.random.get_state()
        assert np.array_equal(want[1], got[1]) and want[2] == got[2], (n_pix, k)
        assert np.array_equal(VLMapBuilder.sample_pixels(640, 7), (lambda m: (np.random.set_state(want), np.random.shuffle(m), m[::7])[2])(np.arange(640)))


def test_sample_pixels_is_numpys_shuffle():
    """VLMapBuilder.sample_pixels draws shuffle_mask[::rate] through the library's host C code (int32 indices, branch-free
    rejection): the samples and the state the global RNG is left in are exactly NumPy's, for every array-length class"""
    from avlmaps_amd.map.vlmap_builder import VLMapBuilder
    for n_pix, rate, k in ((1, 1, 2), (2, 1, 3), (3, 2, 3), (5, 1, 3), (8, 3, 2), (9, 2, 2), (1000, 7, 3), (4096, 5, 2), (4097, 100, 2), (70000, 100, 2)):
        np.random.seed(7)
        np.random.rand(3)
        want = []
        for _ in range(k):
            m = np.arange(n_pix)
            np.random.shuffle(m)
            want.append(m[::rate].astype(np.int32))
        ws = np.random.get_state()
        np.random.seed(7)
        np.random.rand(3)
        got = [VLMapBuilder.sample_pixels(n_pix, rate) for _ in range(k)]
        gs = np.random.get_state()
        assert all(np.array_equal(a, b) and b.dtype == np.int32 for a, b in zip(want, got)), (n_pix, rate)
        assert np.array_equal(ws[1], gs[1]) and ws[2] == gs[2], (n_pix, rate)
    assert 0.0 <= np.random.rand() < 1.0                                            # the global RNG still works afterwards


def test_every_simd_form_of_the_shuffle_draws_numpys_numbers():
    """avl_mt19937_shuffle_sample / avl_mt19937_skip_shuffles dispatch on the host's instruction set (AVX-512, AVX2, scalar); each
    form, forced through AVL_NO_AVX512 / AVL_NO_AVX2 in a fresh process, must produce NumPy's samples and leave NumPy's state --
    full 720 x 1080 frames included (every constant-mask run, block edges at the 624-word refills, ambiguous blocks)"""
    import hashlib
    import subprocess
    import sys
    cases = ((777600, 100, 3), (70001, 7, 2), (1 << 16, 3, 2), ((1 << 16) + 1, 1, 2), (1000, 1, 5), (65, 2, 4), (17, 1, 4))
    h = hashlib.sha256()
    np.random.seed(3)
    for n_pix, rate, k in cases:
        for _ in range(k):
            m = np.arange(n_pix)
            np.random.shuffle(m)
            h.update(np.ascontiguousarray(m[::rate], dtype=np.int32).tobytes())
        for _ in range(2):                                     # two more shuffles, skipped by the draws-only entry
            np.random.shuffle(np.arange(n_pix))
        st = np.random.get_state()
        h.update(st[1].tobytes() + bytes([st[2] & 0xff, st[2] >> 8]))
    want = h.hexdigest()
    prog = (
        "import hashlib, sys, numpy as np\n"
        f"sys.path.insert(0, {str(Path(__file__).resolve().parent.parent)!r})\n"
        "from avlmaps_amd.map.vlmap_builder import VLMapBuilder\n"
        "h = hashlib.sha256(); np.random.seed(3)\n"
        f"for n_pix, rate, k in {cases!r}:\n"
        "    for _ in range(k):\n"
        "        h.update(VLMapBuilder.sample_pixels(n_pix, rate).tobytes())\n"
        "    VLMapBuilder.skip_pixel_shuffles(2, n_pix)\n"
        "    st = np.random.get_state()\n"
        "    h.update(st[1].tobytes() + bytes([st[2] & 0xff, st[2] >> 8]))\n"
        "print(h.hexdigest())\n")
    import os
    for form, env in (("widest", {}), ("avx2", {"AVL_NO_AVX512": "1"}), ("scalar", {"AVL_NO_AVX512": "1", "AVL_NO_AVX2": "1"})):
        r = subprocess.run([sys.executable, "-c", prog], env={**os.environ, **env}, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        assert r.stdout.strip().splitlines()[-1] == want, form


def test_lseg_window_plan_reproduces_the_reference_windows(golden):
    """WindowPlan + window_batch (the geometry of lseg_utils.py:36-96 as one canvas and one batch of views) with the reference's
    accumulation done in torch on the CPU: the reference run's feature maps (g5) bit for bit -- so what is left for the GPU
    kernel is the summation itself"""
    import sys
    import torch
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tools"))
    from gen_golden import FakeLSeg
    from avlmaps_amd.utils.lseg_utils import WindowPlan, default_transform, window_batch
    g = golden("g5_lseg_protocol.npz")
    for name, n_win in (("pad_short", 2), ("grid_2x3", 6), ("tall", 6)):
        crop, base = (int(x) for x in g[f"{name}_cfg"])
        img = default_transform(g[f"{name}_img"]).unsqueeze(0)
        plan = WindowPlan.make(img.shape[2], img.shape[3], crop, base)
        assert len(plan.origins) == n_win and (plan.height, plan.width) == g[f"{name}_feat"].shape[2:]
        batch = window_batch(img, plan, (0.5,) * 3, (0.5,) * 3)
        assert tuple(batch.shape) == (n_win, 3, crop, crop)
        out, _ = FakeLSeg()(batch, ["example"])
        acc = torch.zeros((out.shape[1],) + plan.canvas)
        cnt = torch.zeros((1,) + plan.canvas)
        for (h0, w0), o in zip(plan.origins, out):
            acc[:, h0:h0 + crop, w0:w0 + crop] += o
            cnt[:, h0:h0 + crop, w0:w0 + crop] += 1
        assert np.array_equal((acc / cnt)[:, :plan.height, :plan.width].numpy(), g[f"{name}_feat"][0])
    # base_size <= crop_size: one window holds the resized image (upstream raises UnboundLocalError on that branch)
    p1 = WindowPlan.make(720, 1080, 480, 400)
    assert p1.origins == [(0, 0)] and p1.canvas == (480, 480) and (p1.height, p1.width) == (267, 400)
    # the reference's default: 720 x 1080 -> 347 x 520, crops of 480, stride 320 -> 1 x 2 windows
    p2 = WindowPlan.make(720, 1080, 480, 520)
    assert (p2.height, p2.width) == (347, 520) and p2.origins == [(0, 0), (0, 320)] and p2.canvas == (480, 800)


def test_category_matcher_hook():
    """find_similar_category_id resolves what it can locally and hands the rest to an injected matcher (upstream: an LLM call,
    index_utils.py:8-32); without one an unmatched name is a KeyError, a matcher's out-of-range answer a ValueError"""
    from avlmaps_amd.utils import index_utils as iu
    cats = ["void", "Sofa", "dining table", "chair"]
    assert iu.find_similar_category_id("chair", cats) == 3 and iu.find_similar_category_id("sofa", cats) == 1
    assert iu.find_similar_category_id("table", cats) == 2
    with pytest.raises(KeyError):
        iu.find_similar_category_id("couch", cats)
    seen = []
    assert iu.find_similar_category_id("couch", cats, matcher=lambda name, cl: seen.append((name, tuple(cl))) or 1) == 1
    assert seen == [("couch", tuple(cats))]
    iu.set_category_matcher(lambda name, cl: cl.index("Sofa"))
    try:
        assert iu.find_similar_category_id("somewhere to sit down", cats) == 1
        assert iu.find_similar_category_id("chair", cats) == 3          # local rules still come first
    finally:
        iu.set_category_matcher(None)
    with pytest.raises(ValueError):
        iu.find_similar_category_id("couch", cats, matcher=lambda name, cl: 9)


def test_map_saves_are_atomic_and_interrupted_patches_are_detectable(tmp_path):
    """MapFileWriter: a full save is written next to the target and renamed over it; an in-place incremental patch clears a
    marker dataset first and sets it last, so that a file whose patch was interrupted says so (map_checkpoint_complete) -- a
    resumed build then keeps its voxels but does not trust its frame list (VLMapBuilder._resume)"""
    from avlmaps_amd.utils import h5lite
    if not h5lite.available():
        pytest.skip("libhdf5 not found")
    rng = np.random.default_rng(4)
    n, D, gs, vh = 200, 8, 16, 4
    pos = np.stack(np.unravel_index(rng.choice(gs * gs * vh, n, replace=False), (gs, gs, vh)), 1).astype(np.int32)
    occ = -np.ones((gs, gs, vh), np.int32)
    occ[pos[:, 0], pos[:, 1], pos[:, 2]] = np.arange(n)
    arrays = dict(grid_feat=rng.standard_normal((n, D)).astype(np.float32), grid_pos=pos, weight=rng.random(n).astype(np.float32),
                  occupied_ids=occ, grid_rgb=rng.integers(0, 255, (n, 3)).astype(np.uint8))
    p = tmp_path / "vlmaps.h5df"
    w = mu.MapFileWriter(p)
    w.save({k: v.copy() for k, v in arrays.items()}, [0, 1])
    assert p.exists() and not (tmp_path / "vlmaps.h5df.tmp").exists() and mu.map_checkpoint_complete(p)
    assert mu.read_map_dataset(p, "mapped_iter_list").tolist() == [0, 1] and mu.read_map_dataset(p, "nope") is None
    assert mu.load_3d_map(p)[0] == [0, 1]                       # the extra marker dataset does not disturb the reference's reader
    dirty = np.zeros(n, np.uint8)
    dirty[[3, 70]] = 1
    arrays["grid_feat"][[3, 70]] += 1
    w.save({k: v.copy() for k, v in arrays.items()}, [0, 1, 2], dirty)
    assert w.stats[-1]["mode"] == "incremental" and mu.map_checkpoint_complete(p)
    assert np.array_equal(mu.load_3d_map(p)[1], arrays["grid_feat"]) and mu.load_3d_map(p)[0] == [0, 1, 2]
    # a patch that dies half way: marker cleared, rows written, list not yet
    with h5lite.H5File(p, "r+") as f:
        f.write_rows(mu.MapFileWriter.MARKER, 0, np.zeros(1, np.int32))
    assert not mu.map_checkpoint_complete(p)
    # files written without the marker (upstream's writer, save_3d_map) count as complete
    q = tmp_path / "plain.h5df"
    mu.save_3d_map(q, arrays["grid_feat"], pos, arrays["weight"], occ, [5], arrays["grid_rgb"])
    assert mu.map_checkpoint_complete(q)
