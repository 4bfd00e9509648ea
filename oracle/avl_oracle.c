/*
 * avl_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the AVLMaps map-build / landmark-index hot path, used only by
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker.  The product
 * (avlmaps_amd/) never imports, links or executes anything in this directory.
 *
 * Parity is PINNED: tests/test_oracle_golden.py checks every function here against the golden
 * vectors in tests/golden/ (npz files), which tools/gen_golden.py produced by executing the upstream
 * reference itself (the reference ships no tests of its own: SURVEY.md section 4).
 *
 * Each function cites the reference lines (relative to the upstream repo root) it restates.
 * Floating-point notes that make the restatement bit-faithful:
 *   - NumPy float64 matmul with >=3 columns goes through OpenBLAS dgemm whose micro-kernels
 *     accumulate k = 0..K-1 sequentially with FMA:  acc = fma(a_k, b_k, acc), acc0 = 0.
 *   - NumPy (3,3)@(3,1) goes through OpenBLAS dgemv whose 3-term tail evaluates
 *     fma(a2,x2, fma(a0,x0, a1*x1))  (measured against the reference; identical to the dgemm
 *     order for pinhole intrinsics whose off-diagonal terms are zero).
 *   - Python int() truncates toward zero; all divisions are true IEEE float64 divisions.
 *   - NumPy >= 2 (NEP 50) type promotion decides which intermediates are float32 / float64
 *     (see avlo_integrate_frame).
 * Compile with -ffp-contract=off so the only fused operations are the explicit fma() calls.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define AVLO_API __attribute__((visibility("default")))

/* ---- avlmaps/utils/mapping_utils.py:345-349 base_pos2grid_id_3d -------------------------------- */
static inline long long py_int(double v) { return (long long)trunc(v); }

AVLO_API void avlo_base_pos2grid_id_3d(int gs, double cs, double x, double y, double z, long long out[3]) {
    out[0] = py_int((double)gs / 2.0 - (double)py_int(x / cs));
    out[1] = py_int((double)gs / 2.0 - (double)py_int(y / cs));
    out[2] = py_int(z / cs);
}

/* ---- avlmaps/utils/mapping_utils.py:599-605 project_point (cam_mat @ p via dgemv) ------------- */
static inline double gemv3(const double* a, const double* x) {
    return fma(a[2], x[2], fma(a[0], x[0], a[1] * x[1]));
}

AVLO_API void avlo_project_point(const double K[9], const double p[3], long long* px, long long* py, double* pz) {
    double q0 = gemv3(K + 0, p), q1 = gemv3(K + 3, p), q2 = gemv3(K + 6, p);
    *pz = q2;
    *px = py_int(q0 / q2 - 0.5);
    *py = py_int(q1 / q2 - 0.5);
}

/* ---- avlmaps/utils/mapping_utils.py:226-251 depth2pc, for ONE flattened pixel index ------------
 * p_2d = (u+0.5, v+0.5, 1); pc = Kinv @ p_2d (dgemm, FMA chain); pc = pc * z; mask = min < pc_z < max */
AVLO_API int avlo_depth2pc_pixel(const float* depth, int W, const double Kinv[9], int pix,
                                 double min_depth, double max_depth, double pc[3]) {
    double x = (double)(pix % W) + 0.5, y = (double)(pix / W) + 0.5, z = (double)depth[pix];
    for (int i = 0; i < 3; ++i) {
        double acc = Kinv[3 * i + 0] * x;            /* fma(a,b,0) == round(a*b) */
        acc = fma(Kinv[3 * i + 1], y, acc);
        acc = fma(Kinv[3 * i + 2], 1.0, acc);
        pc[i] = acc * z;
    }
    return (pc[2] > min_depth) && (pc[2] < max_depth);
}

/* ---- avlmaps/utils/mapping_utils.py:305-315 transform_pc (pose @ [pc;1], dgemm FMA chain) ------- */
AVLO_API void avlo_transform_point(const double T[16], const double p[3], double out[3]) {
    for (int i = 0; i < 3; ++i) {
        double acc = T[4 * i + 0] * p[0];
        acc = fma(T[4 * i + 1], p[1], acc);
        acc = fma(T[4 * i + 2], p[2], acc);
        acc = fma(T[4 * i + 3], 1.0, acc);
        out[i] = acc;
    }
}

/* =================================================================================================
 * Sequential map state: avlmaps/map/vlmap_builder.py:195-224 (_init_map), :286-311 (_reserve_map_space)
 * ================================================================================================= */
typedef struct avlo_map {
    int n0;               /* rows; == gs for the square mobile-base map */
    int gs, vh, D;        /* gs = columns, vh = heights */
    double cs;
    long long cap, max_id;
    float* grid_feat;     /* (cap, D) float32 */
    int32_t* grid_pos;    /* (cap, 3) */
    double* weight;       /* holds float32-rounded values until the first growth, float64 after */
    double* grid_rgb;     /* holds uint8 values until the first growth, float32 values after   */
    int32_t* occupied;    /* (gs, gs, vh), -1 = empty */
    int grown;            /* number of capacity doublings so far */
    long long n_points;   /* points that updated a voxel (statistics) */
} avlo_map;

/* rectangular grid n0 x n1 x n2: vlmap_builder_multi_floor.py:222-228 (occupied_ids = grid_size[[0, 2, 1]], capacity n0*n1) */
AVLO_API avlo_map* avlo_map_create_grid(int n0, int gs, int vh, double cs, int D) {
    avlo_map* m = (avlo_map*)calloc(1, sizeof(avlo_map));
    m->n0 = n0; m->gs = gs; m->cs = cs; m->vh = vh; m->D = D;
    m->cap = (long long)n0 * gs;                      /* vlmap_builder.py:202 / vlmap_builder_multi_floor.py:225 */
    m->grid_feat = (float*)calloc((size_t)m->cap * D, sizeof(float));
    m->grid_pos = (int32_t*)calloc((size_t)m->cap * 3, sizeof(int32_t));
    m->weight = (double*)calloc((size_t)m->cap, sizeof(double));
    m->grid_rgb = (double*)calloc((size_t)m->cap * 3, sizeof(double));
    size_t ncell = (size_t)n0 * gs * vh;
    m->occupied = (int32_t*)malloc(ncell * sizeof(int32_t));
    for (size_t i = 0; i < ncell; ++i) m->occupied[i] = -1;
    return m;
}

AVLO_API avlo_map* avlo_map_create(int gs, double cs, int vh, int D) { return avlo_map_create_grid(gs, gs, vh, cs, D); }

AVLO_API void avlo_map_destroy(avlo_map* m) {
    if (!m) return;
    free(m->grid_feat); free(m->grid_pos); free(m->weight); free(m->grid_rgb); free(m->occupied); free(m);
}

static void avlo_grow(avlo_map* m) {                  /* vlmap_builder.py:286-311 */
    long long nc = m->cap * 2;
    m->grid_feat = (float*)realloc(m->grid_feat, (size_t)nc * m->D * sizeof(float));
    memset(m->grid_feat + (size_t)m->cap * m->D, 0, (size_t)m->cap * m->D * sizeof(float));
    m->grid_pos = (int32_t*)realloc(m->grid_pos, (size_t)nc * 3 * sizeof(int32_t));
    memset(m->grid_pos + (size_t)m->cap * 3, 0, (size_t)m->cap * 3 * sizeof(int32_t));
    m->weight = (double*)realloc(m->weight, (size_t)nc * sizeof(double));
    memset(m->weight + m->cap, 0, (size_t)m->cap * sizeof(double));
    m->grid_rgb = (double*)realloc(m->grid_rgb, (size_t)nc * 3 * sizeof(double));
    memset(m->grid_rgb + (size_t)m->cap * 3, 0, (size_t)m->cap * 3 * sizeof(double));
    m->cap = nc;
    m->grown += 1;   /* weight: f32 ++ int32 zeros -> float64;  grid_rgb: uint8 ++ f32 zeros -> float32 */
}

/* vlmap_builder.py:141-178 == vlmap_builder_multi_floor.py:148-194: everything after the voxel index is known.
 * returns 1 if a voxel was updated, 0 if the sample was skipped, -1 where the reference would raise IndexError */
static int avlo_update_voxel(avlo_map* m, long long row, long long col, long long h, const double pl[3], const double K[9],
                             const double Kf[9], int H, int W, const float* feat, int Hf, int Wf, const uint8_t* rgb) {
    const int D = m->D, gs = m->gs, vh = m->vh;
    long long px, py; double pz;
    avlo_project_point(K, pl, &px, &py, &pz);                                        /* :141 */
    if (px < 0) px += W;                  /* numpy negative-index wrap of rgb[py, px, :] (:142) */
    if (py < 0) py += H;
    if (px < 0 || px >= W || py < 0 || py >= H) return -1;
    const uint8_t* rgb_v = rgb + ((size_t)py * W + px) * 3;
    avlo_project_point(Kf, pl, &px, &py, &pz);                                       /* :143 */

    if (m->max_id >= m->cap) avlo_grow(m);                                           /* :151-152 */

    double radial = (pl[0] * pl[0] + pl[1] * pl[1]) + pl[2] * pl[2];                 /* :156 */
    double alpha = exp(-radial / (2 * 0.6));                                         /* :157-158 */

    if (px < 0 || py < 0 || px >= Wf || py >= Hf) return 0;                          /* :161 */
    const float* f = feat + (size_t)py * Wf + px;          /* pix_feats[0, :, py, px], stride Hf*Wf */
    const size_t fs = (size_t)Hf * Wf;
    int32_t* cell = &m->occupied[((size_t)row * gs + col) * vh + h];
    if (*cell == -1) {                                                               /* :164-170 */
        long long id_new = m->max_id;
        *cell = (int32_t)id_new;
        float* gf = m->grid_feat + (size_t)id_new * D;
        for (int d = 0; d < D; ++d) gf[d] = (float)((double)f[d * fs] * alpha);      /* f32*f64 -> f64 -> f32 */
        for (int c = 0; c < 3; ++c) m->grid_rgb[id_new * 3 + c] = (double)rgb_v[c];
        double w = m->weight[id_new] + alpha;
        m->weight[id_new] = m->grown ? w : (double)(float)w;
        m->grid_pos[id_new * 3 + 0] = (int32_t)row;
        m->grid_pos[id_new * 3 + 1] = (int32_t)col;
        m->grid_pos[id_new * 3 + 2] = (int32_t)h;
        m->max_id++;
    } else {                                                                         /* :171-178 */
        long long oid = *cell;
        float* gf = m->grid_feat + (size_t)oid * D;
        double w = m->weight[oid];
        double denom = w + alpha;
        if (!m->grown) {
            /* weight[oid] is np.float32: grid_feat*w is a float32 product, grid_rgb(u8)*w is float32 */
            float wf = (float)w;
            for (int d = 0; d < D; ++d)
                gf[d] = (float)(((double)(gf[d] * wf) + (double)f[d * fs] * alpha) / denom);
            for (int c = 0; c < 3; ++c) {
                float prod = (float)m->grid_rgb[oid * 3 + c] * wf;
                double v = ((double)prod + (double)rgb_v[c] * alpha) / denom;
                m->grid_rgb[oid * 3 + c] = (double)(uint8_t)v;                       /* store into uint8 array */
            }
            m->weight[oid] = (double)(float)denom;
        } else {
            /* after _reserve_map_space: weight float64, grid_rgb float32 */
            for (int d = 0; d < D; ++d)
                gf[d] = (float)(((double)gf[d] * w + (double)f[d * fs] * alpha) / denom);
            for (int c = 0; c < 3; ++c) {
                double v = (m->grid_rgb[oid * 3 + c] * w + (double)rgb_v[c] * alpha) / denom;
                m->grid_rgb[oid * 3 + c] = (double)(float)v;
            }
            m->weight[oid] = denom;
        }
    }
    return 1;
}

/* ---- avlmaps/map/vlmap_builder.py:129-178: one frame of create_mobile_base_map -----------------
 * depth (H,W) f32; Kinv = inv(calib) f64; K = calib; Kf = get_sim_cam_mat(Hf,Wf); T = pc_transform
 * (vlmap_builder.py:133, computed on the host in float64); sample_idx = shuffle_mask[::rate]
 * (vlmap_builder.py:275-277) in the reference's order; feat is the reference layout (1,D,Hf,Wf) CHW;
 * rgb (H,W,3) u8.  Returns the number of points that updated a voxel, or -1 on an index the
 * reference would have raised IndexError for. */
AVLO_API long long avlo_integrate_frame(avlo_map* m, const float* depth, int H, int W, const double Kinv[9],
                                        const double K[9], const double Kf[9], const double T[16],
                                        const int32_t* sample_idx, int P, const float* feat, int Hf, int Wf,
                                        const uint8_t* rgb, double min_depth, double max_depth) {
    const int gs = m->gs, vh = m->vh;
    long long used = 0;
    for (int s = 0; s < P; ++s) {
        double pl[3], pg[3];
        if (!avlo_depth2pc_pixel(depth, W, Kinv, sample_idx[s], min_depth, max_depth, pl)) continue;
        avlo_transform_point(T, pl, pg);
        long long id[3];
        avlo_base_pos2grid_id_3d(gs, m->cs, pg[0], pg[1], pg[2], id);
        long long row = id[0], col = id[1], h = id[2];
        if (col >= gs || row >= gs || h >= vh || col < 0 || row < 0 || h < 0) continue;   /* :283-284 */

        int u = avlo_update_voxel(m, row, col, h, pl, K, Kf, H, W, feat, Hf, Wf, rgb);
        if (u < 0) return -1;
        used += u;
    }
    m->n_points += used;
    return used;
}

/* ---- avlmaps/map/vlmap_builder_multi_floor.py:137-199: one frame of create_global_map ------------------------
 * depth_m (H,W) float64 metres (= uint16 png / 1000.0, :138); T = camera_pose_tf @ habitat2cam_rot_tf (:141);
 * voxel index row, height, col = np.round((p - pcd_min) / cs).astype(int) (:146).  A sample outside the grid is
 * skipped (the reference only tests the upper row/col bounds and otherwise wraps around or raises). */
AVLO_API int avlo_depth2pc_pixel_f64(const double* depth, int W, const double Kinv[9], int pix, double min_depth,
                                     double max_depth, double pc[3]) {
    double x = (double)(pix % W) + 0.5, y = (double)(pix / W) + 0.5, z = depth[pix];
    for (int i = 0; i < 3; ++i) {
        double acc = Kinv[3 * i + 0] * x;
        acc = fma(Kinv[3 * i + 1], y, acc);
        acc = fma(Kinv[3 * i + 2], 1.0, acc);
        pc[i] = acc * z;
    }
    return (pc[2] > min_depth) && (pc[2] < max_depth);
}

AVLO_API long long avlo_integrate_frame_global(avlo_map* m, const double* depth_m, int H, int W, const double Kinv[9],
                                               const double K[9], const double Kf[9], const double T[16],
                                               const int32_t* sample_idx, int P, const float* feat, int Hf, int Wf,
                                               const uint8_t* rgb, double min_depth, double max_depth, const double pcd_min[3]) {
    long long used = 0;
    for (int s = 0; s < P; ++s) {
        double pl[3], pg[3];
        if (!avlo_depth2pc_pixel_f64(depth_m, W, Kinv, sample_idx[s], min_depth, max_depth, pl)) continue;
        avlo_transform_point(T, pl, pg);
        long long row = (long long)rint((pg[0] - pcd_min[0]) / m->cs);      /* np.round: half to even */
        long long h = (long long)rint((pg[1] - pcd_min[1]) / m->cs);
        long long col = (long long)rint((pg[2] - pcd_min[2]) / m->cs);
        if (col >= m->gs || row >= m->n0 || h >= m->vh || col < 0 || row < 0 || h < 0) continue;
        int u = avlo_update_voxel(m, row, col, h, pl, K, Kf, H, W, feat, Hf, Wf, rgb);
        if (u < 0) return -1;
        used += u;
    }
    m->n_points += used;
    return used;
}

/* pass 1 (vlmap_builder_multi_floor.py:97-118): fold one frame's transformed sampled points into minmax[6] */
AVLO_API void avlo_points_bbox(const double* depth_m, int W, const double Kinv[9], const double T[16], const int32_t* sample_idx,
                               int P, double min_depth, double max_depth, double minmax[6]) {
    for (int s = 0; s < P; ++s) {
        double pl[3], pg[3];
        if (!avlo_depth2pc_pixel_f64(depth_m, W, Kinv, sample_idx[s], min_depth, max_depth, pl)) continue;
        avlo_transform_point(T, pl, pg);
        for (int c = 0; c < 3; ++c) {
            if (pg[c] < minmax[c]) minmax[c] = pg[c];
            if (pg[c] > minmax[3 + c]) minmax[3 + c] = pg[c];
        }
    }
}

AVLO_API long long avlo_map_size(const avlo_map* m) { return m->max_id; }
AVLO_API int avlo_map_grown(const avlo_map* m) { return m->grown; }

/* copy out [:max_id] slices (vlmap_builder.py:313-327); weight/rgb as float64 holding the exact values */
AVLO_API void avlo_map_export(const avlo_map* m, float* grid_feat, int32_t* grid_pos, double* weight,
                              double* grid_rgb, int32_t* occupied) {
    size_t n = (size_t)m->max_id;
    if (grid_feat) memcpy(grid_feat, m->grid_feat, n * m->D * sizeof(float));
    if (grid_pos) memcpy(grid_pos, m->grid_pos, n * 3 * sizeof(int32_t));
    if (weight) memcpy(weight, m->weight, n * sizeof(double));
    if (grid_rgb) memcpy(grid_rgb, m->grid_rgb, n * 3 * sizeof(double));
    if (occupied) memcpy(occupied, m->occupied, (size_t)m->n0 * m->gs * m->vh * sizeof(int32_t));
}

/* ---- avlmaps/utils/clip_utils.py:227-229 + avlmaps/map/vlmap.py:123-124 -------------------------
 * scores = map_feats @ text_feats.T (float32), argmax(axis=1) with first-max-wins.  Scalar port used
 * for the single-core cpu_baseline and as an order-independent check (float64 accumulation). */
AVLO_API void avlo_sim_scores(const float* feat, long long N, int D, const float* q, int Q,
                              float* scores, int32_t* argmax) {
    for (long long n = 0; n < N; ++n) {
        const float* a = feat + (size_t)n * D;
        int best = 0; float bestv = 0.f;
        for (int j = 0; j < Q; ++j) {
            const float* b = q + (size_t)j * D;
            double acc = 0.0;
            for (int d = 0; d < D; ++d) acc += (double)a[d] * (double)b[d];
            float v = (float)acc;
            if (scores) scores[(size_t)n * Q + j] = v;
            if (j == 0 || v > bestv) { bestv = v; best = j; }
        }
        if (argmax) argmax[n] = best;
    }
}

/* ---- avlmaps/utils/visualize_utils.py:29-49 get_heatmap_from_mask_3d ---------------------------
 * heat = 1 on target voxels; elsewhere clip(1 - (min_t ||pos_t - pos||_2 / cell_size) * decay, 0, 1).
 * np.linalg.norm over int32 rows: squares summed in float64 (exact for these magnitudes), sqrt. */
AVLO_API void avlo_heatmap_from_mask(const int32_t* pos, const uint8_t* mask, long long N, double cell_size,
                                     double decay, float* heat) {
    long long nt = 0;
    for (long long i = 0; i < N; ++i) nt += mask[i] != 0;
    int32_t* tp = (int32_t*)malloc((size_t)(nt ? nt : 1) * 3 * sizeof(int32_t));
    long long k = 0;
    for (long long i = 0; i < N; ++i) if (mask[i]) { memcpy(tp + 3 * k, pos + 3 * i, 3 * sizeof(int32_t)); ++k; }
    for (long long i = 0; i < N; ++i) {
        if (mask[i]) { heat[i] = 1.0f; continue; }
        double best = INFINITY;
        for (long long t = 0; t < nt; ++t) {
            double dx = (double)tp[3 * t] - pos[3 * i], dy = (double)tp[3 * t + 1] - pos[3 * i + 1],
                   dz = (double)tp[3 * t + 2] - pos[3 * i + 2];
            double d = sqrt(dx * dx + dy * dy + dz * dz) / cell_size;
            if (d < best) best = d;
        }
        double v = 1.0 - best * decay;
        if (v < 0) v = 0; if (v > 1) v = 1;
        heat[i] = (float)v;
    }
    free(tp);
}
