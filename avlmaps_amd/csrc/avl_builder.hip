// Map builder for gfx950: depth back-projection + voxelisation + distance-weighted feature fusion.
//
// Replaces the per-frame body of the upstream reference's builder loop (path:line, upstream repo root):
//   avlmaps/map/vlmap_builder.py:129-178          per-point Python loop
//   avlmaps/utils/mapping_utils.py:226-251        depth2pc            (only for the sampled pixels)
//   avlmaps/utils/mapping_utils.py:305-315        transform_pc
//   avlmaps/utils/mapping_utils.py:345-349        base_pos2grid_id_3d (fp64 divide, trunc toward zero)
//   avlmaps/utils/mapping_utils.py:599-605        project_point       (rgb pixel, LSeg pixel)
//
// This file is compiled with -ffp-contract=off: the ONLY fused multiply-adds in the index math are the
// explicit fma() calls that mirror the reference's BLAS dgemm / dgemv summation (see oracle/avl_oracle.c).
//
// Device state (all in HBM, sized for a 288 GB part -- dense tables instead of a probing hash):
//   cell_slot  int32 [gs*gs*vh]   cell -> voxel id (-1 empty).  IS the reference's occupied_ids.
//   slot_cell  int32 [cap]        voxel id -> linear cell
//   slot_key   uint64[cap]        first-touch key (frame << 32 | position in the frame's sample list)
//   sum_feat   f64   [cap, D]     sum_i alpha_i * f_i           (hardware fp64 atomics)
//   sum_w4     f64   [cap, 4]     sum alpha, sum alpha*(r,g,b)
//   first_feat f32   [cap, D]     feature of the first-touch point, first_alpha f64 [cap]
// The reference's order-dependent result has the closed form
//   grid_feat = (sum_feat - a1*(1-a1)*first_feat) / sum_alpha,  weight = sum_alpha
// so the accumulation itself is commutative.  Voxel ids follow first touch: per frame, the earliest
// sample that hits an empty cell claims it (atomicMin on the claim word) and a prefix scan over the
// sample list hands out ids in that order -- slot r == the reference's voxel id r, no sort needed.
//
// Per frame three launches on the caller's stream:
//   K1 bp_voxelize   thread per sampled pixel : fp64 geometry, claims
//   K2 assign_slots  one 1024-thread workgroup: creator flags -> exclusive scan -> voxel ids + slot metadata
//   K3 accumulate    wave per sampled pixel   : channels-last feature gather (2 KB contiguous per point at
//                                               D=512) and fp64 atomic accumulation into the voxel row
#include <climits>

#include "avl_common.h"

namespace avl {

struct FrameParams {
    double kinv[9];   // inv(calib)            (mapping_utils.py:237)
    double k[9];      // calib                 (vlmap_builder.py:98)
    double kf[9];     // get_sim_cam_mat(Hf,Wf)(mapping_utils.py:591-596)
    double t[16];     // pc_transform          (vlmap_builder.py:133)
    double min_depth, max_depth, inv_two_sigma_sq_den;  // den = 2*sigma_sq
    double cs, half_gs;
    int H, W, Hf, Wf, gs, vh, P;
    unsigned long long frame_idx;
};

struct PointRec {
    double alpha;
    int32_t cell;   // -1 = inactive
    int32_t fpix;   // py*Wf + px into the (Hf, Wf, D) feature map
    uint32_t rgb;   // r | g<<8 | b<<16
    uint32_t first; // 1 if this point created its voxel
};

constexpr int kClaimBase = INT_MIN;  // claim word = kClaimBase + sample position  (< -1 == empty)

__device__ __forceinline__ long long py_int(double v) {
    // Python int(): truncate toward zero.  Saturate far-out values (they are out of range anyway).
    if (!(v > -4.0e18 && v < 4.0e18)) return v > 0 ? (long long)4e18 : (long long)-4e18;
    return (long long)v;  // C conversion truncates
}

__device__ __forceinline__ double gemv3(const double* a, double x0, double x1, double x2) {
    // numpy (3,3)@(3,1) -> OpenBLAS dgemv tail:  fma(a2,x2, fma(a0,x0, a1*x1))   (oracle/avl_oracle.c)
    return fma(a[2], x2, fma(a[0], x0, a[1] * x1));
}

__global__ __launch_bounds__(256) void bp_voxelize_kernel(FrameParams fp, const float* __restrict__ depth,
                                                          const int32_t* __restrict__ sample_idx,
                                                          const uint8_t* __restrict__ rgb, int32_t* __restrict__ cell_slot,
                                                          PointRec* __restrict__ recs, int* __restrict__ err_flags) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= fp.P) return;
    PointRec r;
    r.alpha = 0.0;
    r.cell = -1;
    r.fpix = 0;
    r.rgb = 0;
    r.first = 0;

    const int pix = sample_idx[s];
    bool ok = pix >= 0 && pix < fp.H * fp.W;
    double pl0 = 0, pl1 = 0, pl2 = 0;
    if (ok) {
        // depth2pc: p_2d = (u + 0.5, v + 0.5, 1); pc = Kinv @ p_2d (dgemm: FMA chain over k); pc *= z
        const double x = (double)(pix % fp.W) + 0.5, y = (double)(pix / fp.W) + 0.5, z = (double)depth[pix];
        pl0 = fma(fp.kinv[2], 1.0, fma(fp.kinv[1], y, fp.kinv[0] * x)) * z;
        pl1 = fma(fp.kinv[5], 1.0, fma(fp.kinv[4], y, fp.kinv[3] * x)) * z;
        pl2 = fma(fp.kinv[8], 1.0, fma(fp.kinv[7], y, fp.kinv[6] * x)) * z;
        ok = (pl2 > fp.min_depth) && (pl2 < fp.max_depth);  // strict on both sides, NaN fails
    }
    long long row = 0, col = 0, h = 0;
    if (ok) {
        // transform_pc: pose @ [pc; 1]  (dgemm FMA chain k = 0..3)
        const double g0 = fma(fp.t[3], 1.0, fma(fp.t[2], pl2, fma(fp.t[1], pl1, fp.t[0] * pl0)));
        const double g1 = fma(fp.t[7], 1.0, fma(fp.t[6], pl2, fma(fp.t[5], pl1, fp.t[4] * pl0)));
        const double g2 = fma(fp.t[11], 1.0, fma(fp.t[10], pl2, fma(fp.t[9], pl1, fp.t[8] * pl0)));
        // base_pos2grid_id_3d: int(gs/2 - int(x/cs)) with a true fp64 divide
        row = py_int(fp.half_gs - (double)py_int(g0 / fp.cs));
        col = py_int(fp.half_gs - (double)py_int(g1 / fp.cs));
        h = py_int(g2 / fp.cs);
        ok = !(col >= fp.gs || row >= fp.gs || h >= fp.vh || col < 0 || row < 0 || h < 0);
    }
    if (ok) {
        // project_point(calib, p_local) -> rgb[py, px] with numpy's negative-index wrap
        double q0 = gemv3(fp.k + 0, pl0, pl1, pl2), q1 = gemv3(fp.k + 3, pl0, pl1, pl2), q2 = gemv3(fp.k + 6, pl0, pl1, pl2);
        long long px = py_int(q0 / q2 - 0.5), py = py_int(q1 / q2 - 0.5);
        if (px < 0) px += fp.W;
        if (py < 0) py += fp.H;
        if (px < 0 || px >= fp.W || py < 0 || py >= fp.H) {
            atomicOr(err_flags, 2);  // the reference raises IndexError here; we drop the point and flag it
            ok = false;
        } else {
            const uint8_t* c = rgb + ((size_t)py * fp.W + px) * 3;
            r.rgb = (uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16);
        }
        // project_point(get_sim_cam_mat(Hf, Wf), p_local) -> feature pixel, bounds-checked (vlmap_builder.py:161)
        q0 = gemv3(fp.kf + 0, pl0, pl1, pl2);
        q1 = gemv3(fp.kf + 3, pl0, pl1, pl2);
        q2 = gemv3(fp.kf + 6, pl0, pl1, pl2);
        px = py_int(q0 / q2 - 0.5);
        py = py_int(q1 / q2 - 0.5);
        if (px < 0 || py < 0 || px >= fp.Wf || py >= fp.Hf) ok = false;
        r.fpix = (int32_t)(py * fp.Wf + px);
    }
    if (ok) {
        const double radial = (pl0 * pl0 + pl1 * pl1) + pl2 * pl2;  // np.sum(np.square(p_local))
        r.alpha = exp(-radial / fp.inv_two_sigma_sq_den);
        const int32_t cell = (int32_t)((row * fp.gs + col) * fp.vh + h);
        r.cell = cell;
        // claim an empty cell: the EARLIEST sample position wins (deterministic first touch)
        if (cell_slot[cell] < 0) atomicMin(&cell_slot[cell], kClaimBase + s);
    }
    recs[s] = r;
}

constexpr int kScanThreads = 1024;
constexpr int kScanItems = 8;

__global__ __launch_bounds__(kScanThreads) void assign_slots_kernel(int P, unsigned long long frame_idx,
                                                                    unsigned long long key_bias, int64_t capacity,
                                                                    int32_t* __restrict__ cell_slot, PointRec* __restrict__ recs,
                                                                    int32_t* __restrict__ slot_cell,
                                                                    unsigned long long* __restrict__ slot_key,
                                                                    double* __restrict__ first_alpha,
                                                                    long long* __restrict__ counters /* [0]=n_slots [1]=n_points */,
                                                                    int* __restrict__ err_flags) {
    __shared__ int wave_sums[kScanThreads / 64];
    __shared__ int chunk_total;
    __shared__ long long base_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) base_s = counters[0];
    __syncthreads();
    long long active_total = 0;

    for (int c0 = 0; c0 < P; c0 += kScanThreads * kScanItems) {
        int cells[kScanItems];
        int flags[kScanItems];
        int local = 0, nact = 0;
#pragma unroll
        for (int i = 0; i < kScanItems; ++i) {
            const int idx = c0 + tid * kScanItems + i;
            cells[i] = idx < P ? recs[idx].cell : -1;
        }
#pragma unroll
        for (int i = 0; i < kScanItems; ++i) {
            const int idx = c0 + tid * kScanItems + i;
            flags[i] = (cells[i] >= 0 && cell_slot[cells[i]] == kClaimBase + idx) ? 1 : 0;
            local += flags[i];
            nact += cells[i] >= 0;
        }
        active_total += nact;
        // block-wide exclusive scan of `local`
        int incl = local;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            int v = __shfl_up(incl, off, 64);
            if (lane >= off) incl += v;
        }
        if (lane == 63) wave_sums[wave] = incl;
        __syncthreads();
        if (wave == 0) {
            int v = lane < kScanThreads / 64 ? wave_sums[lane] : 0;
            int inc2 = v;
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) {
                int u = __shfl_up(inc2, off, 64);
                if (lane >= off) inc2 += u;
            }
            if (lane < kScanThreads / 64) wave_sums[lane] = inc2 - v;  // exclusive
            if (lane == kScanThreads / 64 - 1) chunk_total = inc2;
        }
        __syncthreads();
        const long long base = base_s;
        long long slot = base + wave_sums[wave] + (incl - local);
        const bool overflow = base + chunk_total > capacity;
#pragma unroll
        for (int i = 0; i < kScanItems; ++i) {
            if (flags[i]) {
                const int idx = c0 + tid * kScanItems + i;
                if (overflow) {
                    cell_slot[cells[i]] = -1;  // give the claim back; the voxel is dropped
                } else {
                    cell_slot[cells[i]] = (int32_t)slot;
                    slot_cell[slot] = cells[i];
                    slot_key[slot] = key_bias | (frame_idx << 32) | (unsigned)idx;
                    first_alpha[slot] = recs[idx].alpha;
                    recs[idx].first = 1;
                }
                ++slot;
            }
        }
        __syncthreads();
        if (tid == 0) {
            if (overflow) atomicOr(err_flags, 1);
            else base_s = base + chunk_total;
        }
        __syncthreads();
    }
    // active point statistics
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) active_total += __shfl_xor(active_total, off, 64);
    __shared__ long long act_s[kScanThreads / 64];
    if (lane == 0) act_s[wave] = active_total;
    __syncthreads();
    if (tid == 0) {
        long long t = 0;
        for (int w = 0; w < kScanThreads / 64; ++w) t += act_s[w];
        counters[0] = base_s;
        counters[1] += t;
    }
}

// wave per sampled point.  D floats per point are contiguous (channels-last): lanes take float4 chunks.
__global__ __launch_bounds__(256) void accumulate_kernel(int P, int D, const PointRec* __restrict__ recs,
                                                         const int32_t* __restrict__ cell_slot, const float* __restrict__ feat,
                                                         double* __restrict__ sum_feat, double* __restrict__ sum_w4,
                                                         float* __restrict__ first_feat) {
    const int lane = threadIdx.x & 63;
    const int s = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (s >= P) return;
    const PointRec r = recs[s];
    if (r.cell < 0) return;
    const int32_t slot = cell_slot[r.cell];
    if (slot < 0) return;  // voxel dropped on capacity overflow
    const double alpha = r.alpha;
    const float* f = feat + (size_t)r.fpix * D;
    double* acc = sum_feat + (size_t)slot * D;
    float* ff = first_feat + (size_t)slot * D;
    if ((D & 3) == 0) {
        for (int c = lane * 4; c < D; c += 256) {
            const float4 v = *reinterpret_cast<const float4*>(f + c);
            unsafeAtomicAdd(acc + c + 0, alpha * (double)v.x);
            unsafeAtomicAdd(acc + c + 1, alpha * (double)v.y);
            unsafeAtomicAdd(acc + c + 2, alpha * (double)v.z);
            unsafeAtomicAdd(acc + c + 3, alpha * (double)v.w);
            if (r.first) *reinterpret_cast<float4*>(ff + c) = v;
        }
    } else {
        for (int c = lane; c < D; c += 64) {
            const float v = f[c];
            unsafeAtomicAdd(acc + c, alpha * (double)v);
            if (r.first) ff[c] = v;
        }
    }
    if (lane < 4) {
        double v = alpha;
        if (lane > 0) v = alpha * (double)((r.rgb >> (8 * (lane - 1))) & 0xffu);
        unsafeAtomicAdd(sum_w4 + (size_t)slot * 4 + lane, v);
    }
}

// wave per voxel row
__global__ __launch_bounds__(256) void finalize_kernel(int64_t n, int D, int gs, int vh, const int32_t* __restrict__ cell,
                                                       const double* __restrict__ sum_feat, const double* __restrict__ sum_w4,
                                                       const float* __restrict__ first_feat,
                                                       const double* __restrict__ first_alpha, float* __restrict__ grid_feat,
                                                       int32_t* __restrict__ grid_pos, float* __restrict__ weight,
                                                       uint8_t* __restrict__ grid_rgb, int32_t* __restrict__ occupied) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r = wave0; r < n; r += nwaves) {
        const double w = sum_w4[r * 4];
        const double a1 = first_alpha[r];
        const double corr = a1 * (1.0 - a1);
        if (grid_feat) {
            const double* s = sum_feat + r * D;
            const float* f1 = first_feat + r * D;
            float* o = grid_feat + r * D;
            for (int c = lane; c < D; c += 64) o[c] = (float)((s[c] - corr * (double)f1[c]) / w);
        }
        if (lane == 0) {
            const int32_t cl = cell[r];
            if (grid_pos) {
                grid_pos[r * 3 + 0] = cl / (gs * vh);
                grid_pos[r * 3 + 1] = (cl / vh) % gs;
                grid_pos[r * 3 + 2] = cl % vh;
            }
            if (weight) weight[r] = (float)w;
            if (occupied) occupied[cl] = (int32_t)r;
        }
        if (grid_rgb && lane < 3) {
            // running mean stored into a uint8 array (truncating cast); we truncate the exact weighted mean
            double m = sum_w4[r * 4 + 1 + lane] / w;
            m = fmin(fmax(m, 0.0), 255.0);
            grid_rgb[r * 3 + lane] = (uint8_t)m;
        }
    }
}

// wave per imported voxel row: rebuild accumulators from a finalised map (resume, vlmap_builder.py:212-222)
__global__ __launch_bounds__(256) void import_map_kernel(int64_t n, int D, int gs, int vh, const float* __restrict__ grid_feat,
                                                         const int32_t* __restrict__ grid_pos, const float* __restrict__ weight,
                                                         const uint8_t* __restrict__ grid_rgb, int32_t* __restrict__ cell_slot,
                                                         int32_t* __restrict__ slot_cell, unsigned long long* __restrict__ slot_key,
                                                         double* __restrict__ sum_feat, double* __restrict__ sum_w4,
                                                         float* __restrict__ first_feat, double* __restrict__ first_alpha,
                                                         int* __restrict__ err_flags) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r = wave0; r < n; r += nwaves) {
        const double w = (double)weight[r];
        for (int c = lane; c < D; c += 64) {
            sum_feat[r * D + c] = (double)grid_feat[r * D + c] * w;   // first-touch weighting is already baked in
            first_feat[r * D + c] = 0.f;
        }
        if (lane == 0) {
            const int row = grid_pos[r * 3], col = grid_pos[r * 3 + 1], h = grid_pos[r * 3 + 2];
            if (row < 0 || row >= gs || col < 0 || col >= gs || h < 0 || h >= vh) {
                atomicOr(err_flags, 4);
            } else {
                const int32_t cell = (row * gs + col) * vh + h;
                cell_slot[cell] = (int32_t)r;
                slot_cell[r] = cell;
            }
            slot_key[r] = (unsigned long long)r;
            first_alpha[r] = 1.0;                                       // a1*(1-a1) == 0: no further correction
            sum_w4[r * 4] = w;
            for (int c = 0; c < 3; ++c) sum_w4[r * 4 + 1 + c] = (grid_rgb ? (double)grid_rgb[r * 3 + c] : 0.0) * w;
        }
    }
}

}  // namespace avl

using namespace avl;

struct avl_builder {
    int gs, vh, D;
    double cs;
    int64_t capacity;
    size_t ncell;
    int32_t* cell_slot = nullptr;
    int32_t* slot_cell = nullptr;
    unsigned long long* slot_key = nullptr;
    double* sum_feat = nullptr;
    double* sum_w4 = nullptr;
    float* first_feat = nullptr;
    double* first_alpha = nullptr;
    long long* counters = nullptr;  // [0] n_slots, [1] n_points
    int* err_flags = nullptr;
    PointRec* recs = nullptr;
    int recs_cap = 0;
    unsigned long long key_bias = 0;  // set after import_map so that imported voxels order before new ones
};

static int builder_check_flags(avl_builder* b, hipStream_t st) {
    int flags = 0;
    AVL_HIP_CHECK(hipMemcpyAsync(&flags, b->err_flags, sizeof(int), hipMemcpyDeviceToHost, st));
    AVL_HIP_CHECK(hipStreamSynchronize(st));
    if (flags & 1) {
        set_error("voxel capacity %lld exhausted: create the builder with a larger capacity", (long long)b->capacity);
        return AVL_ERR_CAPACITY;
    }
    if (flags & 2) {
        set_error("a sampled point projected outside the RGB image (the reference raises IndexError here)");
        return AVL_ERR_INVALID;
    }
    return AVL_OK;
}

extern "C" {

int avl_builder_reset(avl_builder* b, void* stream) {
    AVL_REQUIRE(b, "avl_builder_reset: null handle");
    hipStream_t st = as_stream(stream);
    AVL_HIP_CHECK(hipMemsetAsync(b->cell_slot, 0xFF, b->ncell * sizeof(int32_t), st));
    AVL_HIP_CHECK(hipMemsetAsync(b->sum_feat, 0, (size_t)b->capacity * b->D * sizeof(double), st));
    AVL_HIP_CHECK(hipMemsetAsync(b->sum_w4, 0, (size_t)b->capacity * 4 * sizeof(double), st));
    AVL_HIP_CHECK(hipMemsetAsync(b->first_feat, 0, (size_t)b->capacity * b->D * sizeof(float), st));
    AVL_HIP_CHECK(hipMemsetAsync(b->first_alpha, 0, (size_t)b->capacity * sizeof(double), st));
    AVL_HIP_CHECK(hipMemsetAsync(b->slot_cell, 0xFF, (size_t)b->capacity * sizeof(int32_t), st));
    AVL_HIP_CHECK(hipMemsetAsync(b->slot_key, 0xFF, (size_t)b->capacity * sizeof(unsigned long long), st));
    AVL_HIP_CHECK(hipMemsetAsync(b->counters, 0, 2 * sizeof(long long), st));
    AVL_HIP_CHECK(hipMemsetAsync(b->err_flags, 0, sizeof(int), st));
    b->key_bias = 0;
    return AVL_OK;
}

int avl_builder_destroy(avl_builder* b) {
    if (!b) return AVL_OK;
    (void)hipFree(b->cell_slot); (void)hipFree(b->slot_cell); (void)hipFree(b->slot_key); (void)hipFree(b->sum_feat);
    (void)hipFree(b->sum_w4); (void)hipFree(b->first_feat); (void)hipFree(b->first_alpha); (void)hipFree(b->counters);
    (void)hipFree(b->err_flags); (void)hipFree(b->recs);
    delete b;
    return AVL_OK;
}

int avl_builder_create(avl_builder** h_out, int gs, double cs, int vh, int D, int64_t capacity) {
    AVL_REQUIRE(h_out, "avl_builder_create: null output");
    *h_out = nullptr;
    AVL_REQUIRE(gs > 0 && vh > 0 && D > 0 && cs > 0 && capacity > 0, "avl_builder_create: bad parameters");
    const double ncell_d = (double)gs * gs * vh;
    AVL_REQUIRE(ncell_d < 2.0e9, "avl_builder_create: gs*gs*vh = %.0f cells exceeds the int32 cell index", ncell_d);
    AVL_REQUIRE(capacity < (1ll << 31), "avl_builder_create: capacity must fit int32 voxel ids");
    avl_builder* b = new avl_builder();
    b->gs = gs; b->vh = vh; b->D = D; b->cs = cs; b->capacity = capacity;
    b->ncell = (size_t)gs * gs * vh;
    hipError_t e = hipSuccess;
    auto alloc = [&](void** p, size_t bytes) { if (e == hipSuccess) e = hipMalloc(p, bytes); };
    alloc((void**)&b->cell_slot, b->ncell * sizeof(int32_t));
    alloc((void**)&b->slot_cell, (size_t)capacity * sizeof(int32_t));
    alloc((void**)&b->slot_key, (size_t)capacity * sizeof(unsigned long long));
    alloc((void**)&b->sum_feat, (size_t)capacity * D * sizeof(double));
    alloc((void**)&b->sum_w4, (size_t)capacity * 4 * sizeof(double));
    alloc((void**)&b->first_feat, (size_t)capacity * D * sizeof(float));
    alloc((void**)&b->first_alpha, (size_t)capacity * sizeof(double));
    alloc((void**)&b->counters, 2 * sizeof(long long));
    alloc((void**)&b->err_flags, sizeof(int));
    if (e != hipSuccess) {
        set_error("avl_builder_create: hipMalloc failed: %s (capacity %lld x D %d)", hipGetErrorString(e), (long long)capacity, D);
        avl_builder_destroy(b);
        return AVL_ERR_HIP;
    }
    int rc = avl_builder_reset(b, nullptr);
    if (rc == AVL_OK && hipDeviceSynchronize() != hipSuccess) rc = AVL_ERR_HIP;
    if (rc != AVL_OK) {
        avl_builder_destroy(b);
        return rc;
    }
    *h_out = b;
    return AVL_OK;
}

int avl_builder_integrate_frame(avl_builder* b, const float* d_depth, int H, int W, const double* h_calib,
                                const double* h_calib_inv, const double* h_pc_transform, const int32_t* d_sample_idx, int P,
                                const float* d_feat, int Hf, int Wf, const uint8_t* d_rgb, int64_t frame_idx,
                                double min_depth, double max_depth, double sigma_sq, void* stream) {
    AVL_REQUIRE(b, "avl_builder_integrate_frame: null handle");
    AVL_REQUIRE(H > 0 && W > 0 && Hf > 0 && Wf > 0 && P >= 0, "avl_builder_integrate_frame: bad shape");
    AVL_REQUIRE(P < (1 << 30), "avl_builder_integrate_frame: at most 2^30 samples per frame");
    AVL_REQUIRE(frame_idx >= 0 && frame_idx < (1ll << 31), "avl_builder_integrate_frame: bad frame_idx");
    AVL_REQUIRE(sigma_sq > 0, "avl_builder_integrate_frame: sigma_sq must be positive");
    if (P == 0) return AVL_OK;
    AVL_REQUIRE(d_depth && h_calib && h_calib_inv && h_pc_transform && d_sample_idx && d_feat && d_rgb,
                "avl_builder_integrate_frame: null pointer");
    hipStream_t st = as_stream(stream);
    if (P > b->recs_cap) {
        AVL_HIP_CHECK(hipStreamSynchronize(st));
        if (b->recs) AVL_HIP_CHECK(hipFree(b->recs));
        b->recs = nullptr;
        int cap = P + P / 4 + 1024;
        AVL_HIP_CHECK(hipMalloc((void**)&b->recs, (size_t)cap * sizeof(PointRec)));
        b->recs_cap = cap;
    }
    FrameParams fp;
    for (int i = 0; i < 9; ++i) { fp.kinv[i] = h_calib_inv[i]; fp.k[i] = h_calib[i]; fp.kf[i] = 0.0; }
    // get_sim_cam_mat(h, w): eye(3); [0,0] = [1,1] = w/2; [0,2] = w/2; [1,2] = h/2
    fp.kf[0] = fp.kf[4] = (double)Wf / 2.0;
    fp.kf[2] = (double)Wf / 2.0;
    fp.kf[5] = (double)Hf / 2.0;
    fp.kf[8] = 1.0;
    for (int i = 0; i < 16; ++i) fp.t[i] = h_pc_transform[i];
    fp.min_depth = min_depth; fp.max_depth = max_depth;
    fp.inv_two_sigma_sq_den = 2 * sigma_sq;
    fp.cs = b->cs; fp.half_gs = (double)b->gs / 2.0;
    fp.H = H; fp.W = W; fp.Hf = Hf; fp.Wf = Wf; fp.gs = b->gs; fp.vh = b->vh; fp.P = P;
    fp.frame_idx = (unsigned long long)frame_idx;

    hipLaunchKernelGGL(bp_voxelize_kernel, dim3((P + 255) / 256), dim3(256), 0, st, fp, d_depth, d_sample_idx, d_rgb,
                       b->cell_slot, b->recs, b->err_flags);
    hipLaunchKernelGGL(assign_slots_kernel, dim3(1), dim3(kScanThreads), 0, st, P, fp.frame_idx, b->key_bias, b->capacity, b->cell_slot,
                       b->recs, b->slot_cell, b->slot_key, b->first_alpha, b->counters, b->err_flags);
    hipLaunchKernelGGL(accumulate_kernel, dim3((P + 3) / 4), dim3(256), 0, st, P, b->D, b->recs, b->cell_slot, d_feat,
                       b->sum_feat, b->sum_w4, b->first_feat);
    AVL_HIP_CHECK(hipGetLastError());
    return AVL_OK;
}

static int read_counter(avl_builder* b, int which, int64_t* h_n, hipStream_t st) {
    long long v = 0;
    AVL_HIP_CHECK(hipMemcpyAsync(&v, b->counters + which, sizeof(long long), hipMemcpyDeviceToHost, st));
    AVL_HIP_CHECK(hipStreamSynchronize(st));
    *h_n = v;
    return AVL_OK;
}

int avl_builder_num_voxels(avl_builder* b, int64_t* h_n, void* stream) {
    AVL_REQUIRE(b && h_n, "avl_builder_num_voxels: null argument");
    int rc = builder_check_flags(b, as_stream(stream));
    if (rc != AVL_OK) return rc;
    return read_counter(b, 0, h_n, as_stream(stream));
}

int avl_builder_num_points(avl_builder* b, int64_t* h_n, void* stream) {
    AVL_REQUIRE(b && h_n, "avl_builder_num_points: null argument");
    return read_counter(b, 1, h_n, as_stream(stream));
}

int avl_finalize_raw(int64_t n, int D, int gs, int vh, const int32_t* d_cell, const double* d_sum_feat,
                     const double* d_sum_w4, const float* d_first_feat, const double* d_first_alpha, float* d_grid_feat,
                     int32_t* d_grid_pos, float* d_weight, uint8_t* d_grid_rgb, int32_t* d_occupied_ids, void* stream) {
    AVL_REQUIRE(n >= 0 && D > 0 && gs > 0 && vh > 0, "avl_finalize_raw: bad shape");
    if (n == 0) return AVL_OK;
    AVL_REQUIRE(d_cell && d_sum_w4 && d_first_alpha, "avl_finalize_raw: null input");
    AVL_REQUIRE(!d_grid_feat || (d_sum_feat && d_first_feat), "avl_finalize_raw: grid_feat needs sum_feat and first_feat");
    int64_t blocks = (n + 3) / 4;
    const int64_t maxb = (int64_t)num_cus() * 16;
    if (blocks > maxb) blocks = maxb;
    hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), n, D, gs, vh, d_cell, d_sum_feat,
                       d_sum_w4, d_first_feat, d_first_alpha, d_grid_feat, d_grid_pos, d_weight, d_grid_rgb, d_occupied_ids);
    AVL_HIP_CHECK(hipGetLastError());
    return AVL_OK;
}

int avl_builder_finalize(avl_builder* b, int64_t n, float* d_grid_feat, int32_t* d_grid_pos, float* d_weight,
                         uint8_t* d_grid_rgb, int32_t* d_occupied_ids, void* stream) {
    AVL_REQUIRE(b, "avl_builder_finalize: null handle");
    hipStream_t st = as_stream(stream);
    int64_t have = 0;
    int rc = avl_builder_num_voxels(b, &have, stream);
    if (rc != AVL_OK) return rc;
    AVL_REQUIRE(n == have, "avl_builder_finalize: n=%lld but the map holds %lld voxels", (long long)n, (long long)have);
    if (d_occupied_ids)  // voxel ids are assigned in reference order: the cell table IS occupied_ids
        AVL_HIP_CHECK(hipMemcpyAsync(d_occupied_ids, b->cell_slot, b->ncell * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    rc = avl_finalize_raw(n, b->D, b->gs, b->vh, b->slot_cell, b->sum_feat, b->sum_w4, b->first_feat, b->first_alpha,
                          d_grid_feat, d_grid_pos, d_weight, d_grid_rgb, nullptr, stream);
    if (rc != AVL_OK) return rc;
    AVL_HIP_CHECK(hipStreamSynchronize(st));
    return AVL_OK;
}

int avl_builder_import_map(avl_builder* b, int64_t n, const float* d_grid_feat, const int32_t* d_grid_pos,
                           const float* d_weight, const uint8_t* d_grid_rgb, void* stream) {
    AVL_REQUIRE(b, "avl_builder_import_map: null handle");
    AVL_REQUIRE(n >= 0 && n <= b->capacity, "avl_builder_import_map: %lld voxels exceed the capacity %lld", (long long)n,
                (long long)b->capacity);
    hipStream_t st = as_stream(stream);
    int64_t have = 0;
    int rc = avl_builder_num_voxels(b, &have, stream);
    if (rc != AVL_OK) return rc;
    if (have != 0) {
        set_error("avl_builder_import_map: the map already holds %lld voxels (import into an empty builder)", (long long)have);
        return AVL_ERR_STATE;
    }
    if (n == 0) return AVL_OK;
    AVL_REQUIRE(d_grid_feat && d_grid_pos && d_weight, "avl_builder_import_map: null input");
    int64_t blocks = (n + 3) / 4;
    const int64_t maxb = (int64_t)num_cus() * 16;
    if (blocks > maxb) blocks = maxb;
    hipLaunchKernelGGL(import_map_kernel, dim3((unsigned)blocks), dim3(256), 0, st, n, b->D, b->gs, b->vh, d_grid_feat, d_grid_pos,
                       d_weight, d_grid_rgb, b->cell_slot, b->slot_cell, b->slot_key, b->sum_feat, b->sum_w4, b->first_feat,
                       b->first_alpha, b->err_flags);
    const long long nn = n;
    AVL_HIP_CHECK(hipMemcpyAsync(b->counters, &nn, sizeof(long long), hipMemcpyHostToDevice, st));
    int flags = 0;
    AVL_HIP_CHECK(hipMemcpyAsync(&flags, b->err_flags, sizeof(int), hipMemcpyDeviceToHost, st));
    AVL_HIP_CHECK(hipStreamSynchronize(st));
    if (flags & 4) {
        set_error("avl_builder_import_map: a grid_pos row lies outside the (gs, gs, vh) grid");
        return AVL_ERR_INVALID;
    }
    b->key_bias = 1ull << 62;
    return AVL_OK;
}

int avl_builder_export_raw(avl_builder* b, int64_t n, int32_t* d_cell, uint64_t* d_first_key, double* d_sum_feat,
                           double* d_sum_w4, float* d_first_feat, double* d_first_alpha, void* stream) {
    AVL_REQUIRE(b, "avl_builder_export_raw: null handle");
    hipStream_t st = as_stream(stream);
    int64_t have = 0;
    int rc = avl_builder_num_voxels(b, &have, stream);
    if (rc != AVL_OK) return rc;
    AVL_REQUIRE(n >= 0 && n <= have, "avl_builder_export_raw: n=%lld but the map holds %lld voxels", (long long)n, (long long)have);
    if (n == 0) return AVL_OK;
    const size_t D = (size_t)b->D;
    auto cp = [&](void* dst, const void* src, size_t bytes) {
        return dst ? hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st) : hipSuccess;
    };
    AVL_HIP_CHECK(cp(d_cell, b->slot_cell, (size_t)n * sizeof(int32_t)));
    AVL_HIP_CHECK(cp(d_first_key, b->slot_key, (size_t)n * sizeof(uint64_t)));
    AVL_HIP_CHECK(cp(d_sum_feat, b->sum_feat, (size_t)n * D * sizeof(double)));
    AVL_HIP_CHECK(cp(d_sum_w4, b->sum_w4, (size_t)n * 4 * sizeof(double)));
    AVL_HIP_CHECK(cp(d_first_feat, b->first_feat, (size_t)n * D * sizeof(float)));
    AVL_HIP_CHECK(cp(d_first_alpha, b->first_alpha, (size_t)n * sizeof(double)));
    return AVL_OK;
}

}  // extern "C"
