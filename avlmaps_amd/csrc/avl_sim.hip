// Voxel x query similarity (+ fused row argmax) for gfx950.
//
// Replaces (upstream reference, path:line):
//   avlmaps/utils/clip_utils.py:227-229   scores_list = map_feats @ text_feats.T      (raw dot product)
//   avlmaps/map/vlmap.py:123-124          max_ids = np.argmax(scores_mat, axis=1)     (first max wins)
//
// HBM layout: grid_feat (N, D) float32 row-major (row stride ld), streamed exactly once per 64-query
// chunk; queries are tiny and live in LDS.  Two kernels:
//
//   sim_exact_kernel      float32 FMA on the vector ALU.  One wave per voxel row, lanes span D with
//                         16-byte loads (a row is one or two fully coalesced 1 KiB requests), up to 8
//                         query rows in LDS, wave-level shuffle reduction.  HBM-bound for Q <= 8.
//   sim_split_f16_kernel  for larger Q the fp32 vector/matrix rate (157 TF) is below what the HBM stream
//                         demands (AI = Q/2 flop/B), so the contraction runs on the fp16 matrix cores with
//                         an error-free hi/lo split:  a = ah + al, q = qh + ql (fp16 each, q pre-scaled by
//                         a power of two), a.q ~= ah.qh + ah.ql + al.qh  with fp32 accumulation
//                         (v_mfma_f32_32x32x16_f16 x3).  Dropped term and split residuals are ~2^-22
//                         relative, i.e. float32-class accuracy, at 1/5 of the fp32-MFMA time.
//                         MFMA "A" operand = 32 queries (from LDS), "B" operand = 32 voxels (straight
//                         from HBM into registers: lane (voxel j, half kg) owns one whole 128-byte line of
//                         its row per 64-wide k step, so every fetched line is consumed by the 8 loads of
//                         one lane).  The accumulator then holds, per lane, 16 queries of ONE voxel:
//                         the row argmax is an in-register scan plus one cross-half exchange.
#include <cfloat>
#include <algorithm>
#include <climits>
#include <cstdlib>
#include <vector>

#include "avl_common.h"

namespace avl {

using half8 = __attribute__((ext_vector_type(8))) _Float16;
using half2 = __attribute__((ext_vector_type(2))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

// ------------------------------------------------------------------------------------------------
// exact float32 path
// ------------------------------------------------------------------------------------------------
constexpr int kExactQB = 8;  // query rows resident in LDS per pass

template <bool VEC4>
__global__ __launch_bounds__(256) void sim_exact_kernel(const float* __restrict__ feat, int64_t N, int D, int64_t ld,
                                                        const float* __restrict__ q, int Q, int64_t ldq, int q0,
                                                        float* __restrict__ scores, int32_t* __restrict__ argmax,
                                                        float* __restrict__ best, int first_chunk) {
    extern __shared__ __attribute__((aligned(16))) float qs[];  // [kExactQB][Dp]
    const int Dp = (D + 3) & ~3;
    const int qn = min(kExactQB, Q - q0);
    for (int i = threadIdx.x; i < kExactQB * Dp; i += blockDim.x) {
        int j = i / Dp, d = i - j * Dp;
        qs[i] = (j < qn && d < D) ? q[(int64_t)(q0 + j) * ldq + d] : 0.f;
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    constexpr int R = 4;  // voxel rows per wave iteration: 4 x (D/256) 16-byte loads in flight per lane

    for (int64_t row0 = wave * R; row0 < N; row0 += nwaves * R) {
        const float* rp[R];
#pragma unroll
        for (int r = 0; r < R; ++r) rp[r] = feat + (row0 + r < N ? row0 + r : N - 1) * ld;
        float acc[R][kExactQB];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int j = 0; j < kExactQB; ++j) acc[r][j] = 0.f;
        if constexpr (VEC4) {
            const int nchunk = D >> 2;
            for (int c = lane; c < nchunk; c += 64) {
                f32x4 a[R];
#pragma unroll
                for (int r = 0; r < R; ++r) a[r] = *reinterpret_cast<const f32x4*>(rp[r] + 4 * c);
#pragma unroll
                for (int j = 0; j < kExactQB; ++j) {
                    const f32x4 b = *reinterpret_cast<const f32x4*>(qs + j * Dp + 4 * c);
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        acc[r][j] = fmaf(a[r].x, b.x, acc[r][j]);
                        acc[r][j] = fmaf(a[r].y, b.y, acc[r][j]);
                        acc[r][j] = fmaf(a[r].z, b.z, acc[r][j]);
                        acc[r][j] = fmaf(a[r].w, b.w, acc[r][j]);
                    }
                }
            }
        } else {
            for (int d = lane; d < D; d += 64) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const float a = rp[r][d];
#pragma unroll
                    for (int j = 0; j < kExactQB; ++j) acc[r][j] = fmaf(a, qs[j * Dp + d], acc[r][j]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int j = 0; j < kExactQB; ++j) {
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) acc[r][j] += __shfl_xor(acc[r][j], off, 64);
            }
        // lane r finishes row r
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t row = row0 + r;
            if (lane == r && row < N) {
                float bv = -INFINITY;
                int bi = q0;
                if (!first_chunk) {
                    if (best) bv = best[row];
                    if (argmax) bi = argmax[row];
                }
                bool have = !first_chunk;
#pragma unroll
                for (int j = 0; j < kExactQB; ++j) {
                    if (j < qn) {
                        if (scores) scores[row * (int64_t)Q + q0 + j] = acc[r][j];
                        if (!have || acc[r][j] > bv) {
                            bv = acc[r][j];
                            bi = q0 + j;
                            have = true;
                        }
                    }
                }
                if (argmax) argmax[row] = bi;
                if (best) best[row] = bv;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// split-fp16 MFMA path
// ------------------------------------------------------------------------------------------------
// workspace layout (device memory): [inv_scale: nqc*Qc floats, padded to kHdrAlign][fp16 images]
constexpr int kHdrAlign = 256;
constexpr int kRowPadHalves = 8;  // +16 B per query row: consecutive rows shift one 16-byte LDS slot
// 8 waves per workgroup, one workgroup per CU (LDS-bound).  Measured alternatives on the config-2 shape: 12 or 16 waves
// (0.81 / 1.37 ms vs 0.75 ms) and no register prefetch (1.10 ms) are slower.
constexpr int kSplitThreads = 512;
// register buffers of the unrolled k loop (resident kernel): the loads of the next ring - 1 steps are in flight while a step is
// computed.  Same-box A/B at 2 M x 512 x 64 (profiles/r03_ab_ring_depth.txt): raw float32 map 0.7100 (2) / 0.7053 (3) / 0.878 ms
// (4: spills); prepared 0.7006 / 0.7221 / 0.7187; compact 0.6083 / 0.6121 / 0.6108 -- so the raw kernel takes 3 since the epilogue
// rewrite of round 3 freed the registers (251 VGPRs), the others and the extra-row variant (spills at 3) stay at 2.
// The column-block variant (QM: 2 KB sub-rows of a wider row) is better off at 2 as well: config 5 on the raw map 2.3362 ms at 2,
// 2.3523 ms at 3 (same box, n = 3, profiles/r03_ab_config5_ring_tb.txt).
#ifdef AVL_RING
template <bool PRE, bool XR, bool QM> struct RingDepth { static constexpr int value = AVL_RING; };
#else
#ifndef AVL_RING_QM
#define AVL_RING_QM 2
#endif
template <bool PRE, bool XR, bool QM> struct RingDepth { static constexpr int value = (!PRE && !XR) ? (QM ? AVL_RING_QM : 3) : 2; };
#endif
constexpr int kTileRows = (kSplitThreads / 64) * 32;  // voxels per workgroup iteration

// One workgroup per (padded) query row: row max -> power-of-two scale 2^S with max|q|*2^S in [512, 1024)
// (keeps the fp16 lo parts out of the subnormal range), then the fp16 hi/lo images of the scaled row.
// image layout: [nkc][2 (hi, lo)][Qtot][KC + pad] fp16, Qtot = Q rounded up to 32
__global__ __launch_bounds__(256) void sim_prep_queries_kernel(const float* __restrict__ q, int Q, int D, int64_t ldq,
                                                               float* __restrict__ inv_scale, _Float16* __restrict__ img,
                                                               int Qtot, int KC, int nkc, int interleaved) {
    __shared__ float red[4];
    __shared__ float scale_s;
    const int qg = blockIdx.x;
    const bool live = qg < Q;
    const float* row = q + (int64_t)qg * ldq;
    float m = 0.f;
    if (live)
        for (int d = threadIdx.x; d < D; d += blockDim.x) m = fmaxf(m, fabsf(row[d]));
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        int S = 0;
        if (mx > 0.f && isfinite(mx)) S = 9 - ilogbf(mx);
        S = max(-60, min(60, S));
        scale_s = ldexpf(1.f, S);
        inv_scale[qg] = ldexpf(1.f, -S);
    }
    __syncthreads();
    const float scale = scale_s;
    const int rowlen = KC + kRowPadHalves;
    if (interleaved) {
        // streaming kernel: [nkc][Qtot][hi KC | lo KC | pad] -- a query chunk's rows are one contiguous block
        const int rowblk = 2 * KC + kRowPadHalves;
        for (int kc = 0; kc < nkc; ++kc) {
            _Float16* hi = img + ((int64_t)kc * Qtot + qg) * rowblk;
            for (int kk = threadIdx.x; kk < KC + kRowPadHalves; kk += blockDim.x) {
                const int k = kc * KC + kk;
                float v = 0.f;
                if (live && kk < KC && k < D) v = row[k] * scale;
                const half2 h = __builtin_bit_cast(half2, __builtin_amdgcn_cvt_pkrtz(v, 0.f));
                if (kk < KC) {
                    hi[kk] = h[0];
                    hi[KC + kk] = (_Float16)(v - (float)h[0]);
                } else {
                    hi[KC + kk] = (_Float16)0;
                }
            }
        }
        return;
    }
    for (int kc = 0; kc < nkc; ++kc) {
        _Float16* hi = img + ((int64_t)(kc * 2) * Qtot + qg) * rowlen;
        _Float16* lo = hi + (int64_t)Qtot * rowlen;
        for (int kk = threadIdx.x; kk < rowlen; kk += blockDim.x) {
            const int k = kc * KC + kk;
            float v = 0.f;
            if (live && kk < KC && k < D) v = row[k] * scale;
            const half2 h = __builtin_bit_cast(half2, __builtin_amdgcn_cvt_pkrtz(v, 0.f));
            hi[kk] = h[0];
            lo[kk] = (_Float16)(v - (float)h[0]);
        }
    }
}

// error-free split of 8 floats into fp16 hi (round-toward-zero) + fp16 lo (the residual x - hi is exact in fp32).
// NOTE: an inline-asm v_fma_mix{lo,hi}_f16 version (1.5 instead of 3 VALU ops per element) was measured: no gain
// (the kernel is not VALU-bound) and it is unsafe -- hipcc does not pad the MFMA-source WAR hazard for asm writes.
__device__ __forceinline__ void split8(const f32x4 v0, const f32x4 v1, half8& hi, half8& lo) {
#ifdef AVL_ABL_NOSPLIT
    hi = __builtin_bit_cast(half8, v0);
    lo = __builtin_bit_cast(half8, v1);
#else
    const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    float m1 = -1.0f;
    asm("" : "+v"(m1));   // opaque -1: keeps fma(hi, -1, x) from being rewritten as x - cvt(hi)
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        half2 h = __builtin_bit_cast(half2, __builtin_amdgcn_cvt_pkrtz(x[2 * p], x[2 * p + 1]));
        // x - hi as fma(hi, -1, x): same value (the difference is exact), and the f16 -> f32 extension folds into
        // v_fma_mix_f32 instead of a separate v_cvt_f32_f16
        float r0 = __builtin_fmaf((float)h[0], m1, x[2 * p]);
        float r1 = __builtin_fmaf((float)h[1], m1, x[2 * p + 1]);
        half2 l = __builtin_bit_cast(half2, __builtin_amdgcn_cvt_pkrtz(r0, r1));
        hi[2 * p] = h[0];
        hi[2 * p + 1] = h[1];
        lo[2 * p] = l[0];
        lo[2 * p + 1] = l[1];
    }
#endif
}

// Range guard of the on-the-fly split (raw float32 maps).  The map operand is converted to fp16 hi/lo UNSCALED (a per-row scale
// would need the row maximum before the first product), so a row is only as accurate as float32 when its largest |element|
// lies in [kGuardLo, kGuardHi): below, the elements sink into the fp16 subnormal range (a voxel touched once from 4 m away
// stores feat * exp(-r^2/1.2) ~ 1e-6, vlmap_builder.py:166-168) and above fp16 saturates.  Every lane tracks the maximum of
// the 32 floats it loads per k step (v_max3_f32 with |.| modifiers: 16 VALU ops per step); the epilogue publishes one 32-bit
// word per (wave, tile) with a bit per out-of-range / non-finite row, and sim_fixup_rows_kernel recomputes exactly those rows
// in float32 (np.argmax semantics for NaN).  Prepared maps carry a per-row power-of-two scale instead (avl_sim_prepare_map).
constexpr float kGuardLo = 0x1p-7f, kGuardHi = 0x1p15f;

__device__ __forceinline__ void guard_max8(float& m, const f32x4 v0, const f32x4 v1) {
#ifndef AVL_GUARD_BUILTIN
    asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(m) : "v"(v0.x), "v"(v0.y));
    asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(m) : "v"(v0.z), "v"(v0.w));
    asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(m) : "v"(v1.x), "v"(v1.y));
    asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(m) : "v"(v1.z), "v"(v1.w));
#else
    m = __builtin_fmaxf(__builtin_fmaxf(m, __builtin_fabsf(v0.x)), __builtin_fabsf(v0.y));
    m = __builtin_fmaxf(__builtin_fmaxf(m, __builtin_fabsf(v0.z)), __builtin_fabsf(v0.w));
    m = __builtin_fmaxf(__builtin_fmaxf(m, __builtin_fabsf(v1.x)), __builtin_fabsf(v1.y));
    m = __builtin_fmaxf(__builtin_fmaxf(m, __builtin_fabsf(v1.z)), __builtin_fabsf(v1.w));
#endif
}

// epilogue shared by the split-fp16 kernels: this lane holds voxel `row`, queries q_base + t*32 + 8g + 4kg + e (g<4, e<4) in
// acc[t][a][4g+e]; the NA partial accumulators are summed, scaled back by the per-query 2^-S, optionally stored, and reduced
// to the row's first maximum (one cross-half shuffle); later query chunks chain through `best`
// XR: the chunk's last 1..4 query rows (local rows 32 QT ..) were contracted by v_mfma_f32_4x4x4_16b_f16 instead of a mostly
// padded 32-row tile: xacc[r] = this lane's HALF (its 32 of the step's 64 columns) of voxel `row` . extra query r
template <int QT, int NA, bool XR = false>
__device__ __forceinline__ void split_epilogue(const f32x16 (&acc)[QT][NA], const float* isc, int q_base, int rows, int Q,
                                               float* __restrict__ scores, int32_t* __restrict__ argmax,
                                               float* __restrict__ best, int64_t row, int64_t N, int kg, int first_chunk,
                                               float rscale = 1.f, uint32_t* __restrict__ flags = nullptr, float rmax = 1.f,
                                               const int32_t* qml = nullptr, f32x4 xacc = f32x4{0.f, 0.f, 0.f, 0.f}) {
    // qml (column-block launches, avl_sim_scores_blocks): this launch's query rows are a gathered subset of the caller's;
    // qml[row of this chunk] = the caller's query index, ascending, so "first maximum" inside the launch is unchanged and only
    // the published index / score column and the tie-break against earlier launches use the caller's numbering.  The table
    // lives in LDS (filled at kernel start): a global-memory table made the compiler hoist its loads above the accumulator
    // reads and spill (17-37 VGPRs in the QM variants of round 2)
    // validity of this lane's 16 * QT columns as comparisons of compile-time constants against ONE per-lane value, opaque to the
    // optimiser: as loop invariants of the tile loop the 16 * QT masks and column indices were hoisted out of it, which cost the
    // kernels ~100 SGPRs (spilled to VGPR lanes) and up to 5 spilled VGPRs
    int lim = (XR ? 32 * QT : rows) - 4 * kg;         // local rows 8 g + e + 32 t (+ 4 kg) < rows (XR: the tiles are full)
    asm volatile("" : "+v"(lim));
    float bv = -INFINITY;
    int bl = INT_MAX;                // best local row minus 4 kg
#pragma unroll
    for (int t = 0; t < QT; ++t) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c0 = t * 32 + 8 * g;          // compile-time
            const int ql = c0 + 4 * kg;
            const int qg = q_base + ql;
            const f32x4 is4 = *reinterpret_cast<const f32x4*>(isc + ql);
            f32x4 v;
            f32x4 r = {acc[t][0][4 * g + 0], acc[t][0][4 * g + 1], acc[t][0][4 * g + 2], acc[t][0][4 * g + 3]};
#pragma unroll
            for (int a = 1; a < NA; ++a) {   // small cross-term sums first would be more accurate still; fp32 add suffices
                r.x += acc[t][a][4 * g + 0]; r.y += acc[t][a][4 * g + 1]; r.z += acc[t][a][4 * g + 2]; r.w += acc[t][a][4 * g + 3];
            }
            v.x = r.x * is4.x * rscale;   // rscale: the prepared map's per-row 2^-s (1 otherwise)
            v.y = r.y * is4.y * rscale;
            v.z = r.z * is4.z * rscale;
            v.w = r.w * is4.w * rscale;
            if (scores && row < N) {
                float* sp = scores + row * (int64_t)Q + qg;
                if (qml) {
                    float* sr = scores + row * (int64_t)Q;
                    const int4 qm4 = *reinterpret_cast<const int4*>(qml + ql);
                    if (c0 + 0 < lim) sr[qm4.x] = v.x;
                    if (c0 + 1 < lim) sr[qm4.y] = v.y;
                    if (c0 + 2 < lim) sr[qm4.z] = v.z;
                    if (c0 + 3 < lim) sr[qm4.w] = v.w;
                } else if (c0 + 3 < lim && (Q & 3) == 0) {
                    *reinterpret_cast<f32x4*>(sp) = v;
                } else {
                    if (c0 + 0 < lim) sp[0] = v.x;
                    if (c0 + 1 < lim) sp[1] = v.y;
                    if (c0 + 2 < lim) sp[2] = v.z;
                    if (c0 + 3 < lim) sp[3] = v.w;
                }
            }
            if (c0 + 0 < lim && v.x > bv) { bv = v.x; bl = c0 + 0; }
            if (c0 + 1 < lim && v.y > bv) { bv = v.y; bl = c0 + 1; }
            if (c0 + 2 < lim && v.z > bv) { bv = v.z; bl = c0 + 2; }
            if (c0 + 3 < lim && v.w > bv) { bv = v.w; bl = c0 + 3; }
        }
    }
    int bi = bl == INT_MAX ? INT_MAX : q_base + 4 * kg + bl;
    float xv[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (XR) {
        const f32x4 xis = *reinterpret_cast<const f32x4*>(isc + 32 * QT);
#pragma unroll
        for (int r = 0; r < 4; ++r) xv[r] = (xacc[r] + __shfl_xor(xacc[r], 32, 64)) * xis[r] * rscale;   // both halves now hold the sum
        if (scores && row < N && kg == 0) {
            float* sr = scores + row * (int64_t)Q;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (32 * QT + r < rows) sr[qml ? qml[32 * QT + r] : q_base + 32 * QT + r] = xv[r];
        }
    }
    if (argmax || best || flags) {
        if (qml && bi != INT_MAX) bi = qml[bi - q_base];
        const float ov = __shfl_xor(bv, 32, 64);
        const int oi = __shfl_xor(bi, 32, 64);
        if (ov > bv || (ov == bv && oi < bi)) {
            bv = ov;
            bi = oi;
        }
        if constexpr (XR) {   // the extra rows come last in the chunk: a strictly greater score only
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (32 * QT + r < rows && xv[r] > bv) {
                    bv = xv[r];
                    bi = qml ? qml[32 * QT + r] : q_base + 32 * QT + r;
                }
        }
        if (flags) {
            // one word per 32-row unit: bit j = row j of the unit must be recomputed in float32 (largest |element| outside
            // the range the unscaled fp16 split resolves, non-finite, or no score compared greater than -inf: NaN)
            const float m = fmaxf(rmax, __shfl_xor(rmax, 32, 64));
            const bool bad = row < N && (!(m == 0.f || (m >= kGuardLo && m < kGuardHi)) || bi == INT_MAX);
            const unsigned long long mask = __ballot(bad);
            const int64_t row0 = __shfl(row, 0, 64);
            if ((threadIdx.x & 63) == 0 && row0 < N) {   // later launches of one call (other query chunks / column blocks) OR in
                const uint32_t w = (uint32_t)(mask | (mask >> 32));
                flags[row0 >> 5] = first_chunk ? w : (flags[row0 >> 5] | w);
            }
        }
        if ((argmax || best) && kg == 0 && row < N) {
            if (bi == INT_MAX) bi = qml ? qml[0] : q_base;
            if (!first_chunk) {  // an equal score keeps the lower query index (np.argmax: first maximum)
                const float pv = best[row];
                const int pi = argmax ? argmax[row] : -1;
                if (pv > bv || (pv == bv && pi < bi)) {
                    bv = pv;
                    bi = argmax ? pi : bi;
                }
            }
            if (argmax) argmax[row] = bi;
            if (best) best[row] = bv;
        }
    }
}

// in-place conversion of a float32 map to the split layout: group g of 8 floats (32 B) -> hi[8] fp16 (16 B) | lo[8] fp16 (16 B),
// the same split8() the kernel applies on the fly.  Wave per row.  With row_scale != nullptr the row is first multiplied by
// the power of two 2^s that brings its largest |element| into [2^14, 2^15) -- exact, and it keeps EVERY row at the full
// ~22 bits of the hi/lo pair whatever its magnitude (a voxel seen once from 5 m away holds feat * 1e-9) -- and 2^-s is stored
// in row_scale[row] for the kernel's epilogue.  Without it (s = 0) prepared and raw maps give bit-identical scores.
// Rows that are all zero or contain a non-finite value are left unscaled; the latter get row_scale = NaN (all their scores are NaN).
__global__ __launch_bounds__(256) void sim_prepare_map_kernel(float* __restrict__ feat, int64_t N, int D, int64_t ld,
                                                              float* __restrict__ row_scale) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int gpr = D >> 3;  // groups of 8 floats per row
    for (int64_t row = wave0; row < N; row += nwaves) {
        f32x4* p = reinterpret_cast<f32x4*>(feat + row * ld);
        float scale = 1.f;
        if (row_scale) {
            unsigned mb = 0;   // max over |x| as integer bits: NaN / inf order above every finite value
            for (int g = lane; g < 2 * gpr; g += 64) {
                const f32x4 v = p[g];
                mb = max(max(mb, __float_as_uint(v.x) & 0x7fffffffu), __float_as_uint(v.y) & 0x7fffffffu);
                mb = max(max(mb, __float_as_uint(v.z) & 0x7fffffffu), __float_as_uint(v.w) & 0x7fffffffu);
            }
            for (int off = 32; off > 0; off >>= 1) mb = max(mb, (unsigned)__shfl_xor((int)mb, off, 64));
            int sh = 0;
            if (mb != 0 && mb < 0x7f800000u) sh = max(-100, min(100, 14 - ilogbf(__uint_as_float(mb))));
            scale = ldexpf(1.f, sh);
            // a row with a non-finite element scores NaN against EVERY query, as in NumPy (0 * nan = nan): also against the
            // queries of a column-block launch whose window does not contain the element
            if (lane == 0) row_scale[row] = mb >= 0x7f800000u ? __uint_as_float(0x7fc00000u) : ldexpf(1.f, -sh);
        }
        for (int g = lane; g < gpr; g += 64) {
            f32x4 v0 = p[2 * g], v1 = p[2 * g + 1];
            v0.x *= scale; v0.y *= scale; v0.z *= scale; v0.w *= scale;
            v1.x *= scale; v1.y *= scale; v1.z *= scale; v1.w *= scale;
            half8 hi, lo;
            split8(v0, v1, hi, lo);
            p[2 * g] = __builtin_bit_cast(f32x4, hi);
            p[2 * g + 1] = __builtin_bit_cast(f32x4, lo);
        }
    }
}

// Compact prepared form (P24 kernels): row r of the output holds, per 32 columns, fp16 hi[32] (64 B, round to nearest) and then one
// byte per element with the residual x - hi in units of ulp(hi) / 256 (u = k + 128, k in [-128, 127]; 32 B); every row is first
// scaled by the power of two that brings its largest |element| into [2^14, 2^15) and 2^-s goes to row_scale[row].  The residual's
// exponent is implied by hi, so all 8 bits are significand: an element is within 2^-19 of its value (hi's 11 bits + 8), against
// 2^-22 for the 4-byte forms; elements below 2^-11 of the row's maximum (E < 4) keep hi only.  Rows with a non-finite element
// get row_scale = NaN like sim_prepare_map_kernel.
// byte of column c inside the residual plane of a compact row (layout: see compact_load_step below)
__host__ __device__ __forceinline__ int compact_lo_offset(int c) { return 128 * (c >> 7) + 64 * ((c >> 5) & 1) + 32 * ((c >> 6) & 1) + (c & 31); }

__global__ __launch_bounds__(256) void sim_prepare_map24_kernel(const float* __restrict__ feat, int64_t N, int D, int64_t ld,
                                                                unsigned char* __restrict__ out, float* __restrict__ row_scale) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int g8 = D >> 3;   // groups of 8 columns per row
    for (int64_t row = wave0; row < N; row += nwaves) {
        const f32x4* p = reinterpret_cast<const f32x4*>(feat + row * ld);
        unsigned mb = 0;
        for (int g = lane; g < 2 * g8; g += 64) {
            const f32x4 v = p[g];
            mb = max(max(mb, __float_as_uint(v.x) & 0x7fffffffu), __float_as_uint(v.y) & 0x7fffffffu);
            mb = max(max(mb, __float_as_uint(v.z) & 0x7fffffffu), __float_as_uint(v.w) & 0x7fffffffu);
        }
        for (int off = 32; off > 0; off >>= 1) mb = max(mb, (unsigned)__shfl_xor((int)mb, off, 64));
        int sh = 0;
        if (mb != 0 && mb < 0x7f800000u) sh = max(-100, min(100, 14 - ilogbf(__uint_as_float(mb))));
        const float scale = ldexpf(1.f, sh);
        if (lane == 0) row_scale[row] = mb >= 0x7f800000u ? __uint_as_float(0x7fc00000u) : ldexpf(1.f, -sh);
        unsigned char* orow = out + row * ((int64_t)D * 3);
        for (int g = lane; g < g8; g += 64) {           // 8 columns: 16 B in the hi plane, 8 B in the residual plane
            const f32x4 v0 = p[2 * g], v1 = p[2 * g + 1];
            const float x[8] = {v0.x * scale, v0.y * scale, v0.z * scale, v0.w * scale, v1.x * scale, v1.y * scale, v1.z * scale, v1.w * scale};
            half8 hi;
            unsigned u[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const _Float16 h = (_Float16)x[e];      // (a scalar: bit-casting a vector ELEMENT is what this compiler folds to element 0)
                hi[e] = h;
                const int eb = (__builtin_bit_cast(unsigned short, h) >> 10) & 31;          // biased exponent of hi
                int k = 0;
#ifndef AVL_COMPACT_FIXED_UNITS
                if (eb > 18 && eb < 31) k = (int)rintf((x[e] - (float)h) * ldexpf(1.f, 33 - eb));       // units of 2^(E - 18), E = eb - 15
#else
                if (eb < 31) k = (int)rintf((x[e] - (float)h) * 16.f);                       // units of 2^-4 (the row's largest element is in [2^14, 2^15))
#endif
                u[e] = (unsigned)(max(-128, min(127, k)) + 128);
            }
            const int w0 = (int)(u[0] | (u[1] << 8) | (u[2] << 16) | (u[3] << 24));
            const int w1 = (int)(u[4] | (u[5] << 8) | (u[6] << 16) | (u[7] << 24));
            *reinterpret_cast<half8*>(orow + (size_t)g * 16) = hi;                                     // hi plane: column c at 2 c
            *reinterpret_cast<int2*>(orow + 2 * (size_t)D + compact_lo_offset(8 * g)) = int2{w0, w1};  // residual plane
        }
    }
}

// Recompute the rows the range guard flagged (one bit per row, one word per 32 rows) in float32 on the vector ALU, wave per
// row: every score, and the row argmax with np.argmax semantics (first NaN wins, else first maximum).  Flagged rows are rare
// on in-range maps (none at all on LSeg-scale rows), so this is a ~3 us scan of N/32 words; on a map full of tiny rows it is
// the slow-but-correct path and the prepared map (per-row scale) is the fast one.
__global__ __launch_bounds__(256) void sim_fixup_rows_kernel(const float* __restrict__ feat, int64_t N, int D, int64_t ld,
                                                             const float* __restrict__ q, int Q, int64_t ldq,
                                                             const uint32_t* __restrict__ flags, float* __restrict__ scores,
                                                             int32_t* __restrict__ argmax, float* __restrict__ best) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t nwords = (N + 31) >> 5;
    for (int64_t base = wave0 * 64; base < nwords; base += nwaves * 64) {
        const uint32_t w = base + lane < nwords ? flags[base + lane] : 0u;
        unsigned long long any = __ballot(w != 0u);
        while (any) {
            const int l = __ffsll((long long)any) - 1;
            any &= any - 1;
            uint32_t word = (uint32_t)__shfl((int)w, l, 64);
            while (word) {
                const int b = __ffs((int)word) - 1;
                word &= word - 1;
                const int64_t row = (base + l) * 32 + b;
                if (row >= N) continue;
                const float* a = feat + row * ld;
                float bv = -INFINITY;
                int bi = 0;
                bool have = false, nan_seen = false;
                for (int qi = 0; qi < Q; ++qi) {
                    const float* qr = q + (int64_t)qi * ldq;
                    float sum = 0.f;
                    for (int d = lane; d < D; d += 64) sum = fmaf(a[d], qr[d], sum);
                    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
                    if (lane == 0 && scores) scores[row * (int64_t)Q + qi] = sum;
                    if (!nan_seen) {
                        if (sum != sum) { nan_seen = true; bv = sum; bi = qi; }
                        else if (!have || sum > bv) { bv = sum; bi = qi; have = true; }
                    }
                }
                if (lane == 0) {
                    if (argmax) argmax[row] = bi;
                    if (best) best[row] = bv;
                }
            }
        }
    }
}

// column-block launches: out[r, 0:cols] = q[rows[r], col0 : col0 + cols]  (the gathered query rows of one support group)
__global__ __launch_bounds__(256) void sim_gather_queries_kernel(const float* __restrict__ q, int64_t ldq, const int32_t* __restrict__ rows,
                                                                 int nrows, int col0, int cols, float* __restrict__ out) {
    const int r = blockIdx.x;
    if (r >= nrows) return;
    const int32_t sr = rows[r];                 // -1: a zero row
    const float* src = q + (int64_t)(sr < 0 ? 0 : sr) * ldq + col0;
    float* dst = out + (int64_t)r * cols;
    for (int c = threadIdx.x; c < cols; c += blockDim.x) dst[c] = sr < 0 ? 0.f : src[c];
}

// QT = number of 32-query MFMA tiles of this chunk (1..3); `rows` = valid query rows of the chunk (<= 32*QT):
// only those rows are resident in LDS (lanes of a partial tile re-read the last valid row; their results are masked).
// PRE: the map was converted in place by sim_prepare_map_kernel -- every 8 floats hold their fp16 hi[8] | lo[8] images, so
// the loaded registers ARE the MFMA operands and the fp32->fp16 split (8 % of the kernel time when power-throttled) is gone
// FQ (needs nkc == 1 and D <= 512): the workgroup builds its LDS query image itself from the raw float32 query rows -- the
// same arithmetic as sim_prep_queries_kernel, so the scores are bit-identical -- instead of copying a prepared image: no
// prep launch and no workspace, which is ~8 us per query on maps of a few hundred thousand voxels.
// Residuals of the compact prepared form: byte u of word w encodes k = u - 128 units of ulp(hi) / 256 = 2^(E - 18), E = exponent of
// the hi value it belongs to.  Two of them -> packed fp16: v_perm_b32 builds the fp16 bit patterns 0x6400 | u = 1024 + u, a packed add
// of -1152 gives k exactly, and the packed fp16 2^(E - 18) comes from hi's own exponent field (saturating subtract: 0 below
// E = 4, where the residual is dropped).  Five vector-ALU instructions per two elements.
// ---- the COMPACT resident form (3 bytes per element, D a multiple of 128), round-4 layout: two planes per row.
//   hi plane   [0, 2 D):   fp16 hi of column c at 2 c -- the 128-byte line of a 64-column step is consumed by the two lane halves
//                          of a wave in the same group of loads;
//   residuals  [2 D, 3 D): one byte per column, arranged per block of 128 columns (one line) so that the 32 + 32 bytes lane half kg
//                          needs in the block's two steps are contiguous: byte of column c at 128 (c >> 7) + 64 ((c >> 5) & 1) +
//                          32 ((c >> 6) & 1) + (c & 31).
// The first compact layout interleaved hi[32] | residuals[32] per 96 bytes: a step was 192 B = one and a half lines, every other step
// boundary fell inside a line, and that line was requested twice (L1 -> L2 requests 1.2 x the lines, profiles/r04_tcc_requests_raw_vs_compact.txt;
// a timing ablation of this layout: 0.617 -> 0.542 ms at 2 M x 512 x 64).

// one 64-column step of a compact row for lane half kg: 4 x 16 B of hi -> b[0..3], its 32 residual bytes -> b[4..5].
// row = first byte of the map row, ld = columns of a full row, sa = the step's index in the row (column / 64)
__device__ __forceinline__ void compact_load_step(f32x4 (&b)[8], const char* row, int64_t ld, int kg, int sa) {
    const f32x4* gh = reinterpret_cast<const f32x4*>(row + 128 * sa + 64 * kg);
    const f32x4* gl = reinterpret_cast<const f32x4*>(row + 2 * ld + 128 * (sa >> 1) + 64 * kg + 32 * (sa & 1));
#pragma unroll
    for (int t = 0; t < 4; ++t) b[t] = gh[t];
#pragma unroll
    for (int t = 0; t < 2; ++t) b[4 + t] = gl[t];
}

template <int PAIR>
__device__ __forceinline__ half2 residual_pair(unsigned w, unsigned hi2) {
#ifndef AVL_COMPACT_FIXED_UNITS
    using ushort2v = __attribute__((ext_vector_type(2))) unsigned short;
    const unsigned pat = __builtin_amdgcn_perm(0x64646464u, w, PAIR == 0 ? 0x04010400u : 0x04030402u);
    const half2 k = __builtin_bit_cast(half2, pat) + half2{(_Float16)-1152.0f, (_Float16)-1152.0f};
    const ushort2v e = __builtin_bit_cast(ushort2v, hi2 & 0x7C007C00u);
    const ushort2v sc = __builtin_elementwise_sub_sat(e, ushort2v{0x4800, 0x4800});
    return k * __builtin_bit_cast(half2, sc);
#else
    // round 6 experiment (-DAVL_COMPACT_FIXED_UNITS, NOT the default: same-box A/B 0.585 -> 0.570 ms at 2 M x 512 x 64, 1.817 -> 1.779 ms on config 5,
    // i.e. 2 %, for a max score error of 8.3e-6 instead of 1.8e-6 -- profiles/r06_ab_compact_units.txt): the residual in units of 2^-4 of the ROW-SCALED value (the row's largest element lies in [2^14, 2^15), where ulp(hi) / 256 is
    // exactly 2^-4): fp16 bit pattern 0x5400 | u = 64 + u / 16, minus 72 = (u - 128) / 16 -- TWO vector-ALU instructions per two elements
    // instead of five (the rebuild, not the bytes, bounded the compact kernels: MFMA busy 0.42)
    (void)hi2;
    const unsigned pat = __builtin_amdgcn_perm(0x54545454u, w, PAIR == 0 ? 0x04010400u : 0x04030402u);
    return __builtin_bit_cast(half2, pat) + half2{(_Float16)-72.0f, (_Float16)-72.0f};
#endif
}

// B operands (voxel side) of the m-th 8-column group of a 64-column step from the registers one lane loaded for it:
// raw float32 (guard + on-the-fly split), prepared hi[8] | lo[8], or the compact form (hi[32] | residual bytes[32] per 96 B)
// compact form: the 8 hi values of one k group (16 B) and the 16-byte register that holds their residual bytes (second = which half)
__device__ __forceinline__ void compact_operands(const f32x4& hi, const f32x4& lo, int second, half8& bh, half8& bl) {
    bh = __builtin_bit_cast(half8, hi);
    // (bit-casting ONE element of a float ext_vector to int folded every element to element 0 in this compiler;
    // cast the whole vector first)
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
    const u32x4 hw = __builtin_bit_cast(u32x4, hi);                       // the 8 hi values, two per word
    const u32x4 lw = __builtin_bit_cast(u32x4, lo);
    const unsigned w0 = lw[second * 2], w1 = lw[second * 2 + 1];          // their 8 residual bytes
    const half2 l0 = residual_pair<0>(w0, hw[0]), l1 = residual_pair<1>(w0, hw[1]);
    const half2 l2 = residual_pair<0>(w1, hw[2]), l3 = residual_pair<1>(w1, hw[3]);
    bl = half8{l0[0], l0[1], l1[0], l1[1], l2[0], l2[1], l3[0], l3[1]};
}

template <bool PRE, bool P24>
__device__ __forceinline__ void map_operands(const f32x4 (&b)[8], int m, half8& bh, half8& bl, float& rmax) {
    if constexpr (P24) {
        compact_operands(b[m], b[4 + (m >> 1)], m & 1, bh, bl);
    } else if constexpr (PRE) {
        bh = __builtin_bit_cast(half8, b[2 * m]);
        bl = __builtin_bit_cast(half8, b[2 * m + 1]);
    } else {
#ifndef AVL_ABL_NOGUARD
        guard_max8(rmax, b[2 * m], b[2 * m + 1]);
#endif
        split8(b[2 * m], b[2 * m + 1], bh, bl);
    }
}

// P24 (with PRE): the map is the COMPACT prepared form of sim_prepare_map24_kernel -- per 32 columns 64 B of fp16 hi[32] followed by
// 32 B of int8 residuals in units of ulp(hi) / 256, 3 bytes per element instead of 4: a quarter less HBM traffic per pass, the
// residuals are rebuilt as fp16 in registers (residual_pair) and the three MFMAs stay fp16.
// XR (with FQ): rows = 32 QT + 1..4 -- the last 1..4 query rows ("64 categories + other", clip_utils.py:213-215: 65 columns) are
// contracted by v_mfma_f32_4x4x4_16b_f16 (16 blocks of 4 voxel-halves x the same 4 query rows: D[lane][r] = sum_k A[4 (lane / 4)
// + r][k] * B[lane][k], tools/probe_mfma4.hip) instead of a third 32-row tile with one live row: 1/16 of its multiply-adds in
// half of its issue cycles -- at the power cap that is time (Q = 65: profiles/HISTORY.md 4.1)
template <int QT, int NSTEPS, bool PRE, bool FQ, bool QM = false, bool P24 = false, bool XR = false>
__global__ __launch_bounds__(kSplitThreads) void sim_split_f16_kernel(
    const float* __restrict__ feat, int64_t N, int D, int64_t ld, const _Float16* __restrict__ img,
    const float* __restrict__ inv_scale, const float* __restrict__ q_raw, int64_t ldq, int Qtot, int KC, int nkc, int q_base,
    int rows, int Q, float* __restrict__ scores, int32_t* __restrict__ argmax, float* __restrict__ best, int first_chunk,
    const float* __restrict__ row_scale, uint32_t* __restrict__ flags, const int32_t* __restrict__ qmap, int col0) {
    // col0 (P24 only): first column of the window this launch contracts; a compact map's `feat` is never offset by a window (two planes)
    // row_scale (PRE only, nullable): per-row 2^-s of a map prepared with scaling.  flags (!PRE): range-guard words, see kGuardLo.
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NA = QT == 1 ? 3 : (QT == 2 ? 2 : 1);   // accumulator sets per tile (register budget: 16 VGPRs each)
    constexpr int X1 = NA > 1 ? 1 : 0, X2 = NA > 2 ? 2 : X1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, kg = lane >> 5;
    const int row_b = (KC + kRowPadHalves) * 2;  // bytes per query row
    // FQ: a partial last MFMA tile gets one extra all-zero row that its padding lanes read -- multiplying zeros toggles far
    // fewer matrix-core bits than re-reading a valid row, and at the power cap that is time (Q = 65: profiles/HISTORY.md 4.1)
    static_assert(!XR || FQ, "the extra-row path builds its query image in the kernel");
    constexpr int TQ = (QT + (XR ? 1 : 0)) * 32;   // rows of the per-query tables
    const int zrow = (FQ && rows < (XR ? 32 * QT + 4 : 32 * QT)) ? 1 : 0;
    const int img_b = (rows + zrow) * row_b;     // bytes of the hi (or lo) image resident in LDS
    float* isc = reinterpret_cast<float*>(smem + 2 * img_b);  // per-query 2^-S of this chunk
    int32_t* qml = reinterpret_cast<int32_t*>(isc + TQ);  // QM: the caller's query index of every row of this chunk
    if constexpr (QM) {
        if (threadIdx.x < TQ) qml[threadIdx.x] = threadIdx.x < rows ? qmap[q_base + threadIdx.x] : INT_MAX;
    }
    if constexpr (!FQ) {
        if (threadIdx.x < TQ) isc[threadIdx.x] = threadIdx.x < rows ? inv_scale[q_base + threadIdx.x] : 0.f;
    } else {
        if (threadIdx.x >= rows && threadIdx.x < TQ) isc[threadIdx.x] = 0.f;
        // wave w converts query rows w, w+8, ...: the loads of four rows are issued together (lane owns k = lane + 64 t),
        // then row max -> power-of-two scale -> fp16 hi/lo straight into the LDS image, all from registers
        constexpr int RB = 4, NW = kSplitThreads / 64;
        for (int r0 = wave; r0 < rows; r0 += NW * RB) {
            float v[RB][8];
#pragma unroll
            for (int b = 0; b < RB; ++b) {
                const int r = r0 + NW * b;
                const float* qr = q_raw + (int64_t)(q_base + (r < rows ? r : rows - 1)) * ldq;
#pragma unroll
                for (int t = 0; t < 8; ++t) v[b][t] = (lane + 64 * t < D) ? qr[lane + 64 * t] : 0.f;
            }
#pragma unroll
            for (int b = 0; b < RB; ++b) {
                const int r = r0 + NW * b;
                if (r < rows) {
                    float m = 0.f;
#pragma unroll
                    for (int t = 0; t < 8; ++t) m = fmaxf(m, fabsf(v[b][t]));
                    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
                    int S = 0;
                    if (m > 0.f && isfinite(m)) S = 9 - ilogbf(m);
                    S = max(-60, min(60, S));
                    const float scale = ldexpf(1.f, S);
                    if (lane == 0) isc[r] = ldexpf(1.f, -S);
                    _Float16* hi = reinterpret_cast<_Float16*>(smem + r * row_b);
                    _Float16* lo = reinterpret_cast<_Float16*>(smem + img_b + r * row_b);
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const int k = lane + 64 * t;
                        if (k < KC) {
                            const float x = v[b][t] * scale;
                            const half2 h = __builtin_bit_cast(half2, __builtin_amdgcn_cvt_pkrtz(x, 0.f));
                            hi[k] = h[0];
                            lo[k] = (_Float16)(x - (float)h[0]);
                        }
                    }
                    if (lane < kRowPadHalves) {
                        hi[KC + lane] = (_Float16)0;
                        lo[KC + lane] = (_Float16)0;
                    }
                }
            }
        }
        if (zrow) {
            uint32_t* zh = reinterpret_cast<uint32_t*>(smem + rows * row_b);
            uint32_t* zl = reinterpret_cast<uint32_t*>(smem + img_b + rows * row_b);
            for (int i = threadIdx.x; i < row_b / 4; i += kSplitThreads) {
                zh[i] = 0u;
                zl[i] = 0u;
            }
        }
    }

    auto fill_lds = [&](int kc) {
        const char* hi = reinterpret_cast<const char*>(img) + ((int64_t)(kc * 2) * Qtot + q_base) * row_b;
        const char* lo = hi + (int64_t)Qtot * row_b;
        const uint4* s0 = reinterpret_cast<const uint4*>(hi);
        const uint4* s1 = reinterpret_cast<const uint4*>(lo);
        uint4* d0 = reinterpret_cast<uint4*>(smem);
        uint4* d1 = reinterpret_cast<uint4*>(smem + img_b);
        for (int i = threadIdx.x; i < img_b / 16; i += kSplitThreads) {
            d0[i] = s0[i];
            d1[i] = s1[i];
        }
    };
    if constexpr (!FQ) {
        if (nkc == 1) fill_lds(0);
    }
    __syncthreads();

    const int64_t ntiles = (N + kTileRows - 1) / kTileRows;
    const char* a_base[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) a_base[t] = smem + min(t * 32 + j, rows - 1 + zrow) * row_b + kg * 64;
    // XR: lane (block lane / 4, i = lane % 4) supplies extra query row i (the zero row beyond the last one) as the A operand
    const char* xa_base = smem + min(32 * QT + (lane & 3), rows - 1 + zrow) * row_b + kg * 64;
    (void)xa_base;

    // work split: full rounds of interleaved 256-row tiles (all workgroups sweep one compact window of the map: measured
    // 1.2 % faster at 2 M voxels than one contiguous range per workgroup), then ONE tail round in which what is left
    // (< gridDim.x tiles) is dealt out in 32-row units, balanced over all workgroups -- every CU stays busy with a partial
    // load instead of a third of the chip idling (300 k voxels = 4.58 rounds: -4 %; 200 k: -14 %)
    (void)ntiles;
    const int64_t G = gridDim.x;
    const int64_t R = N / (G * kTileRows);                       // complete interleaved rounds
    const int64_t tail_row0 = R * G * kTileRows;
    const int64_t tail_units = (N - tail_row0 + 31) / 32;        // <= 8 per workgroup
    const int64_t tu0 = tail_units * blockIdx.x / G, tu1 = tail_units * (blockIdx.x + 1) / G;
    for (int64_t it = 0; it <= R; ++it) {
        const bool tail = it == R;
        if (tail && tu0 == tu1) break;
        const bool active = !tail || tu0 + wave < tu1;
        if constexpr (FQ) {
            if (!active) continue;   // no barrier inside the tile loop when the whole image is resident
        }
        // inactive waves of a K-chunked launch (barriers in the loop) idle on row N-1 and write nothing
        // Complete rounds: workgroup b takes tile it * G + b -- all workgroups sweep one compact window of the map.  With a row stride
        // that is a multiple of 4 KiB (512 columns of a 1024- or 2048-float row) that window camps on a few HBM channels (0.79 / 0.77
        // ms against 0.72 contiguous, profiles/r04_row_stride_probe.txt): there a workgroup takes kRun CONSECUTIVE tiles of every
        // super-round of G * kRun tiles, as the K-swap kernel does, so that at any moment the workgroups are spread over the map.
        constexpr int64_t kRun = 4;
#ifndef AVL_SPREAD_MODE
#define AVL_SPREAD_MODE 0
#endif
        const bool spread = !P24 && (AVL_SPREAD_MODE == 2 || (ld != D && (AVL_SPREAD_MODE == 1 || ((ld * 4) & 4095) == 0))) && it < (R / kRun) * kRun;
        const int64_t tile = spread ? (it / kRun) * (G * kRun) + (int64_t)blockIdx.x * kRun + (it % kRun) : it * G + blockIdx.x;
        const int64_t row = !active ? N : (tail ? tail_row0 + (tu0 + wave) * 32 + j : tile * kTileRows + wave * 32 + j);
        const int64_t rowc = row < N ? row : N - 1;
        const float* rp = feat + rowc * ld + 32 * kg;
        const char* rp24 = reinterpret_cast<const char*>(feat) + rowc * (ld * 3);   // P24: first byte of the row, ld = columns of a full row

        // NA independent accumulator sets per query tile (hi*hi | cross terms) so that back-to-back MFMAs never wait on
        // each other's result: a dependent 32x32x16 MFMA cannot issue until its predecessor retires
        f32x16 acc[QT][NA];
#pragma unroll
        for (int t = 0; t < QT; ++t)
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[t][a][e] = 0.f;
        float rmax = 0.f;   // !PRE: largest |element| this lane has loaded of its row (range guard)
        f32x4 xacc0 = {0.f, 0.f, 0.f, 0.f}, xacc1 = {0.f, 0.f, 0.f, 0.f};   // XR: two chains so that back-to-back 4x4x4 MFMAs alternate

        for (int kc = 0; kc < nkc; ++kc) {
            if (nkc > 1) {
                __syncthreads();
                fill_lds(kc);
                __syncthreads();
            }
            const int klen = min(KC, D - kc * KC);
            const int nsteps = klen >> 6;
            const float* p = rp + kc * KC;
            const int sa0 = (col0 + kc * KC) >> 6;          // P24: index of this chunk's first step in the row

            f32x4 buf0[8], buf1[8];
            auto load = [&](f32x4(&b)[8], int s) {
                if constexpr (P24) {
                    compact_load_step(b, rp24, ld, kg, sa0 + s);
                } else {
                    const f32x4* g = reinterpret_cast<const f32x4*>(p + 64 * s);
#pragma unroll
                    for (int t = 0; t < 8; ++t) b[t] = g[t];
                }
            };
            // operands(m, bh, bl): the voxel-side fragments of the m-th 8-column group of step s
            auto compute_with = [&](auto&& operands, int s) {
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    half8 bh, bl;
                    operands(m, bh, bl);
                    const int off = (s * 64 + 8 * m) * 2;
                    half8 ah[QT], al[QT];
#pragma unroll
                    for (int t = 0; t < QT; ++t) {
#ifdef AVL_ABL_NOLDS
                        for (int e = 0; e < 8; ++e) { ah[t][e] = (_Float16)1; al[t][e] = (_Float16)2; }
                        asm volatile("" : "+v"(ah[t]), "+v"(al[t]));
#else
                        ah[t] = *reinterpret_cast<const half8*>(a_base[t] + off);
                        al[t] = *reinterpret_cast<const half8*>(a_base[t] + off + img_b);
#endif
                    }
#ifdef AVL_ABL_NOMFMA
#pragma unroll
                    for (int t = 0; t < QT; ++t) asm volatile("" ::"v"(ah[t]), "v"(al[t]), "v"(bh), "v"(bl));
#else
                    // term-major, tile-minor issue order: consecutive MFMAs target different accumulators
#pragma unroll
                    for (int t = 0; t < QT; ++t) acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], bh, acc[t][0], 0, 0, 0);
#ifndef AVL_ABL_HIONLY   // ablation for the argmax-only question: what ONE product per pair would cost
#pragma unroll
                    for (int t = 0; t < QT; ++t) acc[t][X1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[t], bh, acc[t][X1], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < QT; ++t) acc[t][X2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], bl, acc[t][X2], 0, 0, 0);
#endif
#endif
                    if constexpr (XR) {
                        using half4 = __attribute__((ext_vector_type(4))) _Float16;
                        const half8 xh = *reinterpret_cast<const half8*>(xa_base + off);
                        const half8 xl = *reinterpret_cast<const half8*>(xa_base + off + img_b);
                        const half4 xh0 = __builtin_shufflevector(xh, xh, 0, 1, 2, 3), xh1 = __builtin_shufflevector(xh, xh, 4, 5, 6, 7);
                        const half4 xl0 = __builtin_shufflevector(xl, xl, 0, 1, 2, 3), xl1 = __builtin_shufflevector(xl, xl, 4, 5, 6, 7);
                        const half4 bh0 = __builtin_shufflevector(bh, bh, 0, 1, 2, 3), bh1 = __builtin_shufflevector(bh, bh, 4, 5, 6, 7);
                        const half4 bl0 = __builtin_shufflevector(bl, bl, 0, 1, 2, 3), bl1 = __builtin_shufflevector(bl, bl, 4, 5, 6, 7);
                        xacc0 = __builtin_amdgcn_mfma_f32_4x4x4f16(xh0, bh0, xacc0, 0, 0, 0);
                        xacc1 = __builtin_amdgcn_mfma_f32_4x4x4f16(xh1, bh1, xacc1, 0, 0, 0);
                        xacc0 = __builtin_amdgcn_mfma_f32_4x4x4f16(xl0, bh0, xacc0, 0, 0, 0);
                        xacc1 = __builtin_amdgcn_mfma_f32_4x4x4f16(xl1, bh1, xacc1, 0, 0, 0);
                        xacc0 = __builtin_amdgcn_mfma_f32_4x4x4f16(xh0, bl0, xacc0, 0, 0, 0);
                        xacc1 = __builtin_amdgcn_mfma_f32_4x4x4f16(xh1, bl1, xacc1, 0, 0, 0);
                    }
                }
            };
            auto compute = [&](const f32x4(&b)[8], int s) {
                compute_with([&](int m, half8& bh, half8& bl) { map_operands<PRE, P24>(b, m, bh, bl, rmax); }, s);
            };
            if constexpr (P24 && NSTEPS > 0) {
                // Compact map, compile-time trip count, window starting on a 128-column block (the launcher checks): the residual line
                // of a block serves its two steps and is requested ONCE, with the even step -- 64 B per lane half into a buffer of its
                // own (two, by block parity: the odd step still reads one while the next block's is in flight); hi: two buffers by
                // step parity.  One step ahead, like the ring below.
                static_assert(NSTEPS % 2 == 0, "whole 128-column blocks");
                f32x4 hb[2][4], lb[2][4];
                auto load_hi = [&](int st) {
                    const f32x4* g = reinterpret_cast<const f32x4*>(rp24 + 128 * (sa0 + st) + 64 * kg);
#pragma unroll
                    for (int t = 0; t < 4; ++t) hb[st & 1][t] = g[t];
                };
                auto load_lo = [&](int blk) {
                    const f32x4* g = reinterpret_cast<const f32x4*>(rp24 + 2 * ld + 128 * ((sa0 >> 1) + blk) + 64 * kg);
#pragma unroll
                    for (int t = 0; t < 4; ++t) lb[blk & 1][t] = g[t];
                };
                load_hi(0);
                load_lo(0);
#pragma unroll
                for (int st = 0; st < NSTEPS; ++st) {
                    if (st + 1 < NSTEPS) {
                        load_hi(st + 1);
                        if (((st + 1) & 1) == 0) load_lo((st + 1) >> 1);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    compute_with([&](int m, half8& bh, half8& bl) {
                        compact_operands(hb[st & 1][m], lb[(st >> 1) & 1][2 * (st & 1) + (m >> 1)], m & 1, bh, bl);
                    }, st);
                }
            } else if constexpr (NSTEPS > 0) {
                // compile-time trip count, ring of kRing register buffers: the loads of the next kRing - 1 steps are in flight
                // while step s is computed, all waits are counted vmcnt (depth per variant: RingDepth)
                constexpr int kRing = RingDepth<PRE, XR, QM>::value;
                f32x4 ring[kRing][8];
#pragma unroll
                for (int r = 0; r + 1 < kRing; ++r)
                    if (r < NSTEPS) load(ring[r], r);
#pragma unroll
                for (int s = 0; s < NSTEPS; ++s) {
                    if (s + kRing - 1 < NSTEPS) load(ring[(s + kRing - 1) % kRing], s + kRing - 1);
                    __builtin_amdgcn_sched_barrier(0);   // keep the prefetch block ahead of the compute block
                    compute(ring[s % kRing], s);
                }
            } else {
                load(buf0, 0);
                for (int s = 0; s < nsteps; s += 2) {
                    if (s + 1 < nsteps) load(buf1, s + 1);
                    compute(buf0, s);
                    if (s + 1 < nsteps) {
                        if (s + 2 < nsteps) load(buf0, s + 2);
                        compute(buf1, s + 1);
                    }
                }
            }
        }

        float rscale = 1.f;
        if constexpr (PRE) {
            if (row_scale) rscale = row_scale[rowc];
        }
        // QM (column-block launches) is a template parameter so that the dense kernels keep their register budget
        split_epilogue<QT, NA, XR>(acc, isc, q_base, rows, Q, scores, argmax, best, row, N, kg, first_chunk, rscale, PRE ? nullptr : flags, rmax,
                                   QM ? qml : nullptr, xacc0 + xacc1);
    }
}

// ------------------------------------------------------------------------------------------------
// K-swap variant of the resident kernel for feature widths of TWO LDS-sized K chunks (512 < D <= 1024, <= 64 query rows: the
// 1024-column audio block of BASELINE config 5, CLIP ViT-L widths).  The query image of ONE chunk (hi | lo, 133 KB at 64 x 512) is
// resident; a workgroup carries T voxel tiles through it with their accumulators in registers (16 QT VGPRs per tile), swaps the
// image for the other chunk, finishes the T tiles, and swaps back for its next T tiles: two LDS refills of half the image per T
// tiles, against a refill of the whole image per TB = 2 tiles in the streamed kernel, whose per-chunk staging traffic and
// barriers were 9 % of its time (profiles/HISTORY.md 4.1c).  A voxel row is read in two visits of 2 KB.
// image layout = the resident kernel's [kc][hi, lo][Qtot][KC + pad] (sim_prep_queries_kernel, interleaved = 0).
// ------------------------------------------------------------------------------------------------
template <int QT, int T, int NS, bool PRE, bool QM = false, bool P24 = false>
__global__ __launch_bounds__(kSplitThreads) void sim_kswap_f16_kernel(
    const float* __restrict__ feat, int64_t N, int D, int64_t ld, const _Float16* __restrict__ img,
    const float* __restrict__ inv_scale, int Qtot, int KC, int q_base, int rows, int Q, float* __restrict__ scores,
    int32_t* __restrict__ argmax, float* __restrict__ best, int first_chunk, const float* __restrict__ row_scale,
    uint32_t* __restrict__ flags, const int32_t* __restrict__ qmap, int col0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, kg = lane >> 5;
    const int row_b = (KC + kRowPadHalves) * 2;
    const int zrow = rows < 32 * QT ? 1 : 0;
    const int img_b = (rows + zrow) * row_b;
    float* isc = reinterpret_cast<float*>(smem + 2 * img_b);
    int32_t* qml = reinterpret_cast<int32_t*>(isc + QT * 32);
    if constexpr (QM) {
        if (threadIdx.x < QT * 32) qml[threadIdx.x] = threadIdx.x < rows ? qmap[q_base + threadIdx.x] : INT_MAX;
    }
    if (threadIdx.x < QT * 32) isc[threadIdx.x] = threadIdx.x < rows ? inv_scale[q_base + threadIdx.x] : 0.f;
    // the image of chunk kc -> LDS with direct global-to-LDS loads (global_load_lds_dwordx4: a wave instruction lands 64 x 16 B
    // at a wave-uniform LDS address; no staging registers -- the T tiles' accumulators are live across a swap -- and all of a
    // wave's pieces are in flight at once instead of a chain of load/store round trips)
    auto fill_lds = [&](int kc) {
        using gptr = const __attribute__((address_space(1))) void*;
        using lptr = __attribute__((address_space(3))) void*;
        const char* hi = reinterpret_cast<const char*>(img) + ((int64_t)(kc * 2) * Qtot + q_base) * row_b;
        const char* lo = hi + (int64_t)Qtot * row_b;
        const int nb = rows * row_b;                       // bytes per half, a multiple of 16
        for (int c = wave * 1024; c < nb; c += (kSplitThreads / 64) * 1024) {
            if (c + lane * 16 < nb) {
                __builtin_amdgcn_global_load_lds((gptr)(hi + c + lane * 16), (lptr)(smem + c), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gptr)(lo + c + lane * 16), (lptr)(smem + img_b + c), 16, 0, 0);
            }
        }
    };
    auto fill_wait = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
    if (zrow)
        for (int i = threadIdx.x; i < row_b / 4; i += kSplitThreads) {
            reinterpret_cast<uint32_t*>(smem + rows * row_b)[i] = 0u;
            reinterpret_cast<uint32_t*>(smem + img_b + rows * row_b)[i] = 0u;
        }
    fill_lds(0);
    fill_wait();
    __syncthreads();

    const char* a_base[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) a_base[t] = smem + min(t * 32 + j, rows - 1 + zrow) * row_b + kg * 64;

    // work split as in the resident kernel: full rounds of interleaved 256-row tiles, then one tail round in 32-row units
    const int64_t G = gridDim.x;
    const int64_t R = N / (G * kTileRows);
    const int64_t tail_row0 = R * G * kTileRows;
    const int64_t tail_units = (N - tail_row0 + 31) / 32;
    const int64_t tu0 = tail_units * blockIdx.x / G, tu1 = tail_units * (blockIdx.x + 1) / G;
    const int64_t n_it = R + (tu0 < tu1 ? 1 : 0);                  // iterations of this workgroup (workgroup-uniform)
    constexpr int kElB = P24 ? 3 : 4, kLaneB = P24 ? 96 : 128, kStepB = 2 * kLaneB, kLoads = P24 ? 6 : 8;

    for (int64_t it0 = 0; it0 < n_it; it0 += T) {
        // slot b of this group = iteration it0 + b of the workgroup; its row / activity are recomputed where they are needed (three
        // places) instead of being kept in registers next to the T accumulator sets
        // a group's T full rounds cover the tiles [it0 G, (it0 + T) G): this workgroup takes T CONSECUTIVE ones of them rather than
        // one of every round (same-box, 2 M voxels: D = 1024 dense, 4 KiB row stride, 1.603 -> 1.510 ms; config 5 unchanged / -1.6 %);
        // a last, shorter group keeps one tile per round
        const bool full_group = it0 + T <= R;
        const int64_t tile_first = full_group ? it0 * G + (int64_t)blockIdx.x * T : it0 * G + blockIdx.x;
        const int64_t tile_step = full_group ? 1 : G;
        auto slot_row = [&](int b, bool& act) -> int64_t {
            const int64_t it = it0 + b;
            const bool tail = it == R;
            act = it < n_it && (!tail || tu0 + wave < tu1);
            return !act ? N : (tail ? tail_row0 + (tu0 + wave) * 32 + j : (tile_first + b * tile_step) * kTileRows + wave * 32 + j);
        };
        // every row is contracted chunk 0 first, then chunk 1 -- whatever tile group it falls into -- so that its scores do not
        // depend on where it sits in the launch (a band of rows == the same rows of the whole map, bit for bit)
        if (it0 > 0) {                          // chunk 0 was requested before the previous group's epilogue (below)
            fill_wait();
            __syncthreads();
        }
        f32x16 acc[T][QT][1];
        float rmax[T];
#pragma unroll
        for (int b = 0; b < T; ++b) {
            rmax[b] = 0.f;
#pragma unroll
            for (int t = 0; t < QT; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[b][t][0][e] = 0.f;
        }
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            if (pass == 1) {
                __syncthreads();                // every wave is done with chunk 0
                fill_lds(1);
                fill_wait();
                __syncthreads();
            }
            const int klen = min(KC, D - pass * KC);
            const int nsteps = NS > 0 ? NS : (klen >> 6);
#pragma unroll
            for (int b = 0; b < T; ++b) {
                bool act;
                const int64_t row = slot_row(b, act);
                if (!act) continue;             // wave-uniform; idle slots are the last ones of a workgroup's last group
                // P24: p = first byte of the (compact, two-plane) row, sa0 = index of the chunk's first 64-column step in the row
                const char* p = reinterpret_cast<const char*>(feat) + (row < N ? row : N - 1) * (ld * kElB) +
                                (P24 ? 0 : kLaneB * kg + (int64_t)pass * KC * kElB);
                const int sa0 = (col0 + pass * KC) >> 6;
                auto load = [&](f32x4(&buf)[8], int s) {
                    if constexpr (P24) {
                        compact_load_step(buf, p, ld, kg, sa0 + s);
                    } else {
                        const f32x4* g = reinterpret_cast<const f32x4*>(p + kStepB * s);
#pragma unroll
                        for (int t = 0; t < kLoads; ++t) buf[t] = g[t];
                    }
                };
                auto compute_with = [&](auto&& operands, int s) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        half8 bh, bl;
                        operands(m, bh, bl);
                        const int off = (s * 64 + 8 * m) * 2;
                        half8 ah[QT], al[QT];
#pragma unroll
                        for (int t = 0; t < QT; ++t) {
                            ah[t] = *reinterpret_cast<const half8*>(a_base[t] + off);
                            al[t] = *reinterpret_cast<const half8*>(a_base[t] + off + img_b);
                        }
#pragma unroll
                        for (int t = 0; t < QT; ++t) acc[b][t][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], bh, acc[b][t][0], 0, 0, 0);
#pragma unroll
                        for (int t = 0; t < QT; ++t) acc[b][t][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[t], bh, acc[b][t][0], 0, 0, 0);
#pragma unroll
                        for (int t = 0; t < QT; ++t) acc[b][t][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], bl, acc[b][t][0], 0, 0, 0);
                    }
                };
                auto compute = [&](const f32x4(&buf)[8], int s) {
                    compute_with([&](int m, half8& bh, half8& bl) { map_operands<PRE, P24>(buf, m, bh, bl, rmax[b]); }, s);
                };
                if constexpr (NS > 0 && P24) {
                    // compact map, window on a 128-column block: the residual line of a block is requested once, with its even step
                    // (see the resident kernel)
                    static_assert(NS % 2 == 0, "whole 128-column blocks");
                    f32x4 hb[2][4], lb[2][4];
                    auto load_hi = [&](int st) {
                        const f32x4* g = reinterpret_cast<const f32x4*>(p + 128 * (sa0 + st) + 64 * kg);
#pragma unroll
                        for (int t = 0; t < 4; ++t) hb[st & 1][t] = g[t];
                    };
                    auto load_lo = [&](int blk) {
                        const f32x4* g = reinterpret_cast<const f32x4*>(p + 2 * ld + 128 * ((sa0 >> 1) + blk) + 64 * kg);
#pragma unroll
                        for (int t = 0; t < 4; ++t) lb[blk & 1][t] = g[t];
                    };
                    load_hi(0);
                    load_lo(0);
#pragma unroll
                    for (int st = 0; st < NS; ++st) {
                        if (st + 1 < NS) {
                            load_hi(st + 1);
                            if (((st + 1) & 1) == 0) load_lo((st + 1) >> 1);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        compute_with([&](int m, half8& bh, half8& bl) {
                            compact_operands(hb[st & 1][m], lb[(st >> 1) & 1][2 * (st & 1) + (m >> 1)], m & 1, bh, bl);
                        }, st);
                    }
                } else if constexpr (NS > 0) {
                    // (carrying the register buffer of a slot's first step over from the previous slot's last step, and across the
                    // image swap, was measured: no gain on prepared / compact maps -- 1.635 vs 1.621 ms, 1.258 vs 1.239 ms at
                    // 2 M x 1024 x 64 -- and 25-29 spills on raw ones; every slot restarts its two-buffer ring)
                    f32x4 ring[2][8];
                    load(ring[0], 0);
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        if (s + 1 < NS) load(ring[(s + 1) & 1], s + 1);
                        __builtin_amdgcn_sched_barrier(0);
                        compute(ring[s & 1], s);
                    }
                } else {
                    f32x4 buf0[8], buf1[8];
                    load(buf0, 0);
                    for (int s = 0; s < nsteps; s += 2) {
                        if (s + 1 < nsteps) load(buf1, s + 1);
                        compute(buf0, s);
                        if (s + 1 < nsteps) {
                            if (s + 2 < nsteps) load(buf0, s + 2);
                            compute(buf1, s + 1);
                        }
                    }
                }
            }
        }
        // the swap back to chunk 0 for the next group travels while this group's scores are reduced and stored (the epilogue reads the
        // per-query tables behind the images, not the images)
        if (it0 + T < n_it) {
            __syncthreads();                    // every wave is done with chunk 1
            fill_lds(0);
#ifdef AVL_ABL_KSWAP_SYNCSWAP
            fill_wait();
#endif
        }
#pragma unroll
        for (int b = 0; b < T; ++b) {
            bool act;
            const int64_t row = slot_row(b, act);
            if (!act) continue;
            float rscale = 1.f;
            if constexpr (PRE) {
                if (row_scale) rscale = row_scale[row < N ? row : N - 1];
            }
            split_epilogue<QT, 1>(acc[b], isc, q_base, rows, Q, scores, argmax, best, row, N, kg, first_chunk, rscale, PRE ? nullptr : flags,
                                  rmax[b], QM ? qml : nullptr);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// streaming variant of the split kernel for query sets that do not fit LDS whole (Q > 78 at D = 512, or D > 512):
// ONE pass over the map for up to 128 queries (QT <= 4 MFMA tiles, one accumulator set each).  The query image streams
// through LDS in K chunks of KS = 64 * SPC columns, double buffered: while the eight waves contract chunk c out of one
// buffer, every thread stages 16-byte pieces of chunk c+1 (L2-resident, identical for every tile, cyclic) in registers and
// drops them into the other buffer at the end of the chunk; one raw s_barrier per chunk (the LDS writes are drained with
// lgkmcnt only, so the voxel prefetch stays in flight across it).  Voxel rows come straight from HBM exactly as in the
// resident kernel (lane (voxel j, half kg) walks one 128-byte line per 64-wide k step, 2 register buffers), and the last
// step of a tile prefetches step 0 of the workgroup's next tile.
// image layout: [nch][Qtot][hi KS | lo KS | pad 8] fp16 (sim_prep_queries_kernel, interleaved) -- a chunk's rows are one
// contiguous block, copied linearly; the 16-byte pad keeps ds_read_b128 of 32 consecutive rows conflict-free
// ------------------------------------------------------------------------------------------------
constexpr int kStreamFill = 9;   // 16-byte staging registers per thread: one LDS buffer <= 9 * 512 * 16 B = 72 KB

template <int QT, int SPC, bool PRE, bool QM = false, bool P24 = false>
__global__ __launch_bounds__(kSplitThreads) void sim_stream_f16_kernel(
    const float* __restrict__ feat, int64_t N, int D, int64_t ld, const _Float16* __restrict__ img,
    const float* __restrict__ inv_scale, int Qtot, int nch, int q_base, int rows, int Q, float* __restrict__ scores,
    int32_t* __restrict__ argmax, float* __restrict__ best, int first_chunk, const float* __restrict__ row_scale,
    uint32_t* __restrict__ flags, const int32_t* __restrict__ qmap, int col0) {
    static_assert(SPC % 2 == 0, "the register buffer of a chunk's first step must not move between chunks");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KS = 64 * SPC;
    constexpr int row_b = (2 * KS + kRowPadHalves) * 2;   // bytes per query row of one chunk: hi[KS] | lo[KS] | pad
    constexpr int lo_b = 2 * KS;                          // byte offset of the lo half inside a row
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, kg = lane >> 5;
    // a partial last MFMA tile reads one extra all-zero row per buffer instead of re-reading a valid row: zeros toggle far
    // fewer matrix-core bits, and at the power cap that is time
    const int zrow = rows < 32 * QT ? 1 : 0;
    const int buf_b = (rows + zrow) * row_b;              // one LDS buffer = the chunk's rows, a linear copy of the image
    float* isc = reinterpret_cast<float*>(smem + 2 * buf_b);
    if (threadIdx.x < QT * 32) isc[threadIdx.x] = threadIdx.x < rows ? inv_scale[q_base + threadIdx.x] : 0.f;
    int32_t* qml = reinterpret_cast<int32_t*>(isc + QT * 32);   // QM: the caller's query index of every row of this chunk
    if constexpr (QM) {
        if (threadIdx.x < QT * 32) qml[threadIdx.x] = threadIdx.x < rows ? qmap[q_base + threadIdx.x] : INT_MAX;
    }
    const int units = (rows * row_b) >> 4;
    if (zrow)
        for (int i = threadIdx.x; i < row_b / 4; i += kSplitThreads) {
            reinterpret_cast<uint32_t*>(smem + rows * row_b)[i] = 0u;
            reinterpret_cast<uint32_t*>(smem + buf_b + rows * row_b)[i] = 0u;
        }

    // the next chunk is staged in SPC slices, one per k step (registers: 5 x 16 B at SPC = 2, 3 x 16 B at SPC = 4)
    constexpr int NSTG = kStreamFill - ((SPC - 1) * kStreamFill) / SPC;
    f32x4 stg[NSTG];
    auto stage_load = [&](int c, int i0, int i1) {
        const f32x4* src = reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(img) + ((int64_t)c * Qtot + q_base) * row_b);
#pragma unroll
        for (int i = i0; i < i1; ++i) {
            const int u = threadIdx.x + i * kSplitThreads;
            if (u < units) stg[i - i0] = src[u];
        }
    };
    auto stage_store = [&](int b, int i0, int i1) {
        f32x4* dst = reinterpret_cast<f32x4*>(smem + b * buf_b);
#pragma unroll
        for (int i = i0; i < i1; ++i) {
            const int u = threadIdx.x + i * kSplitThreads;
            if (u < units) dst[u] = stg[i - i0];
        }
    };

    const int64_t ntiles = (N + kTileRows - 1) / kTileRows;
    // byte geometry of a lane's walk: float32 / prepared rows are 4 B per column and a lane owns 128 B of every 64-column step;
    // the compact form (P24) is 3 B per column, 96 B per lane and step (hi[32] | residual bytes[32])
    constexpr int kElB = P24 ? 3 : 4, kLaneB = P24 ? 96 : 128, kStepB = 2 * kLaneB, kLoads = P24 ? 6 : 8;
    auto tile_ptr = [&](int64_t tile) {
        const int64_t r = tile * kTileRows + wave * 32 + j;
        return reinterpret_cast<const char*>(feat) + (r < N ? r : N - 1) * (ld * kElB) + (P24 ? 0 : kLaneB * kg);   // P24: first byte of the row
    };
    f32x4 ring[2][8];
    auto load = [&](f32x4(&b)[8], const char* src, int step) {     // step `step` of the window whose row (lane line) starts at src
        if constexpr (P24) {
            compact_load_step(b, src, ld, kg, (col0 >> 6) + step);
        } else {
            const f32x4* g = reinterpret_cast<const f32x4*>(src + kStepB * step);
#pragma unroll
            for (int t = 0; t < kLoads; ++t) b[t] = g[t];
        }
    };
    load(ring[0], tile_ptr(blockIdx.x), 0);   // first tile, step 0: in flight while chunk 0 is brought in
#pragma unroll
    for (int s = 0; s < SPC; ++s) {
        stage_load(0, (s * kStreamFill) / SPC, ((s + 1) * kStreamFill) / SPC);
        stage_store(0, (s * kStreamFill) / SPC, ((s + 1) * kStreamFill) / SPC);
    }
    __syncthreads();
    int cur = 0;

    int a_off[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) a_off[t] = min(t * 32 + j, rows - 1 + zrow) * row_b + kg * 64;

    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t row = tile * kTileRows + wave * 32 + j;
        const char* rp = tile_ptr(tile);
        const bool has_next = tile + gridDim.x < ntiles;
        const char* p_next = has_next ? tile_ptr(tile + gridDim.x) : rp;

        f32x16 acc[QT][1];
#pragma unroll
        for (int t = 0; t < QT; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][0][e] = 0.f;
        float rmax = 0.f;

        for (int c = 0; c < nch; ++c) {
            const int cn = c + 1 < nch ? c + 1 : 0;
            const char* ab = smem + cur * buf_b;
#pragma unroll
            for (int s = 0; s < SPC; ++s) {
                const int i0 = (s * kStreamFill) / SPC, i1 = ((s + 1) * kStreamFill) / SPC;
#ifndef AVL_ABL_NOSTAGE
                stage_load(cn, i0, i1);               // issued ahead of this step's voxel prefetch (older in vmcnt order)
#endif
                if (s + 1 < SPC) {
                    load(ring[(s + 1) & 1], rp, c * SPC + s + 1);
                } else if (c + 1 < nch) {
                    load(ring[0], rp, c * SPC + s + 1);
                } else if (has_next) {
                    load(ring[0], p_next, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                const f32x4(&b)[8] = ring[s & 1];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    half8 bh, bl;
                    map_operands<PRE, P24>(b, m, bh, bl, rmax);
                    const int off = (s * 64 + 8 * m) * 2;
                    if constexpr (QT == 4) {
                        // four tiles: the hi fragments serve both of their products before the lo fragments are fetched, so only
                        // four operand fragments are live at a time (the term-major order below spilled 5 VGPRs here)
                        half8 af[QT];
#pragma unroll
                        for (int t = 0; t < QT; ++t) af[t] = *reinterpret_cast<const half8*>(ab + a_off[t] + off);
#pragma unroll
                        for (int t = 0; t < QT; ++t) acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[t], bh, acc[t][0], 0, 0, 0);
#pragma unroll
                        for (int t = 0; t < QT; ++t) acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[t], bl, acc[t][0], 0, 0, 0);
#pragma unroll
                        for (int t = 0; t < QT; ++t) af[t] = *reinterpret_cast<const half8*>(ab + a_off[t] + off + lo_b);
#pragma unroll
                        for (int t = 0; t < QT; ++t) acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[t], bh, acc[t][0], 0, 0, 0);
                    } else {
                    half8 ah[QT], al[QT];
#pragma unroll
                    for (int t = 0; t < QT; ++t) {
                        ah[t] = *reinterpret_cast<const half8*>(ab + a_off[t] + off);
                        al[t] = *reinterpret_cast<const half8*>(ab + a_off[t] + off + lo_b);
                    }
#pragma unroll
                    for (int t = 0; t < QT; ++t) acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], bh, acc[t][0], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < QT; ++t) acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[t], bh, acc[t][0], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < QT; ++t) acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], bl, acc[t][0], 0, 0, 0);
                    }
                }
#ifndef AVL_ABL_NOSTAGE
                stage_store(cur ^ 1, i0, i1);
#endif
            }
            // publish the next chunk / retire this one: LDS traffic only, the voxel prefetch stays in flight
#ifndef AVL_ABL_NOBARRIER
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
#endif
            cur ^= 1;
        }
        float rscale = 1.f;
        if constexpr (PRE) {
            if (row_scale) rscale = row_scale[row < N ? row : N - 1];
        }
        split_epilogue<QT, 1>(acc, isc, q_base, rows, Q, scores, argmax, best, row, N, kg, first_chunk, rscale, PRE ? nullptr : flags, rmax,
                              QM ? qml : nullptr);
    }
}

// Tile-blocked variant of the streamed kernel for QT <= 2.  TB = voxel tiles a workgroup carries through the chunk loop
// together: a chunk's LDS residency -- its staging traffic from L2 and the s_barrier that retires it, 11 % + 7 % of the kernel
// at one tile per chunk pass (profiles/HISTORY.md 4.1c) -- is paid once per TB tiles: the chunk's columns of tile 0, then of tile 1, ... are
// contracted against the same buffer into TB accumulator sets (16 * QT registers per extra tile: TB = 2 for QT = 2, 3 spills).
// Same-box A/B (2 M voxels): D = 1536, Q = 64: 2.40 -> 2.25 ms; config 5's column-block launches 2.51 -> 2.42 ms; with a
// power-of-two row stride (D = 1024: 4 KiB) the two-tile walk is 5 % SLOWER (HBM channel camping), and for QT >= 3 the extra
// bookkeeping costs 1-4 %, so the launcher only picks it for QT = 2 and row strides that are not a multiple of 4 KiB.
#ifndef AVL_TB2
#define AVL_TB2 2
#endif
#ifndef AVL_TB2_P24
#define AVL_TB2_P24 3
#endif
// (prepared and compact maps need no split / guard registers: three tiles per chunk pass fit without a spill, 245-255 VGPRs)
template <int QT, bool P24, bool PRE = false>
struct StreamTB {
    static constexpr int value = QT == 1 ? 4 : (QT == 2 ? ((P24 || PRE) ? AVL_TB2_P24 : AVL_TB2) : 1);
};

// (Tried and dropped, round 2: ONE pass over several column windows -- the chunk loop switching query rows / running the
// epilogue at window ends, so that config 5 reads whole 6 KB rows once instead of 2 KB + 4 KB pieces in two launches.  Same-box
// A/B at 2 M x 1536: 2.40 / 2.37 / 2.41 ms against 2.39 / 2.39 / 2.43 ms for the two launches (Q = 64 and 96: the same within
// 1 %).  The kernel is bound by the package power the three MFMAs per product draw, not by how the rows are walked.)
// (NT: threads per workgroup.  Two co-resident 256-thread workgroups per CU with 128-column chunks -- so that one computes
// while the other sits in its barrier -- measured 7.5 % SLOWER than one 512-thread workgroup with 256-column chunks: config 5
// 2.54 vs 2.36 ms on one box; only the 512-thread form is launched.)
template <int QT, int SPC, bool PRE, bool QM = false, bool P24 = false, int NT = kSplitThreads>
__global__ __launch_bounds__(NT) void sim_stream_tb_f16_kernel(
    const float* __restrict__ feat, int64_t N, int D, int64_t ld, const _Float16* __restrict__ img,
    const float* __restrict__ inv_scale, int Qtot, int nch, int q_base, int rows, int Q, float* __restrict__ scores,
    int32_t* __restrict__ argmax, float* __restrict__ best, int first_chunk, const float* __restrict__ row_scale,
    uint32_t* __restrict__ flags, const int32_t* __restrict__ qmap, int col0) {
    static_assert(SPC % 2 == 0, "the register buffer of a chunk's first step must not move between chunks");
    constexpr int TB = StreamTB<QT, P24, PRE>::value;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KS = 64 * SPC;
    constexpr int row_b = (2 * KS + kRowPadHalves) * 2;   // bytes per query row of one chunk: hi[KS] | lo[KS] | pad
    constexpr int lo_b = 2 * KS;                          // byte offset of the lo half inside a row
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, kg = lane >> 5;
    // a partial last MFMA tile reads one extra all-zero row per buffer instead of re-reading a valid row: zeros toggle far
    // fewer matrix-core bits, and at the power cap that is time
    const int zrow = rows < 32 * QT ? 1 : 0;
    const int buf_b = (rows + zrow) * row_b;              // one LDS buffer = the chunk's rows, a linear copy of the image
    float* isc = reinterpret_cast<float*>(smem + 2 * buf_b);
    if (threadIdx.x < QT * 32) isc[threadIdx.x] = threadIdx.x < rows ? inv_scale[q_base + threadIdx.x] : 0.f;
    int32_t* qml = reinterpret_cast<int32_t*>(isc + QT * 32);   // QM: the caller's query index of every row of this chunk
    if constexpr (QM) {
        if (threadIdx.x < QT * 32) qml[threadIdx.x] = threadIdx.x < rows ? qmap[q_base + threadIdx.x] : INT_MAX;
    }
    const int units = (rows * row_b) >> 4;
    if (zrow)
        for (int i = threadIdx.x; i < row_b / 4; i += NT) {
            reinterpret_cast<uint32_t*>(smem + rows * row_b)[i] = 0u;
            reinterpret_cast<uint32_t*>(smem + buf_b + rows * row_b)[i] = 0u;
        }

    // the next chunk is staged in SPC slices, one per k step of the block's FIRST tile (registers: 5 x 16 B at SPC = 2,
    // 3 x 16 B at SPC = 4); the other tiles of the block only compute
    constexpr int NSTG = kStreamFill - ((SPC - 1) * kStreamFill) / SPC;
    f32x4 stg[NSTG];
    auto stage_load = [&](int c, int i0, int i1) {
        const f32x4* src = reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(img) + ((int64_t)c * Qtot + q_base) * row_b);
#pragma unroll
        for (int i = i0; i < i1; ++i) {
            const int u = threadIdx.x + i * NT;
            if (u < units) stg[i - i0] = src[u];
        }
    };
    auto stage_store = [&](int b, int i0, int i1) {
        f32x4* dst = reinterpret_cast<f32x4*>(smem + b * buf_b);
#pragma unroll
        for (int i = i0; i < i1; ++i) {
            const int u = threadIdx.x + i * NT;
            if (u < units) dst[u] = stg[i - i0];
        }
    };

    // work split: full rounds of interleaved TB-tile blocks (all workgroups sweep one compact window of the map), then what
    // is left (< gridDim.x * TB tiles) dealt out evenly, at most TB tiles per workgroup
    const int64_t ntiles = (N + (NT / 64 * 32) - 1) / (NT / 64 * 32);
    const int64_t G = gridDim.x;
    const int64_t R = ntiles / (G * TB);
    const int64_t rem0 = R * G * TB, rem = ntiles - rem0;
    const int64_t ra = rem * blockIdx.x / G, rb = rem * (blockIdx.x + 1) / G;
    auto block_of = [&](int64_t it, int64_t& tile0, int& cnt) {
        if (it < R) {
            tile0 = (it * G + blockIdx.x) * TB;
            cnt = TB;
        } else {
            tile0 = rem0 + ra;
            cnt = it == R ? (int)(rb - ra) : 0;
        }
    };
    constexpr int kElB = P24 ? 3 : 4, kLaneB = P24 ? 96 : 128, kStepB = 2 * kLaneB, kLoads = P24 ? 6 : 8;   // see sim_stream_f16_kernel
    auto tile_ptr = [&](int64_t tile) {
        const int64_t r = tile * (NT / 64 * 32) + wave * 32 + j;
        return reinterpret_cast<const char*>(feat) + (r < N ? r : N - 1) * (ld * kElB) + (P24 ? 0 : kLaneB * kg);   // P24: first byte of the row
    };
    f32x4 ring[2][8];
    auto load = [&](f32x4(&b)[8], const char* src, int step) {     // step `step` of the window whose row (lane line) starts at src
        if constexpr (P24) {
            compact_load_step(b, src, ld, kg, (col0 >> 6) + step);
        } else {
            const f32x4* g = reinterpret_cast<const f32x4*>(src + kStepB * step);
#pragma unroll
            for (int t = 0; t < kLoads; ++t) b[t] = g[t];
        }
    };
    int64_t tile0;
    int cnt;
    block_of(0, tile0, cnt);
    if (cnt == 0) return;   // kernel-uniform per workgroup, before any barrier
    load(ring[0], tile_ptr(tile0), 0);   // first tile, step 0: in flight while chunk 0 is brought in
#pragma unroll
    for (int s = 0; s < SPC; ++s) {
        stage_load(0, (s * kStreamFill) / SPC, ((s + 1) * kStreamFill) / SPC);
        stage_store(0, (s * kStreamFill) / SPC, ((s + 1) * kStreamFill) / SPC);
    }
    __syncthreads();
    int cur = 0;

    int a_off[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) a_off[t] = min(t * 32 + j, rows - 1 + zrow) * row_b + kg * 64;

    for (int64_t it = 0; cnt > 0; ++it) {
        int64_t tile0_next;
        int cnt_next;
        block_of(it + 1, tile0_next, cnt_next);
        const char* rp[TB];
#pragma unroll
        for (int b = 0; b < TB; ++b) rp[b] = tile_ptr(tile0 + (b < cnt ? b : 0));
        const char* p_next = cnt_next > 0 ? tile_ptr(tile0_next) : rp[0];

        f32x16 acc[TB][QT][1];
        float rmax[TB];
#pragma unroll
        for (int b = 0; b < TB; ++b) {
            rmax[b] = 0.f;
#pragma unroll
            for (int t = 0; t < QT; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[b][t][0][e] = 0.f;
        }

        for (int c = 0; c < nch; ++c) {
            const int cn = c + 1 < nch ? c + 1 : 0;
            const char* ab = smem + cur * buf_b;
#pragma unroll
            for (int b = 0; b < TB; ++b) {
                if (TB == 1 || b < cnt) {   // workgroup-uniform
#pragma unroll
                    for (int s = 0; s < SPC; ++s) {
                        const int i0 = (s * kStreamFill) / SPC, i1 = ((s + 1) * kStreamFill) / SPC;
#ifndef AVL_ABL_NOSTAGE
                        if (b == 0) stage_load(cn, i0, i1);   // issued ahead of this step's voxel prefetch (older in vmcnt order)
#endif
                        // prefetch the next step in execution order: same tile, next tile of the block (same chunk), first tile
                        // of the next chunk, first tile of the workgroup's next block
                        // (the address is selected, the load itself is unconditional: a load inside a branch makes the compiler's
                        // s_waitcnt bookkeeping fall back to vmcnt(0); the very last step of a workgroup re-reads its own line)
                        if (s + 1 < SPC) {
                            load(ring[(s + 1) & 1], rp[b], c * SPC + s + 1);
                        } else {
                            const char* nxt = p_next;
                            int nstep = 0;
                            if (b + 1 < TB && b + 1 < cnt) { nxt = rp[b + 1 < TB ? b + 1 : b]; nstep = c * SPC; }
                            else if (c + 1 < nch) { nxt = rp[0]; nstep = (c + 1) * SPC; }
                            load(ring[0], nxt, nstep);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        const f32x4(&v)[8] = ring[s & 1];
#pragma unroll
                        for (int m = 0; m < 4; ++m) {
                            half8 bh, bl;
                            map_operands<PRE, P24>(v, m, bh, bl, rmax[b]);
                            const int off = (s * 64 + 8 * m) * 2;
                            half8 ah[QT], al[QT];
#pragma unroll
                            for (int t = 0; t < QT; ++t) {
                                ah[t] = *reinterpret_cast<const half8*>(ab + a_off[t] + off);
                                al[t] = *reinterpret_cast<const half8*>(ab + a_off[t] + off + lo_b);
                            }
#pragma unroll
                            for (int t = 0; t < QT; ++t) acc[b][t][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], bh, acc[b][t][0], 0, 0, 0);
#pragma unroll
                            for (int t = 0; t < QT; ++t) acc[b][t][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[t], bh, acc[b][t][0], 0, 0, 0);
#pragma unroll
                            for (int t = 0; t < QT; ++t) acc[b][t][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], bl, acc[b][t][0], 0, 0, 0);
                        }
#ifndef AVL_ABL_NOSTAGE
                        if (b == 0) stage_store(cur ^ 1, i0, i1);
#endif
                    }
                }
            }
            // publish the next chunk / retire this one: LDS traffic only, the voxel prefetch stays in flight
#ifndef AVL_ABL_NOBARRIER
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
#endif
            cur ^= 1;
        }
#pragma unroll
        for (int b = 0; b < TB; ++b) {
            if (TB == 1 || b < cnt) {
                const int64_t row = (tile0 + b) * (NT / 64 * 32) + wave * 32 + j;
                float rscale = 1.f;
                if constexpr (PRE) {
                    if (row_scale) rscale = row_scale[row < N ? row : N - 1];
                }
                split_epilogue<QT, 1>(acc[b], isc, q_base, rows, Q, scores, argmax, best, row, N, kg, first_chunk, rscale,
                                      PRE ? nullptr : flags, rmax[b], QM ? qml : nullptr);
            }
        }
        tile0 = tile0_next;
        cnt = cnt_next;
    }
}

// ------------------------------------------------------------------------------------------------
// exact fp32 on the matrix cores: v_mfma_f32_32x32x2_f32 is bit-for-bit a k-ordered fmaf chain (no reduced-precision
// operands) at the fp32 vector rate.  Same data flow as the split kernel (queries = MFMA rows from LDS, voxels = MFMA
// columns straight from HBM, lane (voxel j, half kg) owns a 128-byte line per 64-wide k step, fused argmax), one MFMA
// per lane-float: HBM-bound up to Q = 32, ~1 ms at Q = 64 (vs 11.5 ms for the vector-ALU kernel).
// image layout: [nkc][Qtot][KC + 4] float32 (row pad 16 B -> conflict-free ds_read_b128)
// ------------------------------------------------------------------------------------------------
constexpr int kRowPadF32 = 4;

__global__ __launch_bounds__(256) void sim_prep_queries_f32_kernel(const float* __restrict__ q, int Q, int D, int64_t ldq,
                                                                   float* __restrict__ img, int Qtot, int KC, int nkc) {
    const int qg = blockIdx.x;
    const int rowlen = KC + kRowPadF32;
    for (int kc = 0; kc < nkc; ++kc) {
        float* dst = img + ((int64_t)kc * Qtot + qg) * rowlen;
        for (int kk = threadIdx.x; kk < rowlen; kk += blockDim.x) {
            const int k = kc * KC + kk;
            dst[kk] = (qg < Q && kk < KC && k < D) ? q[(int64_t)qg * ldq + k] : 0.f;
        }
    }
}

template <int QT>
__global__ __launch_bounds__(kSplitThreads) void sim_mfma_f32_kernel(
    const float* __restrict__ feat, int64_t N, int D, int64_t ld, const float* __restrict__ img, int Qtot, int KC, int nkc,
    int q_base, int rows, int Q, float* __restrict__ scores, int32_t* __restrict__ argmax, float* __restrict__ best,
    int first_chunk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, kg = lane >> 5;
    const int row_b = (KC + kRowPadF32) * 4;
    const int img_b = rows * row_b;

    auto fill_lds = [&](int kc) {
        const uint4* src = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(img) + ((int64_t)kc * Qtot + q_base) * row_b);
        uint4* dst = reinterpret_cast<uint4*>(smem);
        for (int i = threadIdx.x; i < img_b / 16; i += kSplitThreads) dst[i] = src[i];
    };
    if (nkc == 1) fill_lds(0);
    __syncthreads();

    const int64_t ntiles = (N + kTileRows - 1) / kTileRows;
    const char* a_base[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) a_base[t] = smem + min(t * 32 + j, rows - 1) * row_b + kg * 128;

    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t row = tile * kTileRows + wave * 32 + j;
        const int64_t rowc = row < N ? row : N - 1;
        const float* rp = feat + rowc * ld + 32 * kg;
        f32x16 acc[QT];
#pragma unroll
        for (int t = 0; t < QT; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

        for (int kc = 0; kc < nkc; ++kc) {
            if (nkc > 1) {
                __syncthreads();
                fill_lds(kc);
                __syncthreads();
            }
            const int nsteps = min(KC, D - kc * KC) >> 6;
            const float* p = rp + kc * KC;
            f32x4 buf0[8], buf1[8];
            auto load = [&](f32x4(&b)[8], int s) {
                const f32x4* g = reinterpret_cast<const f32x4*>(p + 64 * s);
#pragma unroll
                for (int t = 0; t < 8; ++t) b[t] = g[t];
            };
            auto compute = [&](const f32x4(&b)[8], int s) {
#pragma unroll
                for (int m4 = 0; m4 < 8; ++m4) {
                    const float bv[4] = {b[m4].x, b[m4].y, b[m4].z, b[m4].w};
#pragma unroll
                    for (int t = 0; t < QT; ++t) {
                        const f32x4 av = *reinterpret_cast<const f32x4*>(a_base[t] + (s * 64 + 4 * m4) * 4);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv[0], acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv[1], acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv[2], acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv[3], acc[t], 0, 0, 0);
                    }
                }
            };
            load(buf0, 0);
            for (int s = 0; s < nsteps; s += 2) {
                if (s + 1 < nsteps) load(buf1, s + 1);
                compute(buf0, s);
                if (s + 1 < nsteps) {
                    if (s + 2 < nsteps) load(buf0, s + 2);
                    compute(buf1, s + 1);
                }
            }
        }

        const int qend = q_base + rows;
        float bv = -INFINITY;
        int bi = INT_MAX;
#pragma unroll
        for (int t = 0; t < QT; ++t) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int qg = q_base + t * 32 + 8 * g + 4 * kg;
                f32x4 v;
                v.x = acc[t][4 * g + 0]; v.y = acc[t][4 * g + 1]; v.z = acc[t][4 * g + 2]; v.w = acc[t][4 * g + 3];
                if (scores && row < N) {
                    float* sp = scores + row * (int64_t)Q + qg;
                    if (qg + 3 < qend && (Q & 3) == 0) {
                        *reinterpret_cast<f32x4*>(sp) = v;
                    } else {
                        if (qg + 0 < qend) sp[0] = v.x;
                        if (qg + 1 < qend) sp[1] = v.y;
                        if (qg + 2 < qend) sp[2] = v.z;
                        if (qg + 3 < qend) sp[3] = v.w;
                    }
                }
                if (qg + 0 < qend && v.x > bv) { bv = v.x; bi = qg + 0; }
                if (qg + 1 < qend && v.y > bv) { bv = v.y; bi = qg + 1; }
                if (qg + 2 < qend && v.z > bv) { bv = v.z; bi = qg + 2; }
                if (qg + 3 < qend && v.w > bv) { bv = v.w; bi = qg + 3; }
            }
        }
        if (argmax || best) {
            const float ov = __shfl_xor(bv, 32, 64);
            const int oi = __shfl_xor(bi, 32, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
            if (kg == 0 && row < N) {
                if (bi == INT_MAX) bi = q_base;
                if (!first_chunk) {
                    const float pv = best[row];
                    if (!(bv > pv)) { bv = pv; bi = argmax ? argmax[row] : bi; }
                }
                if (argmax) argmax[row] = bi;
                if (best) best[row] = bv;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// small utility kernels
// ------------------------------------------------------------------------------------------------
__global__ void mask_from_argmax_kernel(const int32_t* __restrict__ am, int64_t N, int32_t cat, uint8_t* __restrict__ mask) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x)
        mask[i] = am[i] == cat;
}

// the same mask, bit-packed: word w, bit i = (am[64 w + i] == cat) -- np.unpackbits(..., bitorder="little") order.  A wave reads 64
// consecutive indices and publishes its ballot: 250 KB instead of 8 MB cross PCIe for a 2 M-voxel map (VLMap.index_map)
__global__ __launch_bounds__(256) void mask_bits_from_argmax_kernel(const int32_t* __restrict__ am, int64_t N, int32_t cat,
                                                                     unsigned long long* __restrict__ bits) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t nwords = (N + 63) >> 6;
    for (int64_t w = wave0; w < nwords; w += nwaves) {
        const int64_t i = w * 64 + lane;
        const unsigned long long m = __ballot(i < N && am[i] == cat);
        if (lane == 0) bits[w] = m;
    }
}

// first-maximum argmax over a float vector: per-block partials, then one block finishes
__global__ void argmax_partial_kernel(const float* __restrict__ v, int64_t N, float* __restrict__ pv, int64_t* __restrict__ pi) {
    float bv = -INFINITY;
    int64_t bi = INT64_MAX;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
        float x = v[i];
        if (x > bv) { bv = x; bi = i; }   // ascending i per thread: first max kept
    }
    __shared__ float sv[256];
    __shared__ int64_t si[256];
    sv[threadIdx.x] = bv;
    si[threadIdx.x] = bi;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            float ov = sv[threadIdx.x + s];
            int64_t oi = si[threadIdx.x + s];
            if (ov > sv[threadIdx.x] || (ov == sv[threadIdx.x] && oi < si[threadIdx.x])) {
                sv[threadIdx.x] = ov;
                si[threadIdx.x] = oi;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        pv[blockIdx.x] = sv[0];
        pi[blockIdx.x] = si[0];
    }
}

struct SplitChunk {
    int q_base, rows, QT;
};

struct SplitPlan {
    int Qtot, KC, nkc, nchunks, max_rows;
    bool stream;   // sim_stream_f16_kernel: KC = 64 * SPC columns per LDS buffer, nkc chunks streamed per tile
    bool kswap;    // sim_kswap_f16_kernel: two resident K chunks, swapped once per T tiles
    int SPC;
    SplitChunk chunks[64];
    size_t hdr_bytes, ws_bytes;
    size_t lds_bytes(const SplitChunk& c) const {
        // + the per-query 2^-S table and the query-index table of column-block launches (4 B per row each)
        if (stream) return (size_t)2 * (c.rows + 1) * (2 * KC + kRowPadHalves) * 2 + (size_t)c.QT * 32 * 2 * sizeof(float);
        return (size_t)4 * (c.rows + 1) * (KC + kRowPadHalves) + (size_t)c.QT * 32 * 2 * sizeof(float);   // + the zero row
    }
};

#ifndef AVL_SIM_EXTRA_ROWS
#define AVL_SIM_EXTRA_ROWS 1
#endif
constexpr bool kSimExtraRows = AVL_SIM_EXTRA_ROWS != 0;
constexpr size_t kLdsBudget = 163840 - 512;  // 160 KiB per workgroup minus slack

static bool stream_fits(int rows, int KS) {
    const size_t buf = (size_t)rows * (2 * KS + kRowPadHalves) * 2;   // one chunk: rows x (hi | lo | pad)
    const size_t zero_row = (size_t)(2 * KS + kRowPadHalves) * 2;     // + one all-zero row per buffer
    return 2 * (buf + zero_row) + 256 * sizeof(float) <= kLdsBudget && buf <= (size_t)kStreamFill * kSplitThreads * 16;
}

static bool make_split_plan(int D, int Q, SplitPlan& p, bool allow_stream = true) {
    if (D % 64 != 0 || D <= 0 || Q <= 0) return false;
    p.stream = false;
    p.kswap = false;
    p.SPC = 0;
    p.Qtot = (Q + 31) / 32 * 32;
    // rows that fit next to a <=512-wide K chunk: up to 3 MFMA tiles (96 rows) in one pass, e.g. the reference's
    // "64 categories + other" (Q = 65) runs as ONE pass with 65 resident rows instead of 64 + 1
    const int kc0 = D < 512 ? D : 512;
    int r3 = (int)((kLdsBudget - 192 * sizeof(float)) / (4 * (size_t)(kc0 + kRowPadHalves))) - 1;   // one row is the zero row
    if (r3 > 96) r3 = 96;
    // fewest passes over the feature map, balanced: npass = ceil(Q / r3) chunks of ceil(Q / npass) rows
    const int npass = (Q + r3 - 1) / r3;
    if (npass > 64) return false;
    const int per = (Q + npass - 1) / npass;
    p.nchunks = 0;
    p.max_rows = 0;
    for (int base = 0; base < Q; base += per) {
        const int take = Q - base < per ? Q - base : per;
        p.chunks[p.nchunks++] = SplitChunk{base, take, (take + 31) / 32};
        if (take > p.max_rows) p.max_rows = take;
    }
    int kcmax = (int)((kLdsBudget - 192 * sizeof(float)) / (4 * (size_t)(p.max_rows + 1))) - kRowPadHalves;
    kcmax = (kcmax / 64) * 64;
    if (kcmax < 64) return false;
    p.nkc = (D + kcmax - 1) / kcmax;
    const int kc = (D + p.nkc - 1) / p.nkc;
    p.KC = ((kc + 63) / 64) * 64;
    p.nkc = (D + p.KC - 1) / p.KC;
    p.hdr_bytes = (((size_t)p.Qtot * sizeof(float)) + kHdrAlign - 1) / kHdrAlign * kHdrAlign;
    p.ws_bytes = p.hdr_bytes + (size_t)p.nkc * 2 * p.Qtot * (p.KC + kRowPadHalves) * sizeof(_Float16);
    // exactly two LDS-sized K chunks and at most two MFMA tiles of queries in one pass: the K-swap kernel (T tiles per image swap)
    static const bool kswap_on = [] { const char* e = std::getenv("AVL_SIM_KSWAP"); return !(e && e[0] == '0'); }();
    p.kswap = kswap_on && allow_stream && p.nkc == 2 && p.nchunks == 1 && p.max_rows <= 64;
    if (p.kswap) return true;
    // one streamed pass handles up to 128 queries at any D: take it when it saves map passes (Q > 78 at D = 512) or when
    // the resident plan would have to refill LDS per K chunk anyway (D > 512)
    if (allow_stream && D % 128 == 0) {
        const int npass_s = (Q + 127) / 128;
        const int per_s = (Q + npass_s - 1) / npass_s;
        if (npass_s <= 64 && (p.nkc > 1 || npass_s < p.nchunks)) {
            const int spc = (D % 256 == 0 && stream_fits(per_s, 256)) ? 4 : 2;
            if (stream_fits(per_s, 64 * spc)) {
                p.stream = true;
                p.SPC = spc;
                p.KC = 64 * spc;
                p.nkc = D / p.KC;
                p.nchunks = 0;
                p.max_rows = 0;
                for (int base = 0; base < Q; base += per_s) {
                    const int take = Q - base < per_s ? Q - base : per_s;
                    p.chunks[p.nchunks++] = SplitChunk{base, take, (take + 31) / 32};
                    if (take > p.max_rows) p.max_rows = take;
                }
                p.ws_bytes = p.hdr_bytes + (size_t)p.nkc * p.Qtot * (2 * p.KC + kRowPadHalves) * sizeof(_Float16);
            }
        }
    }
    return true;
}

// hipFuncSetAttribute is not free: raise the dynamic-LDS limit of a kernel only when a launch needs more than before
static int ensure_dynamic_lds(const void* kern, size_t bytes) {
    struct Entry { const void* k; size_t bytes; int dev; };
    static thread_local Entry table[64];   // > number of kernel instantiations in this file
    static thread_local int used = 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    for (int i = 0; i < used; ++i)
        if (table[i].k == kern && table[i].dev == dev) {
            if (table[i].bytes >= bytes) return AVL_OK;
            AVL_HIP_CHECK(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
            table[i].bytes = bytes;
            return AVL_OK;
        }
    AVL_HIP_CHECK(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    if (used < 64) table[used++] = Entry{kern, bytes, dev};
    return AVL_OK;
}

static int run_exact(const float* d_feat, int64_t N, int D, int64_t ld, const float* d_q, int Q, int64_t ldq,
                     float* d_scores, int32_t* d_argmax, float* d_best, hipStream_t st) {
    const int Dp = (D + 3) & ~3;
    const size_t lds = (size_t)kExactQB * Dp * sizeof(float);
    AVL_REQUIRE(lds <= 160 * 1024, "avl_sim_scores: D=%d too large for the exact path", D);
    const bool vec4 = (D % 4 == 0) && (ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(d_feat) & 15) == 0);
    const int waves_per_block = 4;
    int64_t blocks = (N + waves_per_block * 4 - 1) / (waves_per_block * 4);  // 4 rows per wave iteration
    const int64_t maxb = (int64_t)num_cus() * 8;
    if (blocks > maxb) blocks = maxb;
    if (blocks < 1) blocks = 1;
    auto kern = vec4 ? sim_exact_kernel<true> : sim_exact_kernel<false>;
    if (lds > 64 * 1024) {
        int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds);
        if (rc != AVL_OK) return rc;
    }
    // a scratch best buffer is needed to chain chunks when the caller did not ask for one
    for (int q0 = 0, first = 1; q0 < Q; q0 += kExactQB, first = 0) {
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, st, d_feat, N, D, ld, d_q, Q, ldq, q0, d_scores,
                           d_argmax, d_best, first);
    }
    AVL_HIP_CHECK(hipGetLastError());
    return AVL_OK;
}

template <int SPC, bool PRE, bool QM, bool P24 = false>
static const void* pick_stream_kernel(int QT, bool tile_block) {
    if (tile_block && QT == 2) return reinterpret_cast<const void*>(sim_stream_tb_f16_kernel<2, SPC, PRE, QM, P24>);
    switch (QT) {
        case 1: return reinterpret_cast<const void*>(sim_stream_f16_kernel<1, SPC, PRE, QM, P24>);
        case 2: return reinterpret_cast<const void*>(sim_stream_f16_kernel<2, SPC, PRE, QM, P24>);
        case 3: return reinterpret_cast<const void*>(sim_stream_f16_kernel<3, SPC, PRE, QM, P24>);
        default: return reinterpret_cast<const void*>(sim_stream_f16_kernel<4, SPC, PRE, QM, P24>);
    }
}

template <bool PRE, bool QM, bool P24>
static const void* pick_stream_kernel_spc(int SPC, int QT, bool tile_block) {
    return SPC == 4 ? pick_stream_kernel<4, PRE, QM, P24>(QT, tile_block) : pick_stream_kernel<2, PRE, QM, P24>(QT, tile_block);
}

#ifndef AVL_KSWAP_T
#define AVL_KSWAP_T 3
#endif
template <int NS, bool PRE, bool QM, bool P24>
static const void* pick_kswap_kernel_ns(int QT) {
    if (QT == 2) return reinterpret_cast<const void*>(sim_kswap_f16_kernel<2, AVL_KSWAP_T, NS, PRE, QM, P24>);
    // one query tile: twice the voxel tiles per image swap in the same 96 accumulator registers (<= 32 rows fit LDS whole up to
    // D = 1216, so two chunks mean a wider map: run-time trip count only)
    return reinterpret_cast<const void*>(sim_kswap_f16_kernel<1, 2 * AVL_KSWAP_T, 0, PRE, QM, P24>);
}
// ns: compile-time steps per chunk -- 8 (D = 1024), 6 (D = 768: CLIP ViT-L), 0 = run-time trip count
template <bool PRE, bool QM, bool P24>
static const void* pick_kswap_kernel(int QT, int ns) {
    return ns == 8 ? pick_kswap_kernel_ns<8, PRE, QM, P24>(QT) : (ns == 6 ? pick_kswap_kernel_ns<6, PRE, QM, P24>(QT) : pick_kswap_kernel_ns<0, PRE, QM, P24>(QT));
}

static bool split_plan_is_fused(const SplitPlan& p, int D) { return !p.stream && p.nkc == 1 && D <= 512; }

template <int QT, bool QM>
static const void* pick_split24_kernel(bool s8, bool fq) {   // compact prepared map (P24)
    if (fq) return s8 ? reinterpret_cast<const void*>(sim_split_f16_kernel<QT, 8, true, true, QM, true>)
                      : reinterpret_cast<const void*>(sim_split_f16_kernel<QT, 0, true, true, QM, true>);
    return reinterpret_cast<const void*>(sim_split_f16_kernel<QT, 0, true, false, QM, true>);
}

// extra-row variants (Q = 32 QT + 1..4 on the fused D = 512 path): raw / prepared / compact map
template <int QT>
static const void* pick_split_xr_kernel(bool prepared, bool p24) {
    if (p24) return reinterpret_cast<const void*>(sim_split_f16_kernel<QT, 8, true, true, false, true, true>);
    if (prepared) return reinterpret_cast<const void*>(sim_split_f16_kernel<QT, 8, true, true, false, false, true>);
    return reinterpret_cast<const void*>(sim_split_f16_kernel<QT, 8, false, true, false, false, true>);
}

template <int QT, bool PRE, bool QM>
static const void* pick_split_kernel(bool s8, bool fq) {
    if (fq) return s8 ? reinterpret_cast<const void*>(sim_split_f16_kernel<QT, 8, PRE, true, QM>)
                      : reinterpret_cast<const void*>(sim_split_f16_kernel<QT, 0, PRE, true, QM>);
    return reinterpret_cast<const void*>(sim_split_f16_kernel<QT, 0, PRE, false, QM>);
}

static int run_fixup(const float* d_feat, int64_t N, int D, int64_t ld, const float* d_q, int Q, int64_t ldq, const uint32_t* d_flags,
                     float* d_scores, int32_t* d_argmax, float* d_best, hipStream_t st) {
    const int64_t nwords = (N + 31) >> 5;
    int64_t blocks = (nwords + 255) / 256;          // a wave scans 64 words: 4 x the waves needed, flagged rows spread out
    const int64_t maxb = (int64_t)num_cus() * 8;
    if (blocks > maxb) blocks = maxb;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(sim_fixup_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, st, d_feat, N, D, ld, d_q, Q, ldq, d_flags, d_scores,
                       d_argmax, d_best);
    AVL_HIP_CHECK(hipGetLastError());
    return AVL_OK;
}

// One column block of a similarity call: Qg gathered query rows x the map columns [feat, feat + D) (the whole call when
// qmap == nullptr).  PRE: d_row_scale = per-row 2^-s of the prepared map (nullable).  !PRE: d_flags = (N + 31) / 32 range-guard
// words, written by the first launch of the call and OR-ed into by the others; the caller runs sim_fixup_rows_kernel at the end.
// Qs = row stride of d_scores (the caller's Q); qmap[local row] = the caller's query index (ascending), nullptr = identity.
template <bool PRE>
static int run_split(const float* d_feat, int64_t N, int D, int64_t ld, const float* d_q, int Qg, int64_t ldq, int Qs,
                     float* d_scores, int32_t* d_argmax, float* d_best, const SplitPlan& p, void* d_ws, const float* d_row_scale,
                     uint32_t* d_flags, const int32_t* d_qmap, bool first_launch, hipStream_t st, bool p24 = false, int col0 = 0) {
    // col0 (compact maps only): first column of the window; d_feat then is the map's first byte (two planes per row: a window is no
    // pointer offset), and the compile-time-unrolled variants, which pair the steps of a 128-column block, need col0 % 128 == 0
    float* inv_scale = reinterpret_cast<float*>(d_ws);
    _Float16* img = reinterpret_cast<_Float16*>(reinterpret_cast<char*>(d_ws) + p.hdr_bytes);
    const bool fq = split_plan_is_fused(p, D);   // resident image built inside the kernel: no prep launch, no workspace
    if (!fq)
        hipLaunchKernelGGL(sim_prep_queries_kernel, dim3(p.Qtot), dim3(256), 0, st, d_q, Qg, D, ldq, inv_scale, img, p.Qtot, p.KC,
                           p.nkc, p.stream ? 1 : 0);
    const int64_t ntiles = (N + kTileRows - 1) / kTileRows;
    int64_t blocks = ntiles < num_cus() ? ntiles : num_cus();
    if (blocks < 1) blocks = 1;
    if (p.stream) {
        for (int ci = 0; ci < p.nchunks; ++ci) {
            const SplitChunk& c = p.chunks[ci];
            const bool tb = ((ld * (p24 ? 3 : 4)) % 4096) != 0;   // tile blocking (QT = 2 only) loses on 4 KiB-multiple row strides, see the kernel
            const void* kern = nullptr;
            if constexpr (PRE) {
                if (p24) kern = d_qmap ? pick_stream_kernel_spc<true, true, true>(p.SPC, c.QT, tb) : pick_stream_kernel_spc<true, false, true>(p.SPC, c.QT, tb);
            }
            if (!kern) kern = d_qmap ? pick_stream_kernel_spc<PRE, true, false>(p.SPC, c.QT, tb) : pick_stream_kernel_spc<PRE, false, false>(p.SPC, c.QT, tb);
            const size_t lds = p.lds_bytes(c);
            int rc = ensure_dynamic_lds(kern, lds);
            if (rc != AVL_OK) return rc;
            const _Float16* img_c = img;
            const float* isc_c = inv_scale;
            int Qtot = p.Qtot, nch = p.nkc, q_base = c.q_base, rows = c.rows, first = (ci == 0 && first_launch) ? 1 : 0;
            void* args[] = {&d_feat, &N, &D, &ld, &img_c, &isc_c, &Qtot, &nch, &q_base, &rows, &Qs, &d_scores, &d_argmax, &d_best, &first,
                            &d_row_scale, &d_flags, &d_qmap, &col0};
            AVL_HIP_CHECK(hipLaunchKernel(kern, dim3((unsigned)blocks), dim3(kSplitThreads), args, lds, st));
        }
        AVL_HIP_CHECK(hipGetLastError());
        return AVL_OK;
    }
    if (p.kswap) {
        const SplitChunk& c = p.chunks[0];
        const int s8 = (p24 && col0 % 128 != 0) ? 0 : ((p.KC == 512 && D == 1024) ? 8 : ((p.KC == 384 && D == 768) ? 6 : 0));
        const void* kern = nullptr;
        if constexpr (PRE) {
            if (p24) kern = d_qmap ? pick_kswap_kernel<true, true, true>(c.QT, s8) : pick_kswap_kernel<true, false, true>(c.QT, s8);
        }
        if (!kern) kern = d_qmap ? pick_kswap_kernel<PRE, true, false>(c.QT, s8) : pick_kswap_kernel<PRE, false, false>(c.QT, s8);
        const size_t lds = p.lds_bytes(c);
        int rc = ensure_dynamic_lds(kern, lds);
        if (rc != AVL_OK) return rc;
        const _Float16* img_c = img;
        const float* isc_c = inv_scale;
        int Qtot = p.Qtot, KC = p.KC, q_base = c.q_base, rows = c.rows, first = first_launch ? 1 : 0;
        void* args[] = {&d_feat, &N, &D, &ld, &img_c, &isc_c, &Qtot, &KC, &q_base, &rows, &Qs, &d_scores, &d_argmax, &d_best, &first,
                        &d_row_scale, &d_flags, &d_qmap, &col0};
        AVL_HIP_CHECK(hipLaunchKernel(kern, dim3((unsigned)blocks), dim3(kSplitThreads), args, lds, st));
        AVL_HIP_CHECK(hipGetLastError());
        return AVL_OK;
    }
    for (int ci = 0; ci < p.nchunks; ++ci) {
        const SplitChunk& c = p.chunks[ci];
        const bool s8 = (p.nkc == 1 && D == 512) && (!p24 || col0 % 128 == 0);   // the LSeg / CLIP ViT-B feature width: fully unrolled k loop
        // "N categories + other": 1..4 rows beyond full 32-row tiles go to the 4x4x4 MFMA path instead of a padded tile
        const int xr = c.rows % 32;
        // (same-box A/B at 2 M voxels: 65 rows 0.753 -> 0.726 ms raw, 0.758 -> 0.718 prepared, 0.645 -> 0.624 compact; 33 rows on a raw
        // map are the one case that loses -- 0.689 -> 0.718 ms, the single-tile kernel with three accumulator sets and the on-the-fly
        // split is at its register limit -- so raw maps take the path from two full tiles on)
        const bool use_xr = kSimExtraRows && s8 && fq && !d_qmap && xr >= 1 && xr <= 4 && c.rows > 32 && c.rows / 32 <= 2 &&
                            (PRE || c.rows / 32 == 2);
        const void* kern = use_xr ? (c.rows / 32 == 2 ? pick_split_xr_kernel<2>(PRE, p24) : pick_split_xr_kernel<1>(PRE, p24))
                           : (p24 && d_qmap) ? (c.QT == 3 ? pick_split24_kernel<3, true>(s8, fq) : (c.QT == 2 ? pick_split24_kernel<2, true>(s8, fq) : pick_split24_kernel<1, true>(s8, fq)))
                           : p24  ? (c.QT == 3 ? pick_split24_kernel<3, false>(s8, fq) : (c.QT == 2 ? pick_split24_kernel<2, false>(s8, fq) : pick_split24_kernel<1, false>(s8, fq)))
                           : d_qmap ? (c.QT == 3 ? pick_split_kernel<3, PRE, true>(s8, fq)
                                               : (c.QT == 2 ? pick_split_kernel<2, PRE, true>(s8, fq) : pick_split_kernel<1, PRE, true>(s8, fq)))
                                  : (c.QT == 3 ? pick_split_kernel<3, PRE, false>(s8, fq)
                                               : (c.QT == 2 ? pick_split_kernel<2, PRE, false>(s8, fq) : pick_split_kernel<1, PRE, false>(s8, fq)));
        const size_t lds = p.lds_bytes(c);
        int rc = ensure_dynamic_lds(kern, lds);
        if (rc != AVL_OK) return rc;
        const _Float16* img_c = img;
        const float* isc_c = inv_scale;
        int Qtot = p.Qtot, KC = p.KC, nkc = p.nkc, q_base = c.q_base, rows = c.rows, first = (ci == 0 && first_launch) ? 1 : 0;
        void* args[] = {&d_feat, &N, &D, &ld, &img_c, &isc_c, &d_q, &ldq, &Qtot, &KC, &nkc, &q_base, &rows, &Qs,
                        &d_scores, &d_argmax, &d_best, &first, &d_row_scale, &d_flags, &d_qmap, &col0};
        AVL_HIP_CHECK(hipLaunchKernel(kern, dim3((unsigned)blocks), dim3(kSplitThreads), args, lds, st));
    }
    AVL_HIP_CHECK(hipGetLastError());
    return AVL_OK;
}

static int run_mfma_f32(const float* d_feat, int64_t N, int D, int64_t ld, const float* d_q, int Q, int64_t ldq,
                        float* d_scores, int32_t* d_argmax, float* d_best, const SplitPlan& p, void* d_ws, hipStream_t st) {
    float* img = reinterpret_cast<float*>(d_ws);   // needs nkc*Qtot*(KC+4)*4 bytes <= the split image of the same plan
    hipLaunchKernelGGL(sim_prep_queries_f32_kernel, dim3(p.Qtot), dim3(256), 0, st, d_q, Q, D, ldq, img, p.Qtot, p.KC, p.nkc);
    const int64_t ntiles = (N + kTileRows - 1) / kTileRows;
    int64_t blocks = ntiles < num_cus() ? ntiles : num_cus();
    if (blocks < 1) blocks = 1;
    for (int ci = 0; ci < p.nchunks; ++ci) {
        const SplitChunk& c = p.chunks[ci];
        auto kern = c.QT == 3 ? sim_mfma_f32_kernel<3> : (c.QT == 2 ? sim_mfma_f32_kernel<2> : sim_mfma_f32_kernel<1>);
        const size_t lds = (size_t)c.rows * (p.KC + kRowPadF32) * sizeof(float);
        int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds);
        if (rc != AVL_OK) return rc;
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(kSplitThreads), lds, st, d_feat, N, D, ld, img, p.Qtot, p.KC, p.nkc, c.q_base,
                           c.rows, Q, d_scores, d_argmax, d_best, ci == 0 ? 1 : 0);
    }
    AVL_HIP_CHECK(hipGetLastError());
    return AVL_OK;
}

}  // namespace avl

using namespace avl;

extern "C" {

static size_t align_up(size_t b, size_t a) { return (b + a - 1) / a * a; }
static size_t guard_bytes(int64_t N) { return align_up((size_t)((N + 31) >> 5) * sizeof(uint32_t), kHdrAlign); }
static size_t blocks_bytes(int D, int Q);

int avl_sim_workspace_bytes(int D, int Q, size_t* h_bytes) {
    AVL_REQUIRE(h_bytes, "avl_sim_workspace_bytes: null output");
    SplitPlan p, r;
    size_t b = kHdrAlign;   // enough for either plan: the exact fp32-MFMA mode always uses the resident one
    if (make_split_plan(D, Q, p, true) && p.ws_bytes > b) b = p.ws_bytes;
    if (make_split_plan(D, Q, r, false) && r.ws_bytes > b) b = r.ws_bytes;
    *h_bytes = align_up(b, kHdrAlign);
    return AVL_OK;
}

int avl_sim_workspace_bytes_n(int64_t N, int D, int Q, size_t* h_bytes) {
    AVL_REQUIRE(N >= 0, "avl_sim_workspace_bytes_n: bad N");
    const int rc = avl_sim_workspace_bytes(D, Q, h_bytes);
    if (rc == AVL_OK) *h_bytes += guard_bytes(N) + blocks_bytes(D, Q);
    return rc;
}

// Workspace layout of the matrix-core paths (every part aligned to kHdrAlign):
//   [query image of the widest plan][range-guard words (raw split path)][gathered queries Q x D f32][query map Q x i32]
// the last two only for column-block calls.
static size_t blocks_bytes(int D, int Q) { return align_up((size_t)Q * D * sizeof(float), kHdrAlign) + align_up((size_t)Q * sizeof(int32_t), kHdrAlign); }

struct ColGroup {
    int col0, cols;
    std::vector<int32_t> rows;   // the caller's query indices, ascending
};

// Column-support groups: queries whose non-zero columns lie in the same [begin, end) window (rounded outward to the 128-column
// granularity of the streamed kernel) are scored together against just those columns of the map -- a text query never meets
// the audio block of a fused visual | audio map (BASELINE config 5), which halves the matrix-core work per voxel byte while the
// map is still read exactly once overall.  Groups are ordered by their first query so that ties resolve as in one dense pass.
static void make_col_groups(int D, int Q, const int32_t* h_begin, const int32_t* h_end, std::vector<ColGroup>& groups) {
    groups.clear();
    std::vector<int> zero_rows;
    for (int qi = 0; qi < Q; ++qi) {
        int b = h_begin[qi] < 0 ? 0 : h_begin[qi], e = h_end[qi] > D ? D : h_end[qi];
        if (e <= b) {   // an all-zero query scores 0 against any window: it joins the first group
            zero_rows.push_back(qi);
            continue;
        }
        b = (b / 128) * 128;
        e = ((e + 127) / 128) * 128;
        if (e > D) e = D;
        size_t g = 0;
        for (; g < groups.size(); ++g)
            if (groups[g].col0 == b && groups[g].cols == e - b) break;
        if (g == groups.size()) groups.push_back(ColGroup{b, e - b, {}});
        groups[g].rows.push_back(qi);
    }
    if (groups.empty()) groups.push_back(ColGroup{0, 128 < D ? 128 : D, {}});
    for (int qi : zero_rows) groups[0].rows.push_back(qi);
    std::sort(groups[0].rows.begin(), groups[0].rows.end());   // rows ascend inside a group: "first maximum" stays the caller's
}

static int sim_scores_impl(const float* d_feat, const float* d_row_scale, int64_t N, int D, int64_t ld_feat, const float* d_queries,
                           int Q, int64_t ld_q, float* d_scores, int32_t* d_argmax, float* d_best, int precision,
                           void* d_workspace, size_t workspace_bytes, void* stream, const int32_t* h_col_begin = nullptr,
                           const int32_t* h_col_end = nullptr) {
    AVL_REQUIRE(N >= 0 && D > 0 && Q > 0, "avl_sim_scores: bad shape N=%lld D=%d Q=%d", (long long)N, D, Q);
    AVL_REQUIRE(ld_feat >= D && ld_q >= D, "avl_sim_scores: row strides must be >= D");
    AVL_REQUIRE(precision >= AVL_SIM_AUTO && precision <= AVL_SIM_PREPARED24, "avl_sim_scores: bad precision %d", precision);
    const bool p24 = precision == AVL_SIM_PREPARED24;
    if (p24) {   // compact prepared map: 3 bytes per element, dense rows (resident, streamed and column-block kernels)
        AVL_REQUIRE(D % 128 == 0 && ld_feat == D && (reinterpret_cast<uintptr_t>(d_feat) & 15) == 0,
                    "avl_sim_scores_prepared24: needs D %% 128 == 0 and dense 16-byte aligned rows (D=%d)", D);
        AVL_REQUIRE(d_row_scale, "avl_sim_scores_prepared24: the compact form always has row scales");
        precision = AVL_SIM_PREPARED;
    }
    if (N == 0) return AVL_OK;
    AVL_REQUIRE(d_feat && d_queries, "avl_sim_scores: null input");
    hipStream_t st = as_stream(stream);
    SplitPlan p;
    const bool aligned = (ld_feat % 4 == 0) && ((reinterpret_cast<uintptr_t>(d_feat) & 15) == 0) &&
                         (!d_scores || (reinterpret_cast<uintptr_t>(d_scores) & 15) == 0);
    const bool can_split = make_split_plan(D, Q, p, precision != AVL_SIM_EXACT) && aligned;
    bool use_split, use_f32_mfma = false;
    if (precision == AVL_SIM_EXACT_VALU) use_split = false;
    else if (precision == AVL_SIM_EXACT) {
        use_split = false;
        use_f32_mfma = can_split;     // same shape constraints; otherwise the vector-ALU kernel
    } else if (precision == AVL_SIM_SPLIT_F16 || precision == AVL_SIM_PREPARED) {
        AVL_REQUIRE(can_split, "avl_sim_scores: SPLIT_F16 needs D %% 64 == 0 and 16-byte aligned rows (D=%d ld=%lld)", D,
                    (long long)ld_feat);
        use_split = true;
    } else use_split = can_split;   // ~1e-6 accurate and HBM-bound for every Q; EXACT remains the fallback / on request

    // column-support groups (only the split path exploits them; more than one group = block launches)
    std::vector<ColGroup> groups;
    std::vector<SplitPlan> gplans;
    if (use_split && h_col_begin && h_col_end) {
        make_col_groups(D, Q, h_col_begin, h_col_end, groups);
        bool ok = groups.size() > 1 && groups.size() <= 16;
        for (size_t g = 0; ok && g < groups.size(); ++g) {
            SplitPlan gp;
            ok = make_split_plan(groups[g].cols, (int)groups[g].rows.size(), gp, true);
            gplans.push_back(gp);
        }
        if (!ok) { groups.clear(); gplans.clear(); }
    }
    const bool blocks = !groups.empty();

    // chaining launches (query chunks, column blocks) needs best + argmax buffers even if the caller does not want them
    float* best = d_best;
    int32_t* amax = d_argmax;
    float* tmp_best = nullptr;
    int32_t* tmp_amax = nullptr;
    int nlaunch = (use_split || use_f32_mfma) ? p.nchunks : (Q + kExactQB - 1) / kExactQB;
    if (blocks) {
        nlaunch = 0;
        for (const SplitPlan& gp : gplans) nlaunch += gp.nchunks;
    }
    if (nlaunch > 1 && (d_argmax || d_best)) {
        if (!best) {
            AVL_HIP_CHECK(hipMallocAsync(reinterpret_cast<void**>(&tmp_best), (size_t)N * sizeof(float), st));
            best = tmp_best;
        }
        if (!amax && blocks) {   // blocks interleave the caller's query indices: the tie-break needs the previous index
            AVL_HIP_CHECK(hipMallocAsync(reinterpret_cast<void**>(&tmp_amax), (size_t)N * sizeof(int32_t), st));
            amax = tmp_amax;
        }
    }
    int rc = AVL_OK;
    if (use_split || use_f32_mfma) {
        const bool prepared = precision == AVL_SIM_PREPARED;
        size_t img_bytes = 0;
        if (blocks) {
            for (const SplitPlan& gp : gplans) img_bytes = img_bytes > gp.ws_bytes ? img_bytes : gp.ws_bytes;
        } else if (use_f32_mfma || !split_plan_is_fused(p, D)) {   // the fused-prep path keeps the query image in LDS only
            img_bytes = p.ws_bytes;
        }
        img_bytes = img_bytes ? align_up(img_bytes, kHdrAlign) : 0;
        const size_t flag_bytes = (use_split && !prepared) ? guard_bytes(N) : 0;
        const size_t blk_bytes = blocks ? blocks_bytes(D, Q) : 0;
        const size_t need = img_bytes + flag_bytes + blk_bytes;
        char* ws = static_cast<char*>(d_workspace);
        void* tmp_ws = nullptr;
        if (need > 0 && (!ws || workspace_bytes < need)) {
            AVL_HIP_CHECK(hipMallocAsync(&tmp_ws, need, st));
            ws = static_cast<char*>(tmp_ws);
        }
        uint32_t* flags = flag_bytes ? reinterpret_cast<uint32_t*>(ws + img_bytes) : nullptr;
        if (!use_split) {
            rc = run_mfma_f32(d_feat, N, D, ld_feat, d_queries, Q, ld_q, d_scores, amax, best, p, ws, st);
        } else if (!blocks) {
            rc = prepared ? run_split<true>(d_feat, N, D, ld_feat, d_queries, Q, ld_q, Q, d_scores, amax, best, p, ws, d_row_scale, nullptr,
                                            nullptr, true, st, p24)
                          : run_split<false>(d_feat, N, D, ld_feat, d_queries, Q, ld_q, Q, d_scores, amax, best, p, ws, nullptr, flags,
                                             nullptr, true, st);
        } else {
            float* qg = reinterpret_cast<float*>(ws + img_bytes + flag_bytes);
            int32_t* qmap = reinterpret_cast<int32_t*>(ws + img_bytes + flag_bytes + align_up((size_t)Q * D * sizeof(float), kHdrAlign));
            std::vector<int32_t> h_map;
            h_map.reserve((size_t)Q);
            for (const ColGroup& g : groups) h_map.insert(h_map.end(), g.rows.begin(), g.rows.end());
            AVL_HIP_CHECK(hipMemcpyAsync(qmap, h_map.data(), (size_t)Q * sizeof(int32_t), hipMemcpyHostToDevice, st));   // pageable: staged
            size_t row_off = 0, qg_off = 0;
            for (size_t g = 0; g < groups.size() && rc == AVL_OK; ++g) {
                const ColGroup& cg = groups[g];
                const int Qg = (int)cg.rows.size();
                float* qg_g = qg + qg_off;
                const int32_t* qmap_g = qmap + row_off;
                hipLaunchKernelGGL(sim_gather_queries_kernel, dim3((unsigned)Qg), dim3(256), 0, st, d_queries, ld_q, qmap_g, Qg, cg.col0,
                                   cg.cols, qg_g);
                // the column window of every row: a pointer offset of 4 B per column; a compact map has two planes per row and its window
                // travels as col0 (windows are 128-aligned: whole residual lines)
                const float* feat_g = p24 ? d_feat : d_feat + cg.col0;
                rc = prepared ? run_split<true>(feat_g, N, cg.cols, ld_feat, qg_g, Qg, cg.cols, Q, d_scores, amax, best, gplans[g], ws,
                                                d_row_scale, nullptr, qmap_g, g == 0, st, p24, p24 ? cg.col0 : 0)
                              : run_split<false>(feat_g, N, cg.cols, ld_feat, qg_g, Qg, cg.cols, Q, d_scores, amax, best, gplans[g], ws, nullptr,
                                                 flags, qmap_g, g == 0, st);
                row_off += (size_t)Qg;
                qg_off += (size_t)Qg * cg.cols;
            }
        }
        if (rc == AVL_OK && use_split && !prepared)
            rc = run_fixup(d_feat, N, D, ld_feat, d_queries, Q, ld_q, flags, d_scores, d_argmax, d_best, st);
        if (tmp_ws) (void)hipFreeAsync(tmp_ws, st);
    } else {
        rc = run_exact(d_feat, N, D, ld_feat, d_queries, Q, ld_q, d_scores, d_argmax, best, st);
    }
    if (tmp_best) (void)hipFreeAsync(tmp_best, st);
    if (tmp_amax) (void)hipFreeAsync(tmp_amax, st);
    return rc;
}

int avl_sim_scores_ws(const float* d_feat, int64_t N, int D, int64_t ld_feat, const float* d_queries, int Q,
                      int64_t ld_q, float* d_scores, int32_t* d_argmax, float* d_best, int precision,
                      void* d_workspace, size_t workspace_bytes, void* stream) {
    return sim_scores_impl(d_feat, nullptr, N, D, ld_feat, d_queries, Q, ld_q, d_scores, d_argmax, d_best, precision, d_workspace,
                           workspace_bytes, stream);
}

int avl_sim_scores_blocks(const float* d_feat, const float* d_row_scale, int64_t N, int D, int64_t ld_feat, const float* d_queries,
                          int Q, int64_t ld_q, const int32_t* h_col_begin, const int32_t* h_col_end, float* d_scores,
                          int32_t* d_argmax, float* d_best, int precision, void* d_workspace, size_t workspace_bytes, void* stream) {
    AVL_REQUIRE(h_col_begin && h_col_end, "avl_sim_scores_blocks: null column support");
    return sim_scores_impl(d_feat, d_row_scale, N, D, ld_feat, d_queries, Q, ld_q, d_scores, d_argmax, d_best, precision, d_workspace,
                           workspace_bytes, stream, h_col_begin, h_col_end);
}

int avl_sim_scores_prepared(const float* d_feat, const float* d_row_scale, int64_t N, int D, int64_t ld_feat,
                            const float* d_queries, int Q, int64_t ld_q, float* d_scores, int32_t* d_argmax, float* d_best,
                            void* d_workspace, size_t workspace_bytes, void* stream) {
    return sim_scores_impl(d_feat, d_row_scale, N, D, ld_feat, d_queries, Q, ld_q, d_scores, d_argmax, d_best, AVL_SIM_PREPARED,
                           d_workspace, workspace_bytes, stream);
}

int avl_sim_scores_prepared24(const void* d_map24, const float* d_row_scale, int64_t N, int D, const float* d_queries, int Q,
                              int64_t ld_q, float* d_scores, int32_t* d_argmax, float* d_best, void* d_workspace,
                              size_t workspace_bytes, void* stream) {
    AVL_REQUIRE(d_row_scale, "avl_sim_scores_prepared24: the compact form always has row scales");
    return sim_scores_impl(reinterpret_cast<const float*>(d_map24), d_row_scale, N, D, D, d_queries, Q, ld_q, d_scores, d_argmax, d_best,
                           AVL_SIM_PREPARED24, d_workspace, workspace_bytes, stream);
}

int avl_sim_prepare_map24(const float* d_feat, int64_t N, int D, int64_t ld_feat, void* d_map24, float* d_row_scale, void* stream) {
    AVL_REQUIRE(N >= 0 && D > 0 && ld_feat >= D, "avl_sim_prepare_map24: bad shape");
    AVL_REQUIRE(D % 128 == 0 && ld_feat % 4 == 0 && (reinterpret_cast<uintptr_t>(d_feat) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(d_map24) & 15) == 0,
                "avl_sim_prepare_map24: needs D %% 128 == 0 (whole 128-column residual lines) and 16-byte aligned rows (D=%d ld=%lld)", D,
                (long long)ld_feat);
    if (N == 0) return AVL_OK;
    AVL_REQUIRE(d_feat && d_map24 && d_row_scale, "avl_sim_prepare_map24: null pointer");
    int64_t blocks = (N + 3) / 4;
    const int64_t maxb = (int64_t)num_cus() * 16;
    if (blocks > maxb) blocks = maxb;
    hipLaunchKernelGGL(sim_prepare_map24_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), d_feat, N, D, ld_feat,
                       reinterpret_cast<unsigned char*>(d_map24), d_row_scale);
    AVL_HIP_CHECK(hipGetLastError());
    return AVL_OK;
}

int avl_sim_prepare_map(float* d_feat, int64_t N, int D, int64_t ld_feat, float* d_row_scale, void* stream) {
    AVL_REQUIRE(N >= 0 && D > 0 && ld_feat >= D, "avl_sim_prepare_map: bad shape");
    AVL_REQUIRE(D % 64 == 0 && ld_feat % 4 == 0 && (reinterpret_cast<uintptr_t>(d_feat) & 15) == 0,
                "avl_sim_prepare_map: needs D %% 64 == 0 and 16-byte aligned rows (D=%d ld=%lld)", D, (long long)ld_feat);
    if (N == 0) return AVL_OK;
    AVL_REQUIRE(d_feat, "avl_sim_prepare_map: null pointer");
    int64_t blocks = (N + 3) / 4;
    const int64_t maxb = (int64_t)num_cus() * 16;
    if (blocks > maxb) blocks = maxb;
    hipLaunchKernelGGL(sim_prepare_map_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), d_feat, N, D, ld_feat, d_row_scale);
    AVL_HIP_CHECK(hipGetLastError());
    return AVL_OK;
}

int avl_sim_scores(const float* d_feat, int64_t N, int D, int64_t ld_feat, const float* d_queries, int Q, int64_t ld_q,
                   float* d_scores, int32_t* d_argmax, float* d_best, int precision, void* stream) {
    return avl_sim_scores_ws(d_feat, N, D, ld_feat, d_queries, Q, ld_q, d_scores, d_argmax, d_best, precision, nullptr, 0,
                             stream);
}

int avl_sim_scores_host(const float* h_feat, int64_t N, int D, const float* h_queries, int Q, float* h_scores,
                        int32_t* h_argmax, float* h_best, int precision) {
    AVL_REQUIRE(N >= 0 && D > 0 && Q > 0, "avl_sim_scores_host: bad shape");
    if (N == 0) return AVL_OK;
    float *d_feat = nullptr, *d_q = nullptr, *d_sc = nullptr, *d_best = nullptr;
    int32_t* d_am = nullptr;
    int rc = AVL_OK;
    auto cleanup = [&]() {
        (void)hipFree(d_feat); (void)hipFree(d_q); (void)hipFree(d_sc); (void)hipFree(d_best); (void)hipFree(d_am);
    };
#define AVL_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            set_error("%s failed: %s", #expr, hipGetErrorString(_e));                       \
            cleanup();                                                                      \
            return AVL_ERR_HIP;                                                             \
        }                                                                                   \
    } while (0)
    AVL_TRY(hipMalloc(reinterpret_cast<void**>(&d_feat), (size_t)N * D * 4));
    AVL_TRY(hipMalloc(reinterpret_cast<void**>(&d_q), (size_t)Q * D * 4));
    AVL_TRY(hipMemcpy(d_feat, h_feat, (size_t)N * D * 4, hipMemcpyHostToDevice));
    AVL_TRY(hipMemcpy(d_q, h_queries, (size_t)Q * D * 4, hipMemcpyHostToDevice));
    if (h_scores) AVL_TRY(hipMalloc(reinterpret_cast<void**>(&d_sc), (size_t)N * Q * 4));
    if (h_argmax) AVL_TRY(hipMalloc(reinterpret_cast<void**>(&d_am), (size_t)N * 4));
    if (h_best) AVL_TRY(hipMalloc(reinterpret_cast<void**>(&d_best), (size_t)N * 4));
    rc = avl_sim_scores(d_feat, N, D, D, d_q, Q, D, d_sc, d_am, d_best, precision, nullptr);
    if (rc == AVL_OK) {
        AVL_TRY(hipDeviceSynchronize());
        if (h_scores) AVL_TRY(hipMemcpy(h_scores, d_sc, (size_t)N * Q * 4, hipMemcpyDeviceToHost));
        if (h_argmax) AVL_TRY(hipMemcpy(h_argmax, d_am, (size_t)N * 4, hipMemcpyDeviceToHost));
        if (h_best) AVL_TRY(hipMemcpy(h_best, d_best, (size_t)N * 4, hipMemcpyDeviceToHost));
    }
#undef AVL_TRY
    cleanup();
    return rc;
}

int avl_mask_from_argmax(const int32_t* d_argmax, int64_t N, int32_t cat_id, uint8_t* d_mask, void* stream) {
    AVL_REQUIRE(N >= 0, "avl_mask_from_argmax: bad N");
    if (N == 0) return AVL_OK;
    AVL_REQUIRE(d_argmax && d_mask, "avl_mask_from_argmax: null pointer");
    int64_t blocks = (N + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(mask_from_argmax_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), d_argmax, N, cat_id,
                       d_mask);
    AVL_HIP_CHECK(hipGetLastError());
    return AVL_OK;
}

int avl_mask_bits_from_argmax(const int32_t* d_argmax, int64_t N, int32_t cat_id, uint64_t* d_bits, void* stream) {
    AVL_REQUIRE(N >= 0, "avl_mask_bits_from_argmax: bad N");
    if (N == 0) return AVL_OK;
    AVL_REQUIRE(d_argmax && d_bits, "avl_mask_bits_from_argmax: null pointer");
    int64_t blocks = ((N + 63) / 64 + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(mask_bits_from_argmax_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), d_argmax, N, cat_id,
                       reinterpret_cast<unsigned long long*>(d_bits));
    AVL_HIP_CHECK(hipGetLastError());
    return AVL_OK;
}

int avl_argmax_f32(const float* d_vals, int64_t N, int64_t* h_index, float* h_value, void* stream) {
    AVL_REQUIRE(N > 0 && d_vals, "avl_argmax_f32: empty input");
    hipStream_t st = as_stream(stream);
    const int nb = 256;
    char* sc = static_cast<char*>(avl::scratch((size_t)(nb + 1) * (sizeof(float) + sizeof(int64_t)) + 64));   // no allocation per call
    if (!sc) return AVL_ERR_HIP;
    int64_t* d_pi = reinterpret_cast<int64_t*>(sc);
    float* d_pv = reinterpret_cast<float*>(sc + (size_t)(nb + 1) * sizeof(int64_t));
    hipLaunchKernelGGL(argmax_partial_kernel, dim3(nb), dim3(256), 0, st, d_vals, N, d_pv, d_pi);
    float hv[nb];
    int64_t hi[nb];
    AVL_HIP_CHECK(hipMemcpyAsync(hv, d_pv, nb * sizeof(float), hipMemcpyDeviceToHost, st));
    AVL_HIP_CHECK(hipMemcpyAsync(hi, d_pi, nb * sizeof(int64_t), hipMemcpyDeviceToHost, st));
    AVL_HIP_CHECK(hipStreamSynchronize(st));
    float bv = -INFINITY;
    int64_t bi = INT64_MAX;
    for (int b = 0; b < nb; ++b)
        if (hv[b] > bv || (hv[b] == bv && hi[b] < bi)) { bv = hv[b]; bi = hi[b]; }
    if (bi == INT64_MAX) bi = 0;
    if (h_index) *h_index = bi;
    if (h_value) *h_value = bv;
    return AVL_OK;
}

}  // extern "C"
