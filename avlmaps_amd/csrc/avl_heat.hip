// Nearest-target distance-decay heatmap for gfx950.
//
// Replaces avlmaps/utils/visualize_utils.py:29-49 get_heatmap_from_mask_3d (upstream reference), the
// O(N_other * N_target) Python loop behind AVLMap.index_object (avlmaps/map/avlmap.py:73-75):
//   heat[i] = 1                                            if mask[i]
//           = clip(1 - (min_t ||pos_t - pos_i||_2 / cell_size) * decay, 0, 1)   otherwise
// (the reference divides voxel-index distances by cell_size again -- kept as is).
//
// Squared distances between int32 voxel indices are exact integers, sqrt/divide are correctly rounded in
// fp64, so the result is bit-identical to the reference's float64 arithmetic cast to float32.
//
// Two strategies, same results:
//   * windowed: heat is exactly 0 once ||d|| >= R = cell_size / decay, so only targets inside the cube of
//     radius ceil(R) around a voxel matter.  Targets are scattered into a dense byte grid over the map's
//     bounding box as a BIT grid (64 z cells per word: 4 MB for a 1000 x 1000 x 30 map, L2 resident); every voxel scans
//     the (2R+1)^2 columns of its window and finds the nearest set bit of each column with clz / ffs.
//   * brute force: targets staged through LDS, every thread owns one voxel (used when the window would
//     be larger than the target list or the dense grid would not fit).
#include <algorithm>
#include <climits>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "avl_common.h"

namespace avl {

// (one atomic per value and WORKGROUP, at most 512 workgroups: the first version issued six atomics per wave on one cache line --
// 49 k of them at 2 M voxels, and a hot word sustains ~90 atomics per microsecond: 0.6 ms of a 1.8 ms call)
__global__ __launch_bounds__(256) void heat_bbox_kernel(const int32_t* __restrict__ pos, int64_t N, int* __restrict__ bbox /* min xyz, max xyz */) {
    __shared__ int red[6][4];
    int mn[3] = {INT_MAX, INT_MAX, INT_MAX}, mx[3] = {INT_MIN, INT_MIN, INT_MIN};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            int v = pos[i * 3 + c];
            mn[c] = min(mn[c], v);
            mx[c] = max(mx[c], v);
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        for (int off = 32; off > 0; off >>= 1) {
            mn[c] = min(mn[c], __shfl_xor(mn[c], off, 64));
            mx[c] = max(mx[c], __shfl_xor(mx[c], off, 64));
        }
        if ((threadIdx.x & 63) == 0) {
            red[c][threadIdx.x >> 6] = mn[c];
            red[3 + c][threadIdx.x >> 6] = mx[c];
        }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int c = threadIdx.x;
        int v = red[c][0];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) v = c < 3 ? min(v, red[c][w]) : max(v, red[c][w]);
        if (c < 3) atomicMin(&bbox[c], v);
        else atomicMax(&bbox[c], v);
    }
}

// targets as a bit grid: one 64-bit word covers 64 consecutive z cells of an (x, y) column
// ... and, one level up, a byte per block of kCoarse x kCoarse columns: "some target lives in this block" (round 4).  A voxel whose
// window touches no occupied block is done after reading <= 9 bytes (from LDS when the coarse grid fits: 16 KB for a 1000 x 1000
// map) instead of scanning (2R + 1)^2 columns; empty blocks inside a window are stepped over.  Pure pruning: the same bits.
constexpr int kCoarseShift = 3;   // blocks of 8 x 8 columns
constexpr int kCoarseLdsBytes = 48 * 1024;

__global__ void heat_scatter_kernel(const int32_t* __restrict__ pos, const uint8_t* __restrict__ mask, int64_t N, int ox,
                                    int oy, int oz, int ny, int wz, unsigned long long* __restrict__ grid, int cny,
                                    uint8_t* __restrict__ coarse, unsigned long long* __restrict__ colmap, int cwy) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
        if (mask[i]) {
            const int x = pos[i * 3] - ox, y = pos[i * 3 + 1] - oy, z = pos[i * 3 + 2] - oz;
            const size_t w = ((size_t)x * ny + y) * wz + (z >> 6);
            atomicOr(&grid[w], 1ull << (z & 63));
            atomicOr(&colmap[(size_t)x * cwy + (y >> 6)], 1ull << (y & 63));
            coarse[(size_t)(x >> kCoarseShift) * cny + (y >> kCoarseShift)] = 1;      // idempotent plain store
        }
    }
}

__device__ __forceinline__ float heat_from_d2(long long d2, double cell_size, double decay) {
    const double dist = sqrt((double)d2) / cell_size;
    double v = 1.0 - dist * decay;
    v = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
    return (float)v;
}

// smallest |c - z| over the set bits c of `word` (bit b = cell wbase + b) restricted to [z0, z1]; INT_MAX if none
__device__ __forceinline__ int nearest_bit(unsigned long long word, int wbase, int z, int z0, int z1) {
    const int lo = max(z0 - wbase, 0), hi = min(z1 - wbase, 63);
    if (lo > hi || word == 0) return INT_MAX;
    const unsigned long long range = (hi == 63 ? ~0ull : ((1ull << (hi + 1)) - 1ull)) & ~((1ull << lo) - 1ull);
    const unsigned long long m = word & range;
    if (!m) return INT_MAX;
    const int zl = z - wbase;  // may lie outside [0, 63]
    int best = INT_MAX;
    // highest set bit at or below zl
    if (zl >= 0) {
        const unsigned long long below = zl >= 63 ? m : (m & ((1ull << (zl + 1)) - 1ull));
        if (below) best = zl - (63 - __clzll((long long)below));
    }
    // lowest set bit above zl
    if (zl < 63) {
        const unsigned long long above = zl < 0 ? m : (m & ~((1ull << (zl + 1)) - 1ull));
        if (above) best = min(best, (__ffsll((long long)above) - 1) - zl);
    }
    return best;
}

// Between the coarse bytes and the bit grid sits a COLUMN MAP (round 4): one bit per (x, y) column, "some target lives in this
// column", 64 columns of a grid row per word (128 KB for a 1000 x 1000 map).  A voxel reads the one or two words that cover its
// window in a grid row and visits only the columns whose bit is set -- with 28 k targets scattered over a million columns that is
// 3 of the 121 columns of the reference's window (decay 0.01) instead of all of them; the first version loaded every column's word
// (one dependent L2 round trip each, 1.0 ms at 2 M voxels; then in batches of 8).  Rows are taken from the voxel's own row outwards
// and the scan stops at the first |dx| whose dx^2 cannot beat the best distance found: the minimum over the window is unchanged,
// so are the results.

// PLANNED (avl_heat_plan): the voxels are walked in CELL order -- the plan's one-time radix sort of the map's cells -- instead of in
// voxel-id order.  Voxel ids are in first-touch order (the sampled pixels of a frame in random order), so the 64 voxels of a wave
// sit anywhere in the map and each of their column loads is a transaction of its own; in cell order the lanes of a wave are
// neighbours in z and y, their loads fall on the same or adjacent words (window kernel 3x faster), and a wave is either all near
// a target or not at all.  Coordinates come out of the sorted cell, the mask is gathered and the heat scattered through `order`.
template <bool COARSE_LDS, bool PLANNED>
__global__ __launch_bounds__(256) void heat_window_kernel(const int32_t* __restrict__ pos, const uint8_t* __restrict__ mask,
                                                          int64_t N, int ox, int oy, int oz, int nx, int ny, int nz, int wz, int R,
                                                          const unsigned long long* __restrict__ grid, int cnx, int cny,
                                                          const uint8_t* __restrict__ coarse_g, double cell_size,
                                                          double decay, float* __restrict__ heat,
                                                          const uint32_t* __restrict__ cells, const int32_t* __restrict__ order,
                                                          const unsigned long long* __restrict__ colmap, int cwy) {
    extern __shared__ uint8_t coarse_s[];
    const uint8_t* coarse = coarse_g;
    if (COARSE_LDS) {
        const int nb = cnx * cny;
        for (int k = threadIdx.x * 4; k < nb; k += blockDim.x * 4)                 // (the buffer is padded to a multiple of 4 bytes)
            *reinterpret_cast<uint32_t*>(coarse_s + k) = *reinterpret_cast<const uint32_t*>(coarse_g + k);
        __syncthreads();
        coarse = coarse_s;
    }
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < N; t += (int64_t)gridDim.x * blockDim.x) {
        int64_t i = t;
        int x, y, z;
        if (PLANNED) {
            const uint32_t cell = cells[t];
            i = order[t];
            z = (int)(cell % (uint32_t)nz);
            y = (int)((cell / (uint32_t)nz) % (uint32_t)ny);
            x = (int)(cell / ((uint32_t)nz * (uint32_t)ny));
        } else {
            x = pos[t * 3] - ox, y = pos[t * 3 + 1] - oy, z = pos[t * 3 + 2] - oz;
        }
        if (mask[i]) {
            heat[i] = 1.0f;
            continue;
        }
        const int x0 = max(0, x - R), x1 = min(nx - 1, x + R);
        const int y0 = max(0, y - R), y1 = min(ny - 1, y + R);
        // coarse level: does ANY block of columns that the window touches hold a target?
        bool any = false;
        for (int ca = x0 >> kCoarseShift; ca <= (x1 >> kCoarseShift); ++ca)
            for (int cb = y0 >> kCoarseShift; cb <= (y1 >> kCoarseShift); ++cb) any |= coarse[ca * cny + cb] != 0;
        if (!any) {
            heat[i] = 0.0f;      // every target is farther than R >= cell_size / decay: the heat clips to exactly 0
            continue;
        }
        const int z0 = max(0, z - R), z1 = min(nz - 1, z + R);
        const int w0 = z0 >> 6, w1 = z1 >> 6;
        int best = INT_MAX;
        // one word per column (nz <= 64: every map of the reference, vh = 30): the masks that nearest_bit() derives from (z, z0, z1)
        // are the same for every column of this voxel -- below = bits z0 .. z, above = bits z + 1 .. z1 -- and are built once
        const unsigned long long upto_z = z >= 63 ? ~0ull : ((1ull << (z + 1)) - 1ull);
        const unsigned long long m_below = upto_z & ~((1ull << z0) - 1ull);
        const unsigned long long m_above = ~upto_z & (z1 >= 63 ? ~0ull : ((1ull << (z1 + 1)) - 1ull));
        // window bits of a grid row: bit k = column y0 + k holds a target (the window is at most 33 columns wide)
        const int cw0 = y0 >> 6, cw1 = y1 >> 6, csh = y0 & 63;
        const unsigned long long wmask = (1ull << (y1 - y0 + 1)) - 1ull;
        auto row_bits = [&](int a) -> unsigned long long {
            const unsigned long long* cm = colmap + (size_t)a * cwy;
            unsigned long long bits = cm[cw0] >> csh;
            if (cw1 != cw0) bits |= cm[cw1] << (64 - csh);      // csh > 0 whenever the window straddles two words
            return bits & wmask;
        };
        for (int da = 0; da <= R; ++da) {
            const int dx2 = da * da;
            if (dx2 >= best) break;                             // rows farther out cannot hold a nearer target
            const int a_lo = x - da, a_hi = x + da;
            // both rows at this |dx|: their column words are requested together
            const unsigned long long bits_lo = a_lo >= 0 ? row_bits(a_lo) : 0ull;
            const unsigned long long bits_hi = (da > 0 && a_hi < nx) ? row_bits(a_hi) : 0ull;
#pragma unroll
            for (int side = 0; side < 2; ++side) {
                unsigned long long bits = side ? bits_hi : bits_lo;
                const unsigned long long* grow = grid + (size_t)(side ? a_hi : a_lo) * ny * wz;
                while (bits) {
                    const int b = y0 + __ffsll((long long)bits) - 1;
                    bits &= bits - 1ull;
                    const int dxy2 = dx2 + (b - y) * (b - y);
                    if (dxy2 >= best) continue;
                    if (wz == 1) {
                        const unsigned long long word = grow[b];
                        const unsigned long long below = word & m_below, above = word & m_above;
                        int dz = INT_MAX;
                        if (below) dz = z - (63 - __clzll((long long)below));
                        if (above) dz = min(dz, (__ffsll((long long)above) - 1) - z);
                        if (dz != INT_MAX) best = min(best, dxy2 + dz * dz);
                    } else {
                        const unsigned long long* gp = grow + (size_t)b * wz;
                        for (int w = w0; w <= w1; ++w) {
                            const int dz = nearest_bit(gp[w], w << 6, z, z0, z1);
                            if (dz != INT_MAX) best = min(best, dxy2 + dz * dz);
                        }
                    }
                }
            }
        }
        // a target outside the window is at distance > R >= cell_size/decay: its heat clips to exactly 0
        heat[i] = best == INT_MAX ? 0.0f : heat_from_d2(best, cell_size, decay);
    }
}

__global__ void heat_compact_targets_kernel(const int32_t* __restrict__ pos, const uint8_t* __restrict__ mask, int64_t N,
                                            int32_t* __restrict__ tpos, unsigned long long* __restrict__ count) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
        if (mask[i]) {
            unsigned long long k = atomicAdd(count, 1ull);  // order irrelevant: only the minimum distance is used
            tpos[k * 3] = pos[i * 3];
            tpos[k * 3 + 1] = pos[i * 3 + 1];
            tpos[k * 3 + 2] = pos[i * 3 + 2];
        }
    }
}

constexpr int kHeatTile = 1024;

__global__ __launch_bounds__(256) void heat_brute_kernel(const int32_t* __restrict__ pos, const uint8_t* __restrict__ mask,
                                                         int64_t N, const int32_t* __restrict__ tpos,
                                                         const unsigned long long* __restrict__ count, double cell_size,
                                                         double decay, float* __restrict__ heat) {
    __shared__ int32_t ts[kHeatTile * 3];
    const long long nt = (long long)*count;
    const int64_t nblk_items = (N + blockDim.x - 1) / blockDim.x;
    for (int64_t blk = blockIdx.x; blk < nblk_items; blk += gridDim.x) {
        const int64_t i = blk * blockDim.x + threadIdx.x;
        const bool live = i < N;
        int x = 0, y = 0, z = 0;
        bool is_t = false;
        if (live) {
            x = pos[i * 3]; y = pos[i * 3 + 1]; z = pos[i * 3 + 2];
            is_t = mask[i] != 0;
        }
        long long best = LLONG_MAX;
        for (long long t0 = 0; t0 < nt; t0 += kHeatTile) {
            const int cnt = (int)min((long long)kHeatTile, nt - t0);
            __syncthreads();
            for (int k = threadIdx.x; k < cnt * 3; k += blockDim.x) ts[k] = tpos[t0 * 3 + k];
            __syncthreads();
            if (live && !is_t) {
                for (int k = 0; k < cnt; ++k) {
                    const long long dx = ts[3 * k] - x, dy = ts[3 * k + 1] - y, dz = ts[3 * k + 2] - z;
                    const long long d2 = dx * dx + dy * dy + dz * dz;
                    best = d2 < best ? d2 : best;
                }
            }
        }
        if (live) {
            if (is_t) heat[i] = 1.0f;
            else if (best == LLONG_MAX) heat[i] = 0.0f;  // no target at all: np.argmin of an empty array would raise upstream
            else heat[i] = heat_from_d2(best, cell_size, decay);
        }
    }
}

}  // namespace avl

using namespace avl;

extern "C" int avl_heatmap_from_mask(const int32_t* d_grid_pos, const uint8_t* d_mask, int64_t N, double cell_size,
                                     double decay_rate, float* d_heat, void* stream) {
    AVL_REQUIRE(N >= 0 && cell_size > 0, "avl_heatmap_from_mask: bad arguments");
    if (N == 0) return AVL_OK;
    AVL_REQUIRE(d_grid_pos && d_mask && d_heat, "avl_heatmap_from_mask: null pointer");
    hipStream_t st = as_stream(stream);
    keep_mempool_once();
    int64_t blocks = (N + 255) / 256;
    const int64_t maxb = (int64_t)num_cus() * 8;
    if (blocks > maxb) blocks = maxb;

    // window radius in voxels: heat == 0 once dist_cells / cell_size * decay >= 1
    bool windowed = false;
    int R = 0;
    if (decay_rate > 0) {
        const double r = cell_size / decay_rate;
        if (r < 64.0) {
            R = (int)ceil(r);
            windowed = true;
        }
    }
    int h_bbox[6];
    int* d_bbox = nullptr;
    if (windowed) {
        const int init[6] = {INT_MAX, INT_MAX, INT_MAX, INT_MIN, INT_MIN, INT_MIN};
        AVL_HIP_CHECK(hipMallocAsync((void**)&d_bbox, sizeof(init), st));
        AVL_HIP_CHECK(hipMemcpyAsync(d_bbox, init, sizeof(init), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(heat_bbox_kernel, dim3((unsigned)std::min<int64_t>(blocks, 512)), dim3(256), 0, st, d_grid_pos, N, d_bbox);
        AVL_HIP_CHECK(hipMemcpyAsync(h_bbox, d_bbox, sizeof(h_bbox), hipMemcpyDeviceToHost, st));
        AVL_HIP_CHECK(hipStreamSynchronize(st));
        (void)hipFreeAsync(d_bbox, st);
        const double nx = (double)h_bbox[3] - h_bbox[0] + 1, ny = (double)h_bbox[4] - h_bbox[1] + 1,
                     nz = (double)h_bbox[5] - h_bbox[2] + 1;
        const double cells = nx * ny * nz;
        const double window = (2.0 * R + 1) * (2.0 * R + 1) * (2.0 * R + 1);
        if (cells > 3.2e10 || window > 40000.0) windowed = false;   // bit grid: 1 bit per cell
    }
    if (windowed) {
        const int nx = h_bbox[3] - h_bbox[0] + 1, ny = h_bbox[4] - h_bbox[1] + 1, nz = h_bbox[5] - h_bbox[2] + 1;
        const int wz = (nz + 63) / 64;
        const size_t words = (size_t)nx * ny * wz;
        unsigned long long* grid = nullptr;
        const int cnx = (nx >> kCoarseShift) + 1, cny = (ny >> kCoarseShift) + 1;
        const size_t cbytes = ((size_t)cnx * cny + 3) & ~(size_t)3;
        const size_t gbytes = words * sizeof(unsigned long long);
        const int cwy = (ny + 63) / 64;
        const size_t colbytes = (size_t)nx * cwy * sizeof(unsigned long long);
        AVL_HIP_CHECK(hipMallocAsync((void**)&grid, gbytes + colbytes + cbytes, st));          // bit grid | column map | coarse byte grid
        AVL_HIP_CHECK(hipMemsetAsync(grid, 0, gbytes + colbytes + cbytes, st));
        unsigned long long* colmap = grid + words;
        uint8_t* coarse = reinterpret_cast<uint8_t*>(grid) + gbytes + colbytes;
        hipLaunchKernelGGL(heat_scatter_kernel, dim3((unsigned)blocks), dim3(256), 0, st, d_grid_pos, d_mask, N, h_bbox[0],
                           h_bbox[1], h_bbox[2], ny, wz, grid, cny, coarse, colmap, cwy);
        if (cbytes <= (size_t)kCoarseLdsBytes)
            hipLaunchKernelGGL((heat_window_kernel<true, false>), dim3((unsigned)blocks), dim3(256), cbytes, st, d_grid_pos, d_mask, N, h_bbox[0],
                               h_bbox[1], h_bbox[2], nx, ny, nz, wz, R, grid, cnx, cny, coarse, cell_size, decay_rate, d_heat, nullptr, nullptr,
                               colmap, cwy);
        else
            hipLaunchKernelGGL((heat_window_kernel<false, false>), dim3((unsigned)blocks), dim3(256), 0, st, d_grid_pos, d_mask, N, h_bbox[0],
                               h_bbox[1], h_bbox[2], nx, ny, nz, wz, R, grid, cnx, cny, coarse, cell_size, decay_rate, d_heat, nullptr, nullptr,
                               colmap, cwy);
        (void)hipFreeAsync(grid, st);
    } else {
        int32_t* tpos = nullptr;
        unsigned long long* cnt = nullptr;
        AVL_HIP_CHECK(hipMallocAsync((void**)&tpos, (size_t)N * 3 * sizeof(int32_t), st));
        AVL_HIP_CHECK(hipMallocAsync((void**)&cnt, sizeof(unsigned long long), st));
        AVL_HIP_CHECK(hipMemsetAsync(cnt, 0, sizeof(unsigned long long), st));
        hipLaunchKernelGGL(heat_compact_targets_kernel, dim3((unsigned)blocks), dim3(256), 0, st, d_grid_pos, d_mask, N, tpos, cnt);
        hipLaunchKernelGGL(heat_brute_kernel, dim3((unsigned)blocks), dim3(256), 0, st, d_grid_pos, d_mask, N, tpos, cnt, cell_size,
                           decay_rate, d_heat);
        (void)hipFreeAsync(tpos, st);
        (void)hipFreeAsync(cnt, st);
    }
    AVL_HIP_CHECK(hipGetLastError());
    return AVL_OK;
}

namespace avl {
__global__ void heat_cells_kernel(const int32_t* __restrict__ pos, int64_t N, int ox, int oy, int oz, int ny, int nz,
                                  uint32_t* __restrict__ cells, int32_t* __restrict__ iota) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
        cells[i] = ((uint32_t)(pos[i * 3] - ox) * (uint32_t)ny + (uint32_t)(pos[i * 3 + 1] - oy)) * (uint32_t)nz + (uint32_t)(pos[i * 3 + 2] - oz);
        iota[i] = (int32_t)i;
    }
}

// targets into the bit grid + the coarse grid, from the plan's sorted cells (mask gathered through `order`)
__global__ void heat_scatter_planned_kernel(const uint32_t* __restrict__ cells, const int32_t* __restrict__ order,
                                            const uint8_t* __restrict__ mask, int64_t N, int ny, int nz, int wz,
                                            unsigned long long* __restrict__ grid, int cny, uint8_t* __restrict__ coarse,
                                            unsigned long long* __restrict__ colmap, int cwy) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < N; t += (int64_t)gridDim.x * blockDim.x) {
        if (mask[order[t]]) {
            const uint32_t cell = cells[t];
            const int z = (int)(cell % (uint32_t)nz), y = (int)((cell / (uint32_t)nz) % (uint32_t)ny), x = (int)(cell / ((uint32_t)nz * (uint32_t)ny));
            atomicOr(&grid[((size_t)x * ny + y) * wz + (z >> 6)], 1ull << (z & 63));
            atomicOr(&colmap[(size_t)x * cwy + (y >> 6)], 1ull << (y & 63));
            coarse[(size_t)(x >> kCoarseShift) * cny + (y >> kCoarseShift)] = 1;
        }
    }
}
}  // namespace avl

struct avl_heat_plan {
    const int32_t* d_grid_pos = nullptr;    // caller-owned, must outlive the plan (the brute-force fall-back reads it)
    int64_t N = 0;
    int dev = 0;
    int bbox[6] = {0, 0, 0, 0, 0, 0};
    int nx = 0, ny = 0, nz = 0, wz = 0, cnx = 0, cny = 0, cwy = 0;
    size_t gbytes = 0, colbytes = 0, cbytes = 0;
    uint32_t* cells = nullptr;               // (N,) linear cells inside the bounding box, ascending
    int32_t* order = nullptr;                // (N,) voxel id of the t-th cell
    unsigned long long* grid = nullptr;      // bit grid | column map | coarse byte grid, zeroed per call
};

extern "C" int avl_heat_plan_destroy(avl_heat_plan* p) {
    if (!p) return AVL_OK;
    (void)hipFree(p->cells);
    (void)hipFree(p->order);
    (void)hipFree(p->grid);
    delete p;
    return AVL_OK;
}

extern "C" int avl_heat_plan_create(avl_heat_plan** h_out, const int32_t* d_grid_pos, int64_t N, void* stream) {
    AVL_REQUIRE(h_out, "avl_heat_plan_create: null output");
    *h_out = nullptr;
    AVL_REQUIRE(N > 0 && N < (1ll << 31) && d_grid_pos, "avl_heat_plan_create: bad arguments");
    hipStream_t st = as_stream(stream);
    keep_mempool_once();
    int64_t blocks = std::min<int64_t>((N + 255) / 256, (int64_t)num_cus() * 8);
    const int init[6] = {INT_MAX, INT_MAX, INT_MAX, INT_MIN, INT_MIN, INT_MIN};
    int* d_bbox = nullptr;
    AVL_HIP_CHECK(hipMallocAsync((void**)&d_bbox, sizeof(init), st));
    AVL_HIP_CHECK(hipMemcpyAsync(d_bbox, init, sizeof(init), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(heat_bbox_kernel, dim3((unsigned)std::min<int64_t>(blocks, 512)), dim3(256), 0, st, d_grid_pos, N, d_bbox);
    avl_heat_plan* p = new avl_heat_plan();
    p->d_grid_pos = d_grid_pos;
    p->N = N;
    (void)hipGetDevice(&p->dev);
    hipError_t e = hipMemcpyAsync(p->bbox, d_bbox, sizeof(p->bbox), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFreeAsync(d_bbox, st);
    if (e != hipSuccess) {
        delete p;
        set_error("avl_heat_plan_create: %s", hipGetErrorString(e));
        return AVL_ERR_HIP;
    }
    p->nx = p->bbox[3] - p->bbox[0] + 1, p->ny = p->bbox[4] - p->bbox[1] + 1, p->nz = p->bbox[5] - p->bbox[2] + 1;
    const double cells_d = (double)p->nx * p->ny * p->nz;
    if (cells_d >= 4294967296.0) {       // the cell order needs 32-bit cells: the stateless call handles such maps
        delete p;
        set_error("avl_heat_plan_create: the map's bounding box has %.0f cells (>= 2^32); use avl_heatmap_from_mask", cells_d);
        return AVL_ERR_INVALID;
    }
    p->wz = (p->nz + 63) / 64;
    p->cnx = (p->nx >> kCoarseShift) + 1, p->cny = (p->ny >> kCoarseShift) + 1;
    p->gbytes = (size_t)p->nx * p->ny * p->wz * sizeof(unsigned long long);
    p->cbytes = ((size_t)p->cnx * p->cny + 3) & ~(size_t)3;
    p->cwy = (p->ny + 63) / 64;
    p->colbytes = (size_t)p->nx * p->cwy * sizeof(unsigned long long);
    uint32_t* unsorted = nullptr;
    int32_t* iota = nullptr;
    void* tmp = nullptr;
    size_t tmp_bytes = 0;
    int bits = 1;
    while (bits < 32 && (double)(1ull << bits) < cells_d) ++bits;
    e = hipMalloc((void**)&p->cells, (size_t)N * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void**)&p->order, (size_t)N * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc((void**)&p->grid, p->gbytes + p->colbytes + p->cbytes);
    if (e == hipSuccess) e = hipMallocAsync((void**)&unsorted, (size_t)N * sizeof(uint32_t), st);
    if (e == hipSuccess) e = hipMallocAsync((void**)&iota, (size_t)N * sizeof(int32_t), st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(heat_cells_kernel, dim3((unsigned)blocks), dim3(256), 0, st, d_grid_pos, N, p->bbox[0], p->bbox[1], p->bbox[2], p->ny,
                           p->nz, unsorted, iota);
        e = rocprim::radix_sort_pairs(nullptr, tmp_bytes, unsorted, p->cells, iota, p->order, (size_t)N, 0, bits, st);
    }
    if (e == hipSuccess) e = hipMallocAsync(&tmp, tmp_bytes ? tmp_bytes : 16, st);
    if (e == hipSuccess) e = rocprim::radix_sort_pairs(tmp, tmp_bytes, unsorted, p->cells, iota, p->order, (size_t)N, 0, bits, st);
    (void)hipFreeAsync(tmp, st);
    (void)hipFreeAsync(iota, st);
    (void)hipFreeAsync(unsorted, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        (void)avl_heat_plan_destroy(p);
        set_error("avl_heat_plan_create: %s", hipGetErrorString(e));
        return AVL_ERR_HIP;
    }
    *h_out = p;
    return AVL_OK;
}

extern "C" int avl_heatmap_from_mask_planned(avl_heat_plan* p, const uint8_t* d_mask, double cell_size, double decay_rate, float* d_heat,
                                             void* stream) {
    AVL_REQUIRE(p, "avl_heatmap_from_mask_planned: null plan");
    AVL_REQUIRE(cell_size > 0 && d_mask && d_heat, "avl_heatmap_from_mask_planned: bad arguments");
    hipStream_t st = as_stream(stream);
    int R = 0;
    bool windowed = false;
    if (decay_rate > 0) {
        const double r = cell_size / decay_rate;
        if (r < 64.0) {
            R = (int)ceil(r);
            windowed = (2.0 * R + 1) * (2.0 * R + 1) * (2.0 * R + 1) <= 40000.0;
        }
    }
    if (!windowed)      // a window larger than the target list: the stateless call's brute force over LDS-staged targets
        return avl_heatmap_from_mask(p->d_grid_pos, d_mask, p->N, cell_size, decay_rate, d_heat, stream);
    const int64_t blocks = std::min<int64_t>((p->N + 255) / 256, (int64_t)num_cus() * 8);
    AVL_HIP_CHECK(hipMemsetAsync(p->grid, 0, p->gbytes + p->colbytes + p->cbytes, st));
    unsigned long long* colmap = p->grid + p->gbytes / sizeof(unsigned long long);
    uint8_t* coarse = reinterpret_cast<uint8_t*>(p->grid) + p->gbytes + p->colbytes;
    hipLaunchKernelGGL(heat_scatter_planned_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p->cells, p->order, d_mask, p->N, p->ny, p->nz,
                       p->wz, p->grid, p->cny, coarse, colmap, p->cwy);
    if (p->cbytes <= (size_t)kCoarseLdsBytes)
        hipLaunchKernelGGL((heat_window_kernel<true, true>), dim3((unsigned)blocks), dim3(256), p->cbytes, st, nullptr, d_mask, p->N, p->bbox[0],
                           p->bbox[1], p->bbox[2], p->nx, p->ny, p->nz, p->wz, R, p->grid, p->cnx, p->cny, coarse, cell_size, decay_rate, d_heat,
                           p->cells, p->order, colmap, p->cwy);
    else
        hipLaunchKernelGGL((heat_window_kernel<false, true>), dim3((unsigned)blocks), dim3(256), 0, st, nullptr, d_mask, p->N, p->bbox[0],
                           p->bbox[1], p->bbox[2], p->nx, p->ny, p->nz, p->wz, R, p->grid, p->cnx, p->cny, coarse, cell_size, decay_rate, d_heat,
                           p->cells, p->order, colmap, p->cwy);
    AVL_HIP_CHECK(hipGetLastError());
    return AVL_OK;
}

namespace avl {
__global__ void iota64_kernel(int64_t* __restrict__ v, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) v[i] = i;
}

// ---- wave-level top-k (k <= 64) ------------------------------------------------------------------------------------------
// Order: value descending, equal values by ascending index, NaN last -- the order of np.argsort(-v, kind="stable").
// key = order-preserving unsigned image of the float (+0 and -0 coincide, NaN -> 0 = below -inf).
__device__ __forceinline__ uint32_t topk_key(float v) {
    if (v != v) return 0u;
    if (v == 0.f) v = 0.f;
    const uint32_t u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// A wave keeps its k best (key, index) pairs sorted, ONE PER LANE (lane i = rank i).  A chunk of 64 candidates is tested against
// the current k-th entry with one compare + ballot -- after the first few chunks almost nothing passes -- and the few that do
// are inserted by rank: rank = popcount(ballot(entry better than candidate)), lanes behind it shift up by one (one shuffle).
struct WaveTopK {
    uint32_t key;
    int32_t idx;
    __device__ void init() { key = 0u; idx = INT_MAX; }   // sentinel: worse than any real element (NaN has key 0, index < INT_MAX)
    __device__ static bool better(uint32_t ka, int32_t ia, uint32_t kb, int32_t ib) { return ka > kb || (ka == kb && ia < ib); }
    __device__ void offer(uint32_t ck, int32_t ci, bool valid, int k, int lane) {
        uint32_t tk = (uint32_t)__shfl((int)key, k - 1, 64);
        int32_t ti = __shfl(idx, k - 1, 64);
        unsigned long long pass = __ballot(valid && better(ck, ci, tk, ti));
        while (pass) {
            const int src = __ffsll((long long)pass) - 1;
            pass &= pass - 1;
            const uint32_t k1 = (uint32_t)__shfl((int)ck, src, 64);
            const int32_t i1 = __shfl(ci, src, 64);
            if (!better(k1, i1, tk, ti)) continue;             // the threshold rose since the ballot
            const int pos = __popcll(__ballot(lane < k && better(key, idx, k1, i1)));
            const uint32_t upk = (uint32_t)__shfl_up((int)key, 1, 64);
            const int32_t upi = __shfl_up(idx, 1, 64);
            if (lane == pos) { key = k1; idx = i1; }
            else if (lane > pos) { key = upk; idx = upi; }
            tk = (uint32_t)__shfl((int)key, k - 1, 64);
            ti = __shfl(idx, k - 1, 64);
        }
    }
};

// phase 1: every wave selects the k best of its strided share of the vector and writes them (64 slots per wave)
__global__ __launch_bounds__(256) void topk_partial_kernel(const float* __restrict__ v, int64_t N, int k, uint32_t* __restrict__ ckey,
                                                           int32_t* __restrict__ cidx) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    WaveTopK t;
    t.init();
    for (int64_t base = wave * 64; base < N; base += nwaves * 64) {
        const int64_t i = base + lane;
        const bool valid = i < N;
        t.offer(valid ? topk_key(v[i]) : 0u, valid ? (int32_t)i : INT_MAX, valid, k, lane);
    }
    ckey[wave * 64 + lane] = t.key;
    cidx[wave * 64 + lane] = t.idx;
}

// phase 2: one wave merges the candidates of all waves and writes the final k (index, value)
__global__ __launch_bounds__(64) void topk_merge_kernel(const float* __restrict__ v, const uint32_t* __restrict__ ckey,
                                                        const int32_t* __restrict__ cidx, int64_t ncand, int k,
                                                        int64_t* __restrict__ out_idx, float* __restrict__ out_val) {
    const int lane = threadIdx.x;
    WaveTopK t;
    t.init();
    for (int64_t base = 0; base < ncand; base += 64) {
        const int64_t i = base + lane;
        const bool valid = i < ncand && cidx[i] != INT_MAX;
        t.offer(valid ? ckey[i] : 0u, valid ? cidx[i] : INT_MAX, valid, k, lane);
    }
    if (lane < k) {
        out_idx[lane] = t.idx;
        out_val[lane] = t.idx != INT_MAX ? v[t.idx] : 0.f;
    }
}
}  // namespace avl

// k largest values of a float32 vector with their indices, descending; equal values keep ascending index order and NaN comes
// last (the order np.argsort(-v, kind="stable") gives).  The k = 1 case is the navigator's goal voxel
// (habitat_lang_robot.py:427-430).  k <= 64: wave-level selection, two small launches, no allocation (library scratch);
// larger k: a full radix sort.
extern "C" int avl_topk_f32(const float* d_vals, int64_t N, int k, int64_t* h_index, float* h_value, void* stream) {
    AVL_REQUIRE(d_vals && N > 0 && k > 0 && k <= N, "avl_topk_f32: bad arguments (N=%lld k=%d)", (long long)N, k);
    AVL_REQUIRE(N < (1ll << 31), "avl_topk_f32: N must fit 31 bits");
    hipStream_t st = as_stream(stream);
    if (k <= 64) {
        int64_t blocks = (N + 64 * 32 - 1) / (64 * 32) / 4;       // >= 32 chunks per wave, 4 waves per block
        if (blocks > 256) blocks = 256;
        if (blocks < 1) blocks = 1;
        const int64_t nwaves = blocks * 4, ncand = nwaves * 64;
        char* sc = static_cast<char*>(avl::scratch((size_t)ncand * 8 + 64 * 12));
        if (!sc) return AVL_ERR_HIP;
        uint32_t* ckey = reinterpret_cast<uint32_t*>(sc);
        int32_t* cidx = reinterpret_cast<int32_t*>(sc + (size_t)ncand * 4);
        int64_t* oidx = reinterpret_cast<int64_t*>(sc + (size_t)ncand * 8);
        float* oval = reinterpret_cast<float*>(sc + (size_t)ncand * 8 + 64 * 8);
        hipLaunchKernelGGL(avl::topk_partial_kernel, dim3((unsigned)blocks), dim3(256), 0, st, d_vals, N, k, ckey, cidx);
        hipLaunchKernelGGL(avl::topk_merge_kernel, dim3(1), dim3(64), 0, st, d_vals, ckey, cidx, ncand, k, oidx, oval);
        AVL_HIP_CHECK(hipGetLastError());
        if (h_index) AVL_HIP_CHECK(hipMemcpyAsync(h_index, oidx, (size_t)k * sizeof(int64_t), hipMemcpyDeviceToHost, st));
        if (h_value) AVL_HIP_CHECK(hipMemcpyAsync(h_value, oval, (size_t)k * sizeof(float), hipMemcpyDeviceToHost, st));
        AVL_HIP_CHECK(hipStreamSynchronize(st));
        return AVL_OK;
    }
    float* keys_out = nullptr;
    int64_t *iota = nullptr, *order = nullptr;
    void* tmp = nullptr;
    size_t tmp_bytes = 0;
    AVL_HIP_CHECK(hipMallocAsync((void**)&keys_out, (size_t)N * sizeof(float), st));
    AVL_HIP_CHECK(hipMallocAsync((void**)&iota, (size_t)N * sizeof(int64_t), st));
    AVL_HIP_CHECK(hipMallocAsync((void**)&order, (size_t)N * sizeof(int64_t), st));
    hipLaunchKernelGGL(avl::iota64_kernel, dim3((unsigned)((N + 255) / 256 > 4096 ? 4096 : (N + 255) / 256)), dim3(256), 0, st, iota, N);
    AVL_HIP_CHECK(rocprim::radix_sort_pairs_desc(nullptr, tmp_bytes, d_vals, keys_out, iota, order, (size_t)N, 0, 32, st));
    AVL_HIP_CHECK(hipMallocAsync(&tmp, tmp_bytes ? tmp_bytes : 16, st));
    AVL_HIP_CHECK(rocprim::radix_sort_pairs_desc(tmp, tmp_bytes, d_vals, keys_out, iota, order, (size_t)N, 0, 32, st));
    if (h_index) AVL_HIP_CHECK(hipMemcpyAsync(h_index, order, (size_t)k * sizeof(int64_t), hipMemcpyDeviceToHost, st));
    if (h_value) AVL_HIP_CHECK(hipMemcpyAsync(h_value, keys_out, (size_t)k * sizeof(float), hipMemcpyDeviceToHost, st));
    AVL_HIP_CHECK(hipStreamSynchronize(st));
    (void)hipFreeAsync(tmp, st); (void)hipFreeAsync(order, st); (void)hipFreeAsync(iota, st); (void)hipFreeAsync(keys_out, st);
    return AVL_OK;
}
