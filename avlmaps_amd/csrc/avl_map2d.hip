// Top-down 2-D products of the voxel map for gfx950: the consumers that iterate over all N voxels in Python upstream.
//
// Replaces (upstream reference, path:line):
//   avlmaps/utils/visualize_utils.py:77-83   pool_3d_label_to_2d            (Python loop over N voxels)
//   avlmaps/map/map.py:79-95                 Map.generate_obstacle_map      (dense (gs, gs, vh) reduction)
//   avlmaps/map/map.py:106-113               Map.generate_rgb_topdown_map   (Python loop, LAST voxel of a column wins)
//   avlmaps/utils/index_utils.py:163-177     get_dynamic_obstacles_map_3d   (class filter + scatter + mask logic)
// All of them are scatters of a few bytes per voxel into a (gs, gs) image: HBM-latency bound, one pass over grid_pos
// (12 B / voxel) plus the per-voxel operand; they are fed by device-resident arrays (the similarity kernel's argmax never
// leaves HBM) and only the small 2-D result goes back to the host.
#include "avl_common.h"

namespace avl {

__global__ __launch_bounds__(256) void pool_label_kernel(const int32_t* __restrict__ pos, const uint8_t* __restrict__ mask, int64_t N,
                                                         int gs, uint8_t* __restrict__ out, int* __restrict__ err) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
        if (!mask[i]) continue;
        int r = pos[3 * i], c = pos[3 * i + 1];
        if (r < 0) r += gs;     // numpy index wrap
        if (c < 0) c += gs;
        if (r < 0 || r >= gs || c < 0 || c >= gs) { atomicOr(err, 1); continue; }   // IndexError upstream
        out[(int64_t)r * gs + c] = 1;   // mask_2d[row, col] = mask_3d[i] or mask_2d[row, col]: idempotent store, no atomics
    }
}

// pass 1 of the colour map: the voxel with the LARGEST index writes last in the reference's sequential loop
__global__ __launch_bounds__(256) void topdown_winner_kernel(const int32_t* __restrict__ pos, int64_t N, int gs,
                                                             int32_t* __restrict__ winner, int* __restrict__ err) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
        int r = pos[3 * i], c = pos[3 * i + 1];
        if (r < 0) r += gs;
        if (c < 0) c += gs;
        if (r < 0 || r >= gs || c < 0 || c >= gs) { atomicOr(err, 1); continue; }
        atomicMax(&winner[(int64_t)r * gs + c], (int32_t)i);
    }
}

__global__ __launch_bounds__(256) void topdown_paint_kernel(const int32_t* __restrict__ winner, const uint8_t* __restrict__ rgb,
                                                            int64_t cells, uint8_t* __restrict__ out) {
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < cells; p += (int64_t)gridDim.x * blockDim.x) {
        const int32_t w = winner[p];
        out[3 * p + 0] = w >= 0 ? rgb[3 * (int64_t)w + 0] : 0;
        out[3 * p + 1] = w >= 0 ? rgb[3 * (int64_t)w + 1] : 0;
        out[3 * p + 2] = w >= 0 ? rgb[3 * (int64_t)w + 2] : 0;
    }
}

// free[r, c] = no voxel id > 0 in heights [h0, h1)   (map.py:92: np.sum(occupied_ids[..., height_mask] > 0, axis=2) == 0;
// note `> 0`: voxel id 0 does not count upstream)
__global__ __launch_bounds__(256) void obstacle_map_kernel(const int32_t* __restrict__ occ, int64_t cells, int vh, int h0, int h1,
                                                           uint8_t* __restrict__ free_map) {
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < cells; p += (int64_t)gridDim.x * blockDim.x) {
        const int32_t* col = occ + p * vh;
        bool any = false;
        for (int h = h0; h < h1; ++h) any |= col[h] > 0;
        free_map[p] = any ? 0 : 1;
    }
}

// new_obstacles[pos0 - rmin, pos1 - cmin] = 1 for voxels whose class is an obstacle class; then
// out = not (new_obstacles and obstacles_cropped == 0)   (index_utils.py:163-177).  hit must be zero-initialised.
__global__ __launch_bounds__(256) void obstacle_scatter_kernel(const int32_t* __restrict__ pos, const int32_t* __restrict__ cls, int64_t N,
                                                               const uint8_t* __restrict__ is_obstacle, int Q, int rmin, int cmin,
                                                               int H, int W, uint8_t* __restrict__ hit, int* __restrict__ err) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t k = cls[i];
        if (k < 0 || k >= Q || !is_obstacle[k]) continue;
        int r = pos[3 * i] - rmin, c = pos[3 * i + 1] - cmin;
        if (r < 0) r += H;      // numpy wraps negative indices (a voxel outside the crop lands on the far side upstream too)
        if (c < 0) c += W;
        if (r < 0 || r >= H || c < 0 || c >= W) { atomicOr(err, 1); continue; }   // IndexError upstream
        hit[(int64_t)r * W + c] = 1;
    }
}

__global__ __launch_bounds__(256) void obstacle_combine_kernel(const uint8_t* __restrict__ hit, const uint8_t* __restrict__ cropped_free,
                                                               int64_t cells, uint8_t* __restrict__ out_free) {
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < cells; p += (int64_t)gridDim.x * blockDim.x)
        out_free[p] = (hit[p] && cropped_free[p] == 0) ? 0 : 1;
}

static unsigned grid_for(int64_t n) {
    int64_t b = (n + 255) / 256;
    const int64_t maxb = (int64_t)num_cus() * 16;
    if (b > maxb) b = maxb;
    return (unsigned)(b < 1 ? 1 : b);
}

struct ErrFlag {
    int* d = nullptr;
    hipStream_t st;
    int init(hipStream_t s) {
        st = s;
        AVL_HIP_CHECK(hipMallocAsync((void**)&d, sizeof(int), st));
        AVL_HIP_CHECK(hipMemsetAsync(d, 0, sizeof(int), st));
        return AVL_OK;
    }
    int finish(const char* what) {   // synchronises
        int h = 0;
        AVL_HIP_CHECK(hipMemcpyAsync(&h, d, sizeof(int), hipMemcpyDeviceToHost, st));
        AVL_HIP_CHECK(hipStreamSynchronize(st));
        (void)hipFreeAsync(d, st);
        if (h) {
            set_error("%s: a voxel position indexes outside the 2-D map (the reference raises IndexError here)", what);
            return AVL_ERR_INVALID;
        }
        return AVL_OK;
    }
};

}  // namespace avl

using namespace avl;

extern "C" {

int avl_pool_label_2d(const int32_t* d_grid_pos, const uint8_t* d_mask, int64_t N, int gs, uint8_t* d_mask2d, void* stream) {
    AVL_REQUIRE(N >= 0 && gs > 0 && d_mask2d, "avl_pool_label_2d: bad arguments");
    hipStream_t st = as_stream(stream);
    AVL_HIP_CHECK(hipMemsetAsync(d_mask2d, 0, (size_t)gs * gs, st));
    if (N == 0) return AVL_OK;
    AVL_REQUIRE(d_grid_pos && d_mask, "avl_pool_label_2d: null input");
    ErrFlag ef;
    int rc = ef.init(st);
    if (rc != AVL_OK) return rc;
    hipLaunchKernelGGL(pool_label_kernel, dim3(grid_for(N)), dim3(256), 0, st, d_grid_pos, d_mask, N, gs, d_mask2d, ef.d);
    AVL_HIP_CHECK(hipGetLastError());
    return ef.finish("avl_pool_label_2d");
}

int avl_rgb_topdown(const int32_t* d_grid_pos, const uint8_t* d_grid_rgb, int64_t N, int gs, uint8_t* d_rgb2d, void* stream) {
    AVL_REQUIRE(N >= 0 && N < (1ll << 31) && gs > 0 && d_rgb2d, "avl_rgb_topdown: bad arguments");
    hipStream_t st = as_stream(stream);
    const int64_t cells = (int64_t)gs * gs;
    int32_t* winner = nullptr;
    AVL_HIP_CHECK(hipMallocAsync((void**)&winner, (size_t)cells * sizeof(int32_t), st));
    AVL_HIP_CHECK(hipMemsetAsync(winner, 0xFF, (size_t)cells * sizeof(int32_t), st));
    ErrFlag ef;
    int rc = ef.init(st);
    if (rc != AVL_OK) return rc;
    if (N > 0) {
        AVL_REQUIRE(d_grid_pos && d_grid_rgb, "avl_rgb_topdown: null input");
        hipLaunchKernelGGL(topdown_winner_kernel, dim3(grid_for(N)), dim3(256), 0, st, d_grid_pos, N, gs, winner, ef.d);
    }
    hipLaunchKernelGGL(topdown_paint_kernel, dim3(grid_for(cells)), dim3(256), 0, st, winner, d_grid_rgb, cells, d_rgb2d);
    AVL_HIP_CHECK(hipGetLastError());
    (void)hipFreeAsync(winner, st);
    return ef.finish("avl_rgb_topdown");
}

int avl_obstacle_map(const int32_t* d_occupied_ids, int n0, int n1, int vh, int h_begin, int h_end, uint8_t* d_free, void* stream) {
    AVL_REQUIRE(n0 > 0 && n1 > 0 && vh > 0 && d_occupied_ids && d_free, "avl_obstacle_map: bad arguments");
    AVL_REQUIRE(h_begin >= 0 && h_end <= vh && h_begin <= h_end, "avl_obstacle_map: bad height range [%d, %d)", h_begin, h_end);
    const int64_t cells = (int64_t)n0 * n1;
    hipLaunchKernelGGL(obstacle_map_kernel, dim3(grid_for(cells)), dim3(256), 0, as_stream(stream), d_occupied_ids, cells, vh, h_begin, h_end,
                       d_free);
    AVL_HIP_CHECK(hipGetLastError());
    return AVL_OK;
}

int avl_obstacle_scatter(const int32_t* d_grid_pos, const int32_t* d_class, int64_t N, const uint8_t* h_class_is_obstacle, int Q,
                         int rmin, int cmin, int H, int W, const uint8_t* d_cropped_free, uint8_t* d_out_free, void* stream) {
    AVL_REQUIRE(N >= 0 && Q > 0 && Q <= 4096 && H > 0 && W > 0 && h_class_is_obstacle && d_cropped_free && d_out_free,
                "avl_obstacle_scatter: bad arguments");
    hipStream_t st = as_stream(stream);
    const int64_t cells = (int64_t)H * W;
    uint8_t *hit = nullptr, *tab = nullptr;
    AVL_HIP_CHECK(hipMallocAsync((void**)&hit, (size_t)cells, st));
    AVL_HIP_CHECK(hipMemsetAsync(hit, 0, (size_t)cells, st));
    AVL_HIP_CHECK(hipMallocAsync((void**)&tab, (size_t)Q, st));
    AVL_HIP_CHECK(hipMemcpyAsync(tab, h_class_is_obstacle, (size_t)Q, hipMemcpyHostToDevice, st));   // pageable: staged before return
    ErrFlag ef;
    int rc = ef.init(st);
    if (rc != AVL_OK) return rc;
    if (N > 0) {
        AVL_REQUIRE(d_grid_pos && d_class, "avl_obstacle_scatter: null input");
        hipLaunchKernelGGL(obstacle_scatter_kernel, dim3(grid_for(N)), dim3(256), 0, st, d_grid_pos, d_class, N, tab, Q, rmin, cmin, H, W, hit,
                           ef.d);
    }
    hipLaunchKernelGGL(obstacle_combine_kernel, dim3(grid_for(cells)), dim3(256), 0, st, hit, d_cropped_free, cells, d_out_free);
    AVL_HIP_CHECK(hipGetLastError());
    (void)hipFreeAsync(hit, st);
    (void)hipFreeAsync(tab, st);
    return ef.finish("avl_obstacle_scatter");
}

}  // extern "C"
