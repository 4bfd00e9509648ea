// Sliding-window merge of the pixel-feature extractor's output, on the device, channels-last.
//
// Replaces (upstream reference, path:line):
//   avlmaps/utils/lseg_utils.py:61-102   the accumulation half of get_lseg_feat: outputs[:, :, h0:h1, w0:w1] += crop(output),
//                                        count_norm += 1 per window, outputs / count_norm, crop to (height, width), and the
//                                        369 MB device-to-host copy of the (1, D, Hf, Wf) result per frame
//
// The model (LSegEncNet on PyTorch-ROCm) is called ONCE on the batch of G windows and hands back (G, D, crop, crop).  This kernel
// reads every window element once and writes the averaged map once, already in the layout the builder's gather wants: (Hf, Wf, D)
// channels-last float32 (one sampled point = one contiguous 2 KB row).  The NCHW -> HWC transposition goes through a 64 x 64 LDS
// tile: loads are coalesced along x (the windows' fastest axis), stores along the channel axis.  Overlapping windows are summed in
// window order starting from zero and divided by their count: the reference's float32 arithmetic, bit for bit.
#include "avl_common.h"

namespace avl {

constexpr int kMaxWindows = 64;

struct WindowSet {
    int G;
    int h0[kMaxWindows], w0[kMaxWindows];
};

template <typename T>
__global__ __launch_bounds__(256) void lseg_merge_windows_kernel(const T* __restrict__ win, WindowSet ws, int D, int crop, int height, int width,
                                                                 float* __restrict__ out) {
    __shared__ float tile[64][65];
    const int x0 = blockIdx.x * 64, y = blockIdx.y, c0 = blockIdx.z * 64;
    const int xl = threadIdx.x & 63, cq = threadIdx.x >> 6;
    const int x = x0 + xl;
    float acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.f;
    int cnt = 0;
    for (int g = 0; g < ws.G; ++g) {
        const int yy = y - ws.h0[g], xx = x - ws.w0[g];
        if (yy < 0 || yy >= crop) continue;                    // block-uniform
        if (x < width && xx >= 0 && xx < crop) {
            const T* p = win + (((size_t)g * D + c0 + cq) * crop + yy) * (size_t)crop + xx;
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (c0 + cq + 4 * k < D) acc[k] += (float)p[(size_t)4 * k * crop * crop];
            ++cnt;
        }
    }
    const float n = (float)cnt;
#pragma unroll
    for (int k = 0; k < 16; ++k) tile[cq + 4 * k][xl] = cnt ? __fdiv_rn(acc[k], n) : 0.f;
    __syncthreads();
    const int cl = threadIdx.x & 63, xq = threadIdx.x >> 6;
    if (c0 + cl < D) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int xo = x0 + xq + 4 * k;
            if (xo < width) out[((size_t)y * width + xo) * D + c0 + cl] = tile[cl][xq + 4 * k];
        }
    }
}

}  // namespace avl

using namespace avl;

extern "C" int avl_lseg_merge_windows(const void* d_win, int is_f16, int G, int D, int crop, const int32_t* h_origin, int height, int width,
                                      float* d_out, void* stream) {
    AVL_REQUIRE(G > 0 && G <= kMaxWindows && D > 0 && crop > 0 && height > 0 && width > 0,
                "avl_lseg_merge_windows: bad shape (G=%d of at most %d windows, D=%d, crop=%d, %dx%d)", G, kMaxWindows, D, crop, height, width);
    AVL_REQUIRE(d_win && h_origin && d_out, "avl_lseg_merge_windows: null pointer");
    WindowSet ws;
    ws.G = G;
    for (int g = 0; g < G; ++g) {
        ws.h0[g] = h_origin[2 * g];
        ws.w0[g] = h_origin[2 * g + 1];
    }
    // every output pixel must be covered by a window (the reference asserts count_norm != 0, lseg_utils.py:97).  If the union of
    // the windows leaves a hole in [0, height) x [0, width), the hole's top-left corner lies at y in {0, h0 + crop} and x in
    // {0, w0 + crop}: testing those candidate points is exact
    int ys[kMaxWindows + 1], xs[kMaxWindows + 1], ny = 1, nx = 1;
    ys[0] = xs[0] = 0;
    for (int g = 0; g < G; ++g) {
        if (ws.h0[g] + crop > 0 && ws.h0[g] + crop < height) ys[ny++] = ws.h0[g] + crop;
        if (ws.w0[g] + crop > 0 && ws.w0[g] + crop < width) xs[nx++] = ws.w0[g] + crop;
    }
    for (int a = 0; a < ny; ++a)
        for (int b = 0; b < nx; ++b) {
            bool covered = false;
            for (int g = 0; g < G && !covered; ++g)
                covered = ys[a] >= ws.h0[g] && ys[a] < ws.h0[g] + crop && xs[b] >= ws.w0[g] && xs[b] < ws.w0[g] + crop;
            AVL_REQUIRE(covered, "avl_lseg_merge_windows: pixel (%d, %d) is covered by no window", ys[a], xs[b]);
        }
    const dim3 grid((unsigned)((width + 63) / 64), (unsigned)height, (unsigned)((D + 63) / 64));
    if (is_f16)
        hipLaunchKernelGGL(lseg_merge_windows_kernel<_Float16>, grid, dim3(256), 0, as_stream(stream), static_cast<const _Float16*>(d_win), ws, D,
                           crop, height, width, d_out);
    else
        hipLaunchKernelGGL(lseg_merge_windows_kernel<float>, grid, dim3(256), 0, as_stream(stream), static_cast<const float*>(d_win), ws, D, crop,
                           height, width, d_out);
    AVL_HIP_CHECK(hipGetLastError());
    return AVL_OK;
}
