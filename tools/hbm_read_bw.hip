// Practical HBM read ceiling on the box: coalesced 16-byte streaming read of a 4 GiB buffer (read-only, sum to a sink).
// hipcc --offload-arch=gfx950 -O3 tools/hbm_read_bw.hip -o tools/hbm_read_bw && tools/hbm_read_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

template <int UNROLL>
__global__ __launch_bounds__(256) void read_kernel(const float4* __restrict__ p, size_t n, float* sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float acc = 0.f;
    for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
        float4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = p[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    for (; i < n; i += stride) { float4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) *sink = acc;
}

// row-strided pattern of the similarity kernel: each lane walks one 128-byte line with 8 x 16 B loads
__global__ __launch_bounds__(512) void rowline_kernel(const float* __restrict__ feat, long N, int D, float* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, kg = lane >> 5;
    const long ntiles = (N + 255) / 256;
    float acc = 0.f;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        long row = tile * 256 + wave * 32 + j;
        if (row >= N) row = N - 1;
        const float4* g = reinterpret_cast<const float4*>(feat + row * D + 32 * kg);
        for (int s = 0; s < D / 64; ++s) {
            float4 v[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) v[t] = g[s * 16 + t];
#pragma unroll
            for (int t = 0; t < 8; ++t) acc += v[t].x + v[t].y + v[t].z + v[t].w;
        }
    }
    if (acc == 123.456f) *sink = acc;
}

int main() {
    const size_t bytes = 4096000000ull;  // 2M x 512 x 4
    float4* buf; float* sink;
    hipMalloc(&buf, bytes); hipMalloc(&sink, 4);
    hipMemset(buf, 1, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](auto launch, const char* name) {
        std::vector<float> ts;
        for (int it = 0; it < 12; ++it) {
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (it >= 2) ts.push_back(ms);
        }
        std::sort(ts.begin(), ts.end());
        printf("%-34s median %.3f ms  %.0f GB/s   best %.3f ms %.0f GB/s\n", name, ts[ts.size() / 2], bytes / ts[ts.size() / 2] / 1e6, ts[0], bytes / ts[0] / 1e6);
    };
    const size_t n = bytes / 16;
    for (int blocks : {2048, 4096, 8192, 16384})
        timeit([&] { hipLaunchKernelGGL(read_kernel<4>, dim3(blocks), dim3(256), 0, 0, buf, n, sink); }, ("coalesced x4, blocks=" + std::to_string(blocks)).c_str());
    timeit([&] { hipLaunchKernelGGL(read_kernel<8>, dim3(4096), dim3(256), 0, 0, buf, n, sink); }, "coalesced x8, blocks=4096");
    for (int blocks : {256, 512})
        timeit([&] { hipLaunchKernelGGL(rowline_kernel, dim3(blocks), dim3(512), 0, 0, (const float*)buf, 2000000L, 512, sink); }, ("row-line pattern, blocks=" + std::to_string(blocks)).c_str());
    return 0;
}
