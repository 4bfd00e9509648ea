// How fast does the device start waves?  Kernels whose waves leave at once (one scalar test of an argument), for a range of wave
// counts, workgroup sizes and register allocations; time per launch from hipEvents over back-to-back launches and the slope
// between wave counts = the launch rate.  Decides whether the builder's K3 (7 776 waves per frame, 72 % of them idle) is bound by
// starting waves rather than by residency or by the number of workgroups.
// hipcc --offload-arch=gfx950 -O3 tools/wave_launch_rate.hip -o /tmp/wave_launch_rate && /tmp/wave_launch_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x)                                                                     \
    do {                                                                             \
        hipError_t e = (x);                                                          \
        if (e != hipSuccess) {                                                       \
            printf("%s failed: %s\n", #x, hipGetErrorString(e));                     \
            return 1;                                                                \
        }                                                                            \
    } while (0)

// REGS: a live range of that many VGPRs on a path no wave takes (the allocation is what the dispatcher reserves per wave)
template <int THREADS, int REGS>
__global__ __launch_bounds__(THREADS) void idle_kernel(const float* __restrict__ p, float* __restrict__ out) {
    if (!p) return;                      // every wave of the measurement leaves here
    float v[REGS];
#pragma unroll
    for (int i = 0; i < REGS; ++i) v[i] = p[threadIdx.x + i * THREADS];
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < REGS; ++i) acc = acc * v[i] + v[(i + 1) % REGS];
    out[blockIdx.x * THREADS + threadIdx.x] = acc;
}

// the same with ONE load per wave before it leaves (the builder's idle waves read their sample's owner flag)
template <int THREADS>
__global__ __launch_bounds__(THREADS) void flag_kernel(const int* __restrict__ flag, float* __restrict__ out) {
    const int w = (blockIdx.x * THREADS + threadIdx.x) >> 6;
    if (__builtin_amdgcn_readfirstlane(flag[w]) == 0) return;
    out[blockIdx.x * THREADS + threadIdx.x] = 1.f;
}

// the builder's K3 in miniature: `pct` % of the waves (pseudo-random by wave index) stay for `ticks` x 10 ns, the rest leave at
// once; REGS sets the allocation, i.e. how many of the waves are resident together
template <int THREADS, int REGS>
__global__ __launch_bounds__(THREADS) void mixed_kernel(const float* __restrict__ p, float* __restrict__ out, int pct, int ticks) {
    const unsigned w = (blockIdx.x * THREADS + threadIdx.x) >> 6;
    if (((w * 2654435761u) >> 16) % 100u < (unsigned)pct) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(8);
    }
    if (!p) return;
    float v[REGS];
#pragma unroll
    for (int i = 0; i < REGS; ++i) v[i] = p[threadIdx.x + i * THREADS];
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < REGS; ++i) acc = acc * v[i] + v[(i + 1) % REGS];
    out[blockIdx.x * THREADS + threadIdx.x] = acc;
}

template <typename F>
static float time_us(F launch, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 20; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a);
    hipEventDestroy(b);
    return ms * 1000.f / reps;
}

int main() {
    float* out = nullptr;
    int* flag = nullptr;
    CHECK(hipMalloc(&out, 1 << 24));
    CHECK(hipMalloc(&flag, 1 << 20));
    CHECK(hipMemset(flag, 0, 1 << 20));
    const int waves[] = {64, 512, 2048, 4096, 8192, 16384, 32768};
    const int reps = 2000;
    printf("us per launch (back-to-back launches on one stream: includes the ~1.5-2 us kernel boundary)\n");
    printf("%8s | %12s %12s %12s | %12s %12s | %12s\n", "waves", "256thr 16reg", "256thr 80reg", "256thr 128r", "512thr 80reg", "64thr 80reg", "256thr flag");
    std::vector<std::vector<float>> t;
    for (int w : waves) {
        std::vector<float> r;
        r.push_back(time_us([&] { hipLaunchKernelGGL((idle_kernel<256, 8>), dim3(w / 4), dim3(256), 0, 0, (const float*)nullptr, out); }, reps));
        r.push_back(time_us([&] { hipLaunchKernelGGL((idle_kernel<256, 72>), dim3(w / 4), dim3(256), 0, 0, (const float*)nullptr, out); }, reps));
        r.push_back(time_us([&] { hipLaunchKernelGGL((idle_kernel<256, 120>), dim3(w / 4), dim3(256), 0, 0, (const float*)nullptr, out); }, reps));
        r.push_back(time_us([&] { hipLaunchKernelGGL((idle_kernel<512, 72>), dim3(w / 8), dim3(512), 0, 0, (const float*)nullptr, out); }, reps));
        r.push_back(time_us([&] { hipLaunchKernelGGL((idle_kernel<64, 72>), dim3(w), dim3(64), 0, 0, (const float*)nullptr, out); }, reps));
        r.push_back(time_us([&] { hipLaunchKernelGGL((flag_kernel<256>), dim3(w / 4), dim3(256), 0, 0, (const int*)flag, out); }, reps));
        printf("%8d | %12.2f %12.2f %12.2f | %12.2f %12.2f | %12.2f\n", w, r[0], r[1], r[2], r[3], r[4], r[5]);
        t.push_back(r);
    }
    const int n = (int)t.size();
    printf("slope between %d and %d waves, ns per wave:", waves[2], waves[n - 1]);
    for (size_t k = 0; k < t[0].size(); ++k) printf(" %.3f", 1000.f * (t[n - 1][k] - t[2][k]) / (float)(waves[n - 1] - waves[2]));
    printf("\n");
    printf("\n7776 waves, 28 %% of them busy for 4 us (the rest leave at once): us per launch\n");
    printf("%28s %10s %10s %10s\n", "", "40 regs", "80 regs", "128 regs");
    for (int pct : {0, 28, 100}) {
        const float a = time_us([&] { hipLaunchKernelGGL((mixed_kernel<256, 32>), dim3(1944), dim3(256), 0, 0, (const float*)nullptr, out, pct, 400); }, reps);
        const float b = time_us([&] { hipLaunchKernelGGL((mixed_kernel<256, 72>), dim3(1944), dim3(256), 0, 0, (const float*)nullptr, out, pct, 400); }, reps);
        const float c = time_us([&] { hipLaunchKernelGGL((mixed_kernel<256, 120>), dim3(1944), dim3(256), 0, 0, (const float*)nullptr, out, pct, 400); }, reps);
        printf("256 threads, %3d %% busy      %10.2f %10.2f %10.2f\n", pct, a, b, c);
    }
    for (int pct : {28}) {
        const float a = time_us([&] { hipLaunchKernelGGL((mixed_kernel<512, 32>), dim3(972), dim3(512), 0, 0, (const float*)nullptr, out, pct, 400); }, reps);
        const float b = time_us([&] { hipLaunchKernelGGL((mixed_kernel<512, 72>), dim3(972), dim3(512), 0, 0, (const float*)nullptr, out, pct, 400); }, reps);
        const float c = time_us([&] { hipLaunchKernelGGL((mixed_kernel<64, 72>), dim3(7776), dim3(64), 0, 0, (const float*)nullptr, out, pct, 400); }, reps);
        printf("28 %% busy: 512 thr 40 / 80 regs, 64 thr 80 regs %10.2f %10.2f %10.2f\n", a, b, c);
    }
    hipFree(out);
    hipFree(flag);
    return 0;
}
