// Energy per matrix instruction on gfx950, measured at the package sensor: which MFMA flavour should carry the cross terms?
//   f16  v_mfma_f32_32x32x16_f16   (16 K MACs)      i8  v_mfma_i32_32x32x32_i8  (32 K MACs)      f16s  v_mfma_f32_4x4x4_16b_f16
//   f16m v_mfma_f32_16x16x32_f16   (8 K MACs)       fp8 v_mfma_f32_32x32x16_fp8_fp8 (16 K MACs)
// Every wave of a full-chip grid issues the instruction back to back (two independent accumulator chains) with GAP idle slots in
// between; operands are random (operand toggling is what draws the power).  For each (kind, gap): instructions/s and mean package
// power over ~1.5 s -> a line fit P = P0 + rate * E gives the energy per instruction.
// build + run (GPU box):  hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma_power tools/mfma_power.hip && /tmp/mfma_power
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dirent.h>
#include <string>
#include <thread>
#include <vector>
using half8 = __attribute__((ext_vector_type(8))) _Float16;
using half4 = __attribute__((ext_vector_type(4))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using i32x16 = __attribute__((ext_vector_type(16))) int;
using i32x4 = __attribute__((ext_vector_type(4))) int;

template <int GAP>
__device__ __forceinline__ void gap() {
    if constexpr (GAP >= 1) __builtin_amdgcn_s_sleep(GAP);
}

template <int KIND, int GAP>
__global__ __launch_bounds__(512) void burn(const int4* __restrict__ src, int iters, float* __restrict__ out) {
    const int4 a4 = src[threadIdx.x], b4 = src[512 + threadIdx.x];
    float sink = 0.f;
    if constexpr (KIND == 0) {
        const half8 a = __builtin_bit_cast(half8, a4), b = __builtin_bit_cast(half8, b4);
        f32x16 c0 = {}, c1 = {};
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
            gap<GAP>();
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c1, 0, 0, 0);
            gap<GAP>();
            asm volatile("" : "+v"(c0), "+v"(c1));
        }
        sink = c0[0] + c1[3];
    } else if constexpr (KIND == 5 || KIND == 6 || KIND == 7) {
        // operand-toggling dependence: A (5) / A and B (6) with the fp16 mantissas cut to 3 bits, (7) A all zero
        int4 am = a4, bm = b4;
        const int mask = (int)0xFF80FF80u;
        if (KIND == 7) am = int4{0, 0, 0, 0};
        else { am.x &= mask; am.y &= mask; am.z &= mask; am.w &= mask; }
        if (KIND == 6) { bm.x &= mask; bm.y &= mask; bm.z &= mask; bm.w &= mask; }
        const half8 a = __builtin_bit_cast(half8, am), b = __builtin_bit_cast(half8, bm);
        f32x16 c0 = {}, c1 = {};
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
            gap<GAP>();
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
            gap<GAP>();
            asm volatile("" : "+v"(c0), "+v"(c1));
        }
        sink = c0[0] + c1[3];
    } else if constexpr (KIND == 3) {
        using f32x4v = __attribute__((ext_vector_type(4))) float;
        const half8 a = __builtin_bit_cast(half8, a4), b = __builtin_bit_cast(half8, b4);
        f32x4v c0 = {}, c1 = {};
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
            gap<GAP>();
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, c1, 0, 0, 0);
            gap<GAP>();
            asm volatile("" : "+v"(c0), "+v"(c1));
        }
        sink = c0[0] + c1[3];
    } else if constexpr (KIND == 4) {
        const long a = ((long)a4.x << 32) | (unsigned)a4.y, b = ((long)b4.x << 32) | (unsigned)b4.y;
        f32x16 c0 = {}, c1 = {};
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a, b, c0, 0, 0, 0);
            gap<GAP>();
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(b, a, c1, 0, 0, 0);
            gap<GAP>();
            asm volatile("" : "+v"(c0), "+v"(c1));
        }
        sink = c0[0] + c1[3];
    } else if constexpr (KIND == 1) {
        i32x16 c0 = {}, c1 = {};
        const i32x4 a = __builtin_bit_cast(i32x4, a4), b = __builtin_bit_cast(i32x4, b4);
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
            gap<GAP>();
            c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(b, a, c1, 0, 0, 0);
            gap<GAP>();
            asm volatile("" : "+v"(c0), "+v"(c1));
        }
        sink = (float)(c0[0] + c1[3]);
    } else {
        const half8 a = __builtin_bit_cast(half8, a4), b = __builtin_bit_cast(half8, b4);
        const half4 a0 = __builtin_shufflevector(a, a, 0, 1, 2, 3), b0 = __builtin_shufflevector(b, b, 0, 1, 2, 3);
        f32x4 c0 = {}, c1 = {};
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_4x4x4f16(a0, b0, c0, 0, 0, 0);
            gap<GAP>();
            c1 = __builtin_amdgcn_mfma_f32_4x4x4f16(b0, a0, c1, 0, 0, 0);
            gap<GAP>();
            asm volatile("" : "+v"(c0), "+v"(c1));
        }
        sink = c0[0] + c1[3];
    }
    if (sink == 12345.678f) out[0] = sink;
}

static std::string find_power_file() {
    char bus[64] = {0};
    hipDeviceGetPCIBusId(bus, sizeof(bus), 0);
    for (char* p = bus; *p; ++p) *p = (char)tolower(*p);
    for (int card = 0; card < 64; ++card) {
        std::string base = "/sys/class/drm/card" + std::to_string(card) + "/device";
        char real[4096];
        if (!realpath(base.c_str(), real)) continue;
        if (!strstr(real, bus + 5) && !strstr(real, bus)) continue;     // "0000:0a:00.0" vs "0a:00.0"
        std::string hw = base + "/hwmon";
        DIR* d = opendir(hw.c_str());
        if (!d) continue;
        while (dirent* e = readdir(d))
            if (!strncmp(e->d_name, "hwmon", 5)) {
                closedir(d);
                return hw + "/" + e->d_name + "/power1_input";
            }
        closedir(d);
    }
    return "";
}

static double read_uw(const std::string& f) {
    FILE* fp = fopen(f.c_str(), "r");
    if (!fp) return 0;
    double v = 0;
    if (fscanf(fp, "%lf", &v) != 1) v = 0;
    fclose(fp);
    return v;
}

template <int KIND, int GAP>
static void run(const char* name, const int4* d_src, float* d_out, const std::string& pf) {
    const int blocks = 256, iters = 4000;
    burn<KIND, GAP><<<blocks, 512>>>(d_src, 100, d_out);
    hipDeviceSynchronize();
    std::atomic<bool> stop{false};
    std::vector<double> samples;
    std::thread th([&] {
        while (!stop) {
            std::this_thread::sleep_for(std::chrono::milliseconds(20));
            samples.push_back(read_uw(pf) * 1e-6);
        }
    });
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    int launches = 0;
    const auto t0 = std::chrono::steady_clock::now();
    hipEventRecord(e0);
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 1.6) {
        for (int k = 0; k < 20; ++k) burn<KIND, GAP><<<blocks, 512>>>(d_src, iters, d_out);
        launches += 20;
        hipDeviceSynchronize();
    }
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    stop = true;
    th.join();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    double w = 0;
    int n = 0;
    for (size_t i = samples.size() / 4; i < samples.size(); ++i) { w += samples[i]; ++n; }     // skip the ramp
    w = n ? w / n : 0;
    const double instr = (double)launches * blocks * 8 * 2.0 * iters;      // waves x 2 per iteration
    printf("%-6s gap %d  %8.3f G instr/s  %7.1f W  (%d samples, %.2f s)\n", name, GAP, instr / (ms * 1e-3) / 1e9, w, n, ms * 1e-3);
}

int main() {
    std::vector<int> h(1024 * 4);
    srand(1);
    // random fp16 values in [-2, 2) / random bytes: the same bit patterns serve both kinds
    for (auto& v : h) {
        unsigned lo = (unsigned)(rand() & 0x3ff) | ((unsigned)(14 + rand() % 2) << 10) | ((unsigned)(rand() & 1) << 15);
        unsigned hi = (unsigned)(rand() & 0x3ff) | ((unsigned)(14 + rand() % 2) << 10) | ((unsigned)(rand() & 1) << 15);
        v = (int)(lo | (hi << 16));
    }
    int4* d_src;
    float* d_out;
    hipMalloc(&d_src, h.size() * 4);
    hipMalloc(&d_out, 64);
    hipMemcpy(d_src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const std::string pf = find_power_file();
    printf("power sensor: %s  (idle %.1f W)\n", pf.c_str(), read_uw(pf) * 1e-6);
    run<0, 0>("f16", d_src, d_out, pf); run<0, 1>("f16", d_src, d_out, pf); run<0, 3>("f16", d_src, d_out, pf);
    run<1, 0>("i8", d_src, d_out, pf);  run<1, 1>("i8", d_src, d_out, pf);  run<1, 3>("i8", d_src, d_out, pf);
    run<2, 0>("f16s", d_src, d_out, pf); run<2, 1>("f16s", d_src, d_out, pf);
    run<3, 0>("f16m", d_src, d_out, pf); run<3, 1>("f16m", d_src, d_out, pf); run<3, 3>("f16m", d_src, d_out, pf);
    run<4, 1>("fp8", d_src, d_out, pf); run<4, 3>("fp8", d_src, d_out, pf);
    run<0, 3>("f16", d_src, d_out, pf);
    run<5, 3>("f16a3", d_src, d_out, pf); run<6, 3>("f16ab3", d_src, d_out, pf); run<7, 3>("f16a0", d_src, d_out, pf);
    run<5, 1>("f16a3", d_src, d_out, pf); run<6, 1>("f16ab3", d_src, d_out, pf); run<7, 1>("f16a0", d_src, d_out, pf);
    return 0;
}
