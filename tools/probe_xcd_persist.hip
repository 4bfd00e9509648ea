// Kill-or-build measurement for the XCD-affine persistent builder.
//
// K3's memory work per frame, isolated: G voxel groups, each gathers one 2 KB float32 feature row (streamed, no reuse) and
// read-modify-writes one 4 KB float64 accumulator row; consecutive frames touch mostly the SAME voxels (a camera moves slowly).
//   (a) launch per frame, wave per group in list order          -- today's kernel boundary: rows leave through memory every frame
//   (b) launch per frame, group g goes to a workgroup on XCD (slot mod 8) (blockIdx mod 8; checked against HW_REG_XCC_ID)
//   (c) ONE persistent launch, every workgroup OWNS the slots (slot mod gridDim) for all frames, no grid barrier at all: a row is
//       only ever touched by one CU, so it can stay in that XCD's L2 (4 MB each) between frames -- the best case of the design
// If (c) is not clearly faster than (a) per frame, affinity cannot pay for the redesign of K2 / K3's work distribution.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_xcd_persist.hip -o /tmp/probe_xcd && /tmp/probe_xcd [groups] [overlap%] [frames]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int D = 512;

__device__ __forceinline__ void rmw_row(const float* __restrict__ feat, double* __restrict__ acc, int lane, double alpha) {
    const float4* f = reinterpret_cast<const float4*>(feat);
    double2* a = reinterpret_cast<double2*>(acc);
    const float4 v0 = f[lane], v1 = f[64 + lane];
    double2 a0 = a[2 * lane], a1 = a[2 * lane + 1], a2 = a[128 + 2 * lane], a3 = a[128 + 2 * lane + 1];
    a0.x += alpha * v0.x; a0.y += alpha * v0.y; a1.x += alpha * v0.z; a1.y += alpha * v0.w;
    a2.x += alpha * v1.x; a2.y += alpha * v1.y; a3.x += alpha * v1.z; a3.y += alpha * v1.w;
    a[2 * lane] = a0; a[2 * lane + 1] = a1; a[128 + 2 * lane] = a2; a[128 + 2 * lane + 1] = a3;
}

// (a) / (b): one frame per launch; list[g] = {slot, pixel}; (b) passes lists bucketed by slot mod 8 with per-bucket offsets
__global__ __launch_bounds__(256) void frame_kernel(int G, const int2* __restrict__ list, const float* __restrict__ feat, double* __restrict__ acc) {
    const int lane = threadIdx.x & 63;
    const int g = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (g >= G) return;
    const int2 e = list[g];
    rmw_row(feat + (size_t)e.y * D, acc + (size_t)e.x * D, lane, 0.5);
}

__global__ __launch_bounds__(256) void frame_xcd_kernel(const int* __restrict__ boff, const int2* __restrict__ list, const float* __restrict__ feat,
                                                        double* __restrict__ acc, int* __restrict__ xcc_miss) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = blockIdx.x & 7, wg = blockIdx.x >> 3, nwg = gridDim.x >> 3;
    if (threadIdx.x == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        if ((int)(id & 0xF) != x) atomicAdd(xcc_miss, 1);
    }
    for (int g = boff[x] + wg * 4 + wave; g < boff[x + 1]; g += nwg * 4) {
        const int2 e = list[g];
        rmw_row(feat + (size_t)e.y * D, acc + (size_t)e.x * D, lane, 0.5);
    }
}

// (c): persistent, workgroup w owns slots with slot % gridDim == w; CSR over (frame, workgroup)
__global__ __launch_bounds__(256) void persistent_kernel(int F, const int* __restrict__ off, const int2* __restrict__ list, const float* __restrict__ feat,
                                                         double* __restrict__ acc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int W = gridDim.x;
    for (int f = 0; f < F; ++f) {
        const int o0 = off[f * W + blockIdx.x], o1 = off[f * W + blockIdx.x + 1];
        for (int g = o0 + wave; g < o1; g += 4) {
            const int2 e = list[g];
            rmw_row(feat + (size_t)e.y * D, acc + (size_t)e.x * D, lane, 0.5);
        }
        __syncthreads();      // a slot is touched once per frame; frames of one workgroup stay in order
    }
}

int main(int argc, char** argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 2140, overlap = argc > 2 ? atoi(argv[2]) : 90, F = argc > 3 ? atoi(argv[3]) : 400;
    const int NSLOT = 300000, NPIX = 347 * 520;
    std::mt19937 rng(1);
    // frame f touches G distinct slots: `overlap` % of the previous frame's, the rest fresh ones
    std::vector<std::vector<int2>> frames(F);
    std::vector<int> cur;
    int fresh = 0;
    for (int f = 0; f < F; ++f) {
        std::vector<int> nxt;
        if (f) {
            std::shuffle(cur.begin(), cur.end(), rng);
            nxt.assign(cur.begin(), cur.begin() + (size_t)G * overlap / 100);
        }
        while ((int)nxt.size() < G) nxt.push_back(fresh++ % NSLOT);
        cur = nxt;
        std::shuffle(nxt.begin(), nxt.end(), rng);
        for (int s : nxt) frames[f].push_back(int2{s, (int)(rng() % NPIX)});
    }
    float* feat; double* acc;
    CK(hipMalloc(&feat, (size_t)NPIX * D * 4));
    CK(hipMalloc(&acc, (size_t)NSLOT * D * 8));
    CK(hipMemset(feat, 0, (size_t)NPIX * D * 4));
    CK(hipMemset(acc, 0, (size_t)NSLOT * D * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    // (a)
    std::vector<int2> flat;
    for (auto& fr : frames) flat.insert(flat.end(), fr.begin(), fr.end());
    int2* dlist; CK(hipMalloc(&dlist, flat.size() * sizeof(int2)));
    CK(hipMemcpy(dlist, flat.data(), flat.size() * sizeof(int2), hipMemcpyHostToDevice));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        for (int f = 0; f < F; ++f) frame_kernel<<<(G + 3) / 4, 256>>>(G, dlist + (size_t)f * G, feat, acc);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    }
    printf("(a) launch per frame, list order:        %.2f us per frame (%d groups, %d %% of a frame's voxels revisited, %.1f MB per frame)\n", 1e3 * ms / F, G,
           overlap, G * 10240.0 / 1e6);
    // (b)
    std::vector<int2> bl; std::vector<int> boff;
    for (auto& fr : frames) {
        for (int x = 0; x < 8; ++x) { boff.push_back((int)bl.size()); for (auto e : fr) if ((e.x & 7) == x) bl.push_back(e); }
        boff.push_back((int)bl.size());
    }
    int2* dbl; int* dboff; int* miss;
    CK(hipMalloc(&dbl, bl.size() * sizeof(int2))); CK(hipMalloc(&dboff, boff.size() * 4)); CK(hipMalloc(&miss, 4)); CK(hipMemset(miss, 0, 4));
    CK(hipMemcpy(dbl, bl.data(), bl.size() * sizeof(int2), hipMemcpyHostToDevice));
    CK(hipMemcpy(dboff, boff.data(), boff.size() * 4, hipMemcpyHostToDevice));
    for (int nb : {536, 1024}) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemset(miss, 0, 4));
            CK(hipEventRecord(e0));
            for (int f = 0; f < F; ++f) frame_xcd_kernel<<<nb, 256>>>(dboff + f * 9, dbl, feat, acc, miss);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        }
        int hm = 0; CK(hipMemcpy(&hm, miss, 4, hipMemcpyDeviceToHost));
        printf("(b) launch per frame, XCD = slot mod 8:  %.2f us per frame (%d workgroups; blockIdx mod 8 != XCC_ID in %d of %d workgroups)\n", 1e3 * ms / F, nb, hm,
               nb * F);
    }
    // (c)
    for (int W : {256, 512, 1024}) {
        std::vector<int2> pl; std::vector<int> off;
        for (auto& fr : frames) {
            std::vector<std::vector<int2>> b(W);
            for (auto e : fr) b[e.x % W].push_back(e);
            for (int w = 0; w < W; ++w) { off.push_back((int)pl.size()); pl.insert(pl.end(), b[w].begin(), b[w].end()); }
        }
        off.push_back((int)pl.size());
        // CSR rows are (frame, workgroup); off[f * W + w + 1] of the last workgroup of a frame = first of the next frame
        int2* dpl; int* doff;
        CK(hipMalloc(&dpl, pl.size() * sizeof(int2))); CK(hipMalloc(&doff, off.size() * 4));
        CK(hipMemcpy(dpl, pl.data(), pl.size() * sizeof(int2), hipMemcpyHostToDevice));
        CK(hipMemcpy(doff, off.data(), off.size() * 4, hipMemcpyHostToDevice));
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            persistent_kernel<<<W, 256>>>(F, doff, dpl, feat, acc);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        }
        printf("(c) ONE persistent launch, slots owned:  %.2f us per frame (%d workgroups, no grid barrier)\n", 1e3 * ms / F, W);
        CK(hipFree(dpl)); CK(hipFree(doff));
    }
    return 0;
}
