// Layout probe of v_mfma_f32_4x4x4_16b_f16 on gfx950 (used by the "extra query rows" path of the similarity kernel).
// hypothesis H1: D[lane][r] = sum_k A[4*(lane/4) + r][k] * B[lane][k]   (block = lane/4, A row = r, B column = lane%4)
// hypothesis H2: D[lane][r] = sum_k A[lane][k] * B[4*(lane/4) + r][k]
// build + run:  hipcc --offload-arch=gfx950 -O2 -o /tmp/probe_mfma4 tools/probe_mfma4.hip && /tmp/probe_mfma4
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
using half4 = __attribute__((ext_vector_type(4))) _Float16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
__global__ void k(const half4* a, const half4* b, f32x4* c) {
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_4x4x4f16(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
    c[threadIdx.x] = acc;
}
int main() {
    float A[64][4], B[64][4];
    half4 ha[64], hb[64];
    for (int l = 0; l < 64; ++l)
        for (int k = 0; k < 4; ++k) {
            A[l][k] = (float)((l * 7 + k * 3) % 11 - 5);
            B[l][k] = (float)((l * 5 + k * 13) % 9 - 4);
            ha[l][k] = (_Float16)A[l][k];
            hb[l][k] = (_Float16)B[l][k];
        }
    half4 *da, *db;
    f32x4* dc;
    hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dc, 64 * sizeof(f32x4));
    hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice);
    hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
    k<<<1, 64>>>(da, db, dc);
    f32x4 hc[64];
    hipMemcpy(hc, dc, sizeof(hc), hipMemcpyDeviceToHost);
    int ok1 = 0, ok2 = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            float h1 = 0, h2 = 0;
            for (int kk = 0; kk < 4; ++kk) {
                h1 += A[4 * (l / 4) + r][kk] * B[l][kk];
                h2 += A[l][kk] * B[4 * (l / 4) + r][kk];
            }
            ok1 += hc[l][r] == h1;
            ok2 += hc[l][r] == h2;
        }
    printf("mfma_4x4x4_16b_f16 layout: H1 %d/256  H2 %d/256\n", ok1, ok2);
    return 0;
}
