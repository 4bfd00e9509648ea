/* Plain-C caller of libavlmaps_hip.so: no Python, no torch.  Builds a random map, scores it against a few queries through the
 * C ABI (include/avlmaps_hip.h), checks a few rows on the host and times the call with the library's own events.
 *
 *   gcc -O2 -Iinclude examples/c_caller.c -Lavlmaps_amd/lib -lavlmaps_hip -Wl,-rpath,$PWD/avlmaps_amd/lib -lm -o /tmp/c_caller
 *   /tmp/c_caller [n_voxels] [n_queries]
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "avlmaps_hip.h"

#define CHECK(call)                                                          \
    do {                                                                     \
        int rc_ = (call);                                                    \
        if (rc_ != 0) {                                                      \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, avl_last_error()); \
            return 1;                                                        \
        }                                                                    \
    } while (0)

static float frand(unsigned* s) {
    *s = *s * 1664525u + 1013904223u;
    return ((*s >> 8) & 0xFFFF) / 32768.0f - 1.0f;
}

int main(int argc, char** argv) {
    const int64_t N = argc > 1 ? atoll(argv[1]) : 100000;
    const int Q = argc > 2 ? atoi(argv[2]) : 2, D = 512;
    int ndev = 0;
    CHECK(avl_device_count(&ndev));
    if (ndev == 0) {
        fprintf(stderr, "no GPU\n");
        return 2;
    }
    CHECK(avl_set_device(0));
    float* h_feat = (float*)malloc((size_t)N * D * sizeof(float));
    float* h_q = (float*)malloc((size_t)Q * D * sizeof(float));
    int32_t* h_am = (int32_t*)malloc((size_t)N * sizeof(int32_t));
    unsigned seed = 7;
    for (int64_t i = 0; i < N * D; ++i) h_feat[i] = frand(&seed);
    for (int i = 0; i < Q * D; ++i) h_q[i] = frand(&seed) / 22.0f;

    void *d_feat, *d_q, *d_am;
    CHECK(avl_malloc(&d_feat, (size_t)N * D * sizeof(float)));
    CHECK(avl_malloc(&d_q, (size_t)Q * D * sizeof(float)));
    CHECK(avl_malloc(&d_am, (size_t)N * sizeof(int32_t)));
    CHECK(avl_memcpy_h2d(d_feat, h_feat, (size_t)N * D * sizeof(float), NULL));
    CHECK(avl_memcpy_h2d(d_q, h_q, (size_t)Q * D * sizeof(float), NULL));

    void *e0, *e1;
    CHECK(avl_event_create(&e0));
    CHECK(avl_event_create(&e1));
    const int warm = 50, iters = 200;
    for (int i = 0; i < warm; ++i)
        CHECK(avl_sim_scores((const float*)d_feat, N, D, D, (const float*)d_q, Q, D, NULL, (int32_t*)d_am, NULL, AVL_SIM_AUTO, NULL));
    CHECK(avl_event_record(e0, NULL));
    for (int i = 0; i < iters; ++i)
        CHECK(avl_sim_scores((const float*)d_feat, N, D, D, (const float*)d_q, Q, D, NULL, (int32_t*)d_am, NULL, AVL_SIM_AUTO, NULL));
    CHECK(avl_event_record(e1, NULL));
    CHECK(avl_event_sync(e1));
    float ms = 0.f;
    CHECK(avl_event_elapsed_ms(e0, e1, &ms));
    CHECK(avl_memcpy_d2h(h_am, d_am, (size_t)N * sizeof(int32_t), NULL));
    CHECK(avl_device_sync());

    /* host check of the first rows: the fused argmax must be a maximum of the float64 scores (within fp32 round-off) */
    int bad = 0;
    for (int64_t r = 0; r < (N < 64 ? N : 64); ++r) {
        double best = -1e300, got = 0;
        for (int c = 0; c < Q; ++c) {
            double s = 0;
            for (int k = 0; k < D; ++k) s += (double)h_feat[r * D + k] * h_q[c * D + k];
            if (s > best) best = s;
            if (c == h_am[r]) got = s;
        }
        if (got < best - 1e-4) ++bad;
    }
    printf("N=%lld Q=%d: %.2f us per call (%.0f GB/s of map read), %d/64 rows off\n", (long long)N, Q, ms / iters * 1e3,
           (double)N * D * 4 / (ms / iters * 1e-3) / 1e9, bad);
    avl_event_destroy(e0);
    avl_event_destroy(e1);
    avl_free(d_feat);
    avl_free(d_q);
    avl_free(d_am);
    free(h_feat);
    free(h_q);
    free(h_am);
    return bad ? 3 : 0;
}
