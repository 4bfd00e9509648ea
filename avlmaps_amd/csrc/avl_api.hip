// libavlmaps_hip.so -- library / device plumbing entry points (include/avlmaps_hip.h, first block).
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <immintrin.h>

#include "avl_common.h"

namespace avl {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

void* scratch(size_t bytes) {
    struct Slot { int dev; void* p; size_t bytes; };
    static thread_local Slot slots[16] = {};
    static thread_local int used = 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    Slot* s = nullptr;
    for (int i = 0; i < used; ++i)
        if (slots[i].dev == dev) s = &slots[i];
    if (!s) {
        if (used == 16) { set_error("scratch: more than 16 devices used from one thread"); return nullptr; }
        s = &slots[used++];
        *s = Slot{dev, nullptr, 0};
    }
    if (s->bytes < bytes) {
        if (s->p) { (void)hipDeviceSynchronize(); (void)hipFree(s->p); s->p = nullptr; s->bytes = 0; }
        const size_t want = bytes < (1u << 20) ? (1u << 20) : bytes;
        hipError_t e = hipMalloc(&s->p, want);
        if (e != hipSuccess) { set_error("scratch: hipMalloc(%zu) failed: %s", want, hipGetErrorString(e)); s->p = nullptr; return nullptr; }
        s->bytes = want;
    }
    return s->p;
}

// The finalisation / replay / merge paths take their temporaries from the device's default stream-ordered pool (hipMallocAsync).
// HIP's default release threshold is 0: every synchronisation hands the pool's memory back to the driver and the next call maps
// it again -- tens of ms per GB, which showed as 30-60 ms of "allocation" inside a 30 ms merge (profiles/r04_build_8ranks_*).
// Keep a BOUNDED amount between calls (AVLMAPS_MEMPOOL_KEEP_MB, default 1024: the temporaries of a checkpoint finalisation of a
// 2 M-voxel map); what a one-off merge grew beyond that goes back at the next synchronisation, and avl_release_scratch trims the
// pool on request (an unbounded threshold kept GBs outside torch's allocator next to a feature extractor).
static uint64_t mempool_keep_bytes() {
    const char* e = getenv("AVLMAPS_MEMPOOL_KEEP_MB");
    long long mb = e ? atoll(e) : 1024;
    if (mb < 0) mb = 0;
    return (uint64_t)mb << 20;
}

void keep_mempool_once() {
    static thread_local int done_for = -1;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || done_for == dev) return;
    hipMemPool_t pool = nullptr;
    if (hipDeviceGetDefaultMemPool(&pool, dev) == hipSuccess && pool) {
        uint64_t keep = mempool_keep_bytes();
        (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
    }
    (void)hipGetLastError();
    done_for = dev;
}

void trim_mempool(uint64_t keep) {
    int dev = 0;
    hipMemPool_t pool = nullptr;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetDefaultMemPool(&pool, dev) == hipSuccess && pool)
        (void)hipMemPoolTrimTo(pool, (size_t)keep);
    (void)hipGetLastError();
}

int num_cus() {
    static thread_local int cached_dev = -1;
    static thread_local int cached = 256;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return cached;
    if (dev != cached_dev) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, dev) == hipSuccess) cached = p.multiProcessorCount;
        cached_dev = dev;
    }
    return cached;
}

}  // namespace avl

namespace avl {

// read-only streaming probes: what this box's HBM delivers to a kernel that does nothing but read
__global__ __launch_bounds__(256) void probe_coalesced_kernel(const float4* __restrict__ p, size_t n, float* sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float acc = 0.f;
    for (; i + 3 * stride < n; i += 4 * stride) {
        const float4 a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
        acc += (a.x + a.y + a.z + a.w) + (b.x + b.y + b.z + b.w) + (c.x + c.y + c.z + c.w) + (d.x + d.y + d.z + d.w);
    }
    for (; i < n; i += stride) { const float4 a = p[i]; acc += a.x + a.y + a.z + a.w; }
    if (acc == 123.456f) *sink = acc;
}

// the access pattern of the similarity kernels: lane (row j, half kg) walks one 128-byte line with 8 x 16-byte loads
__global__ __launch_bounds__(512) void probe_rowline_kernel(const float* __restrict__ feat, long long N, int D, float* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, kg = lane >> 5;
    const long long ntiles = (N + 255) / 256;
    float acc = 0.f;
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        long long row = tile * 256 + wave * 32 + j;
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

}  // namespace avl

using namespace avl;

extern "C" {

const char* avl_last_error(void) { return g_err; }
int avl_version(void) { return 202; }   // 0.2.0: avl_sim_prepare_map gained the row-scale output (ABI change); 0.2.1: deferred fuse / flush; 0.2.2: avl_gather_rows, avl_host_alloc, avl_get_device, avl_mt19937_skip_shuffles / _shuffle_sample, avl_sim_prepare_map24 / avl_sim_scores_prepared24

int avl_device_count(int* h_count) {
    AVL_REQUIRE(h_count, "avl_device_count: null output");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *h_count = 0;
        set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
        return AVL_ERR_NO_DEVICE;
    }
    *h_count = n;
    return AVL_OK;
}

int avl_get_device(int* h_device) {
    AVL_REQUIRE(h_device, "avl_get_device: null output");
    AVL_HIP_CHECK(hipGetDevice(h_device));
    return AVL_OK;
}

int avl_set_device(int device) {
    AVL_HIP_CHECK(hipSetDevice(device));
    return AVL_OK;
}

int avl_device_name(int device, char* h_buf, size_t buf_len) {
    AVL_REQUIRE(h_buf && buf_len > 0, "avl_device_name: bad buffer");
    hipDeviceProp_t p;
    AVL_HIP_CHECK(hipGetDeviceProperties(&p, device));
    snprintf(h_buf, buf_len, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount);
    return AVL_OK;
}

int avl_device_sync(void) {
    AVL_HIP_CHECK(hipDeviceSynchronize());
    return AVL_OK;
}

int avl_stream_create(void** h_stream_out) {
    AVL_REQUIRE(h_stream_out, "avl_stream_create: null output");
    hipStream_t s;
    AVL_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *h_stream_out = s;
    return AVL_OK;
}

int avl_stream_destroy(void* stream) {
    AVL_HIP_CHECK(hipStreamDestroy(as_stream(stream)));
    return AVL_OK;
}

int avl_stream_sync(void* stream) {
    AVL_HIP_CHECK(hipStreamSynchronize(as_stream(stream)));
    return AVL_OK;
}

int avl_malloc(void** h_ptr_out, size_t bytes) {
    AVL_REQUIRE(h_ptr_out, "avl_malloc: null output");
    AVL_HIP_CHECK(hipMalloc(h_ptr_out, bytes ? bytes : 1));
    return AVL_OK;
}

int avl_free(void* d_ptr) {
    if (d_ptr) AVL_HIP_CHECK(hipFree(d_ptr));
    return AVL_OK;
}

// ---- host-only helper: fast-forward NumPy's legacy global RNG past pixel shuffles ------------------------------------------
// np.random.shuffle(arange(n)) (vlmap_builder.py:275-277) draws, for i = n-1 .. 1, one bounded integer j <= i with the legacy
// masked-rejection rule (numpy/random/_legacy: random_interval): mask = smallest 2^k - 1 >= i, 32-bit Mersenne-twister outputs
// are drawn until (x & mask) <= i.  The number of draws is data dependent, so skipping a shuffle means drawing them -- but only
// drawing: no 6 MB index array is permuted.  A rank of a sharded build uses this to reach the RNG state its first frame has in
// the single-process run (VLMapBuilder.skip_pixel_shuffles): ~1.7 ms instead of ~7 ms per skipped 720x1080 frame.
namespace {
struct Mt19937 {
    uint32_t* key;   // 624 words, NumPy's layout (np.random.get_state()[1])
    int pos;         // next word to hand out, 624 = regenerate first
    void regen() {
        constexpr uint32_t kUpper = 0x80000000u, kLower = 0x7fffffffu, kMat = 0x9908b0dfu;
        int i = 0;
        for (; i < 624 - 397; ++i) {
            const uint32_t y = (key[i] & kUpper) | (key[i + 1] & kLower);
            key[i] = key[i + 397] ^ (y >> 1) ^ ((y & 1u) ? kMat : 0u);
        }
        for (; i < 623; ++i) {
            const uint32_t y = (key[i] & kUpper) | (key[i + 1] & kLower);
            key[i] = key[i + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? kMat : 0u);
        }
        const uint32_t y = (key[623] & kUpper) | (key[0] & kLower);
        key[623] = key[396] ^ (y >> 1) ^ ((y & 1u) ? kMat : 0u);
        pos = 0;
    }
    inline uint32_t next() {
        if (pos >= 624) regen();
        uint32_t y = key[pos++];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        return y;
    }
};
}  // namespace

extern "C++" {
namespace {
// n_shuffles legacy shuffles of n_items elements on generator g.  The loop consumes one tempered output per iteration and is
// branch-free inside: a rejected draw (x & mask) > i leaves i where it is (and, with SWAP, swaps arr[i] with itself) -- the
// rejection branch of the textbook form mispredicts on a quarter of the draws and cost more than the generator.
template <int MODE>   // 0: draw only (fast-forward), 1: permute arr, 2: arr[i] = the accepted draw j_i of step i
void run_shuffles(Mt19937& g, int64_t n_items, int64_t n_shuffles, int32_t* arr) {
    uint32_t out[624];                       // tempered outputs of the current state block (filled in one vectorisable sweep)
    auto temper_block = [&]() {
        for (int k = 0; k < 624; ++k) {
            uint32_t y = g.key[k];
            y ^= (y >> 11);
            y ^= (y << 7) & 0x9d2c5680u;
            y ^= (y << 15) & 0xefc60000u;
            y ^= (y >> 18);
            out[k] = y;
        }
    };
    if (g.pos < 624) temper_block();
    for (int64_t s = 0; s < n_shuffles; ++s) {
        if (n_items < 2) continue;
        uint32_t i = (uint32_t)(n_items - 1), mask = i;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
        // The mask (smallest 2^m - 1 >= i) changes only ~log2(n) times per shuffle: the loop is cut into runs of CONSTANT mask,
        // i from its current value down to (mask >> 1) + 1.  Inside a run the only loop-carried dependency is
        // i -= (x <= i) -- two cycles per draw instead of the five of the form that re-derives the mask from i on every draw
        // (round 4: 2.0 -> ~1.2 ms per 720x1080 shuffle, 1.7 -> ~0.8 ms per skipped one; the pixel-faithful pipeline is bound by
        // this one serial loop).
        while (i >= 1) {
            const uint32_t lo = mask >> 1;            // the run ends when i == lo (then the mask halves); lo == 0 for i == 1
            while (i > lo) {
                if (g.pos >= 624) {
                    g.regen();
                    temper_block();
                }
                const int avail = 624 - g.pos;
                const uint32_t* o = out + g.pos;
                int k = 0;
                if (MODE == 1) {          // permute arr in place
                    for (; k < avail && i > lo; ++k) {
                        const uint32_t x = o[k] & mask;
                        const uint32_t acc = x <= i ? 1u : 0u;
                        const uint32_t jj = acc ? x : i;
                        const int32_t t = arr[i];
                        arr[i] = arr[jj];
                        arr[jj] = t;
                        i -= acc;
                    }
                } else if (MODE == 2) {   // record the accepted draw of every step: arr[i] = j_i (a rejected draw is overwritten)
                    for (; k < avail && i > lo; ++k) {
                        const uint32_t x = o[k] & mask;
                        arr[i] = (int32_t)x;
                        i -= x <= i ? 1u : 0u;
                    }
                } else {
                    for (; k < avail && i > lo; ++k) {
                        const uint32_t x = o[k] & mask;
                        i -= x <= i ? 1u : 0u;
                    }
                }
                g.pos += k;
            }
            mask = lo;
        }
    }
}

// AVX2 form of the draw loop (modes 0 and 2).  Inside a constant-mask run eight draws are decided at once: with i the running index
// before the block, a draw x <= i - 7 is accepted whatever happened to the seven before it, a draw x > i is rejected; only
// i - 7 < x <= i is ambiguous (probability 7 / (mask + 1) per draw: about once per 10^5 draws for 720x1080) and sends the block
// through the scalar loop.  The accepted draws are packed with a 256-entry permutation table so that the r-th accepted one lands at
// arr[i - r], exactly where the scalar loop writes it.  Same draws, same state: tests/test_host_mirror.py pins both forms.
alignas(32) static uint32_t g_pack_lut[256][8];
static void build_pack_lut() {
    for (int a = 0; a < 256; ++a) {
        int src[8], cnt = 0;
        for (int b = 0; b < 8; ++b)
            if (a & (1 << b)) src[cnt++] = b;
        for (int lane = 0; lane < 8; ++lane) {
            const int r = 7 - lane;                      // lane 7 holds the first accepted draw (address i), lane 6 the second, ...
            g_pack_lut[a][lane] = (uint32_t)(r < cnt ? src[r] : 0);
        }
    }
}

// the Mersenne-twister state update, eight words at a time (the recurrence reads key[i + 1] and key[i + 397 mod 624]: distances that
// leave eight consecutive words independent), and the tempering of the whole block
__attribute__((target("avx2"))) static inline void mt_step8(uint32_t* key, int i, int j) {     // key[i .. i+7] from key[i .. i+8], key[j .. j+7]
    const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(key + i));
    const __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(key + i + 1));
    const __m256i y = _mm256_or_si256(_mm256_and_si256(a, _mm256_set1_epi32((int)0x80000000u)), _mm256_and_si256(b, _mm256_set1_epi32(0x7fffffff)));
    const __m256i odd = _mm256_sub_epi32(_mm256_setzero_si256(), _mm256_and_si256(y, _mm256_set1_epi32(1)));       // all ones where y is odd
    const __m256i r = _mm256_xor_si256(_mm256_xor_si256(_mm256_loadu_si256(reinterpret_cast<const __m256i*>(key + j)), _mm256_srli_epi32(y, 1)),
                                       _mm256_and_si256(odd, _mm256_set1_epi32((int)0x9908b0dfu)));
    _mm256_storeu_si256(reinterpret_cast<__m256i*>(key + i), r);
}
static inline void mt_step1(uint32_t* key, int i, int i1, int j) {
    const uint32_t y = (key[i] & 0x80000000u) | (key[i1] & 0x7fffffffu);
    key[i] = key[j] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}
__attribute__((target("avx2"))) static void mt_regen_avx2(uint32_t* key) {
    int i = 0;
    for (; i + 8 <= 227; i += 8) mt_step8(key, i, i + 397);
    for (; i < 227; ++i) mt_step1(key, i, i + 1, i + 397);
    for (; i + 8 <= 623; i += 8) mt_step8(key, i, i - 227);
    for (; i < 623; ++i) mt_step1(key, i, i + 1, i - 227);
    mt_step1(key, 623, 0, 396);
}
__attribute__((target("avx2"))) static void mt_temper_avx2(const uint32_t* key, uint32_t* out) {
    const __m256i m1 = _mm256_set1_epi32((int)0x9d2c5680u), m2 = _mm256_set1_epi32((int)0xefc60000u);
    for (int k = 0; k < 624; k += 8) {
        __m256i y = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(key + k));
        y = _mm256_xor_si256(y, _mm256_srli_epi32(y, 11));
        y = _mm256_xor_si256(y, _mm256_and_si256(_mm256_slli_epi32(y, 7), m1));
        y = _mm256_xor_si256(y, _mm256_and_si256(_mm256_slli_epi32(y, 15), m2));
        y = _mm256_xor_si256(y, _mm256_srli_epi32(y, 18));
        _mm256_store_si256(reinterpret_cast<__m256i*>(out + k), y);
    }
}

template <int MODE>
__attribute__((target("avx2,popcnt"))) void run_shuffles_avx2(Mt19937& g, int64_t n_items, int64_t n_shuffles, int32_t* arr) {
    alignas(32) uint32_t out[624 + 8];
    if (g.pos < 624) mt_temper_avx2(g.key, out);
    for (int64_t s = 0; s < n_shuffles; ++s) {
        if (n_items < 2) continue;
        uint32_t i = (uint32_t)(n_items - 1), mask = i;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
        while (i >= 1) {
            const uint32_t lo = mask >> 1;
            const __m256i vmask = _mm256_set1_epi32((int)mask);
            while (i > lo) {
                if (g.pos >= 624) {
                    mt_regen_avx2(g.key);
                    g.pos = 0;
                    mt_temper_avx2(g.key, out);
                }
                const int avail = 624 - g.pos;
                const uint32_t* o = out + g.pos;
                int k = 0;
                for (;;) {
                    // 32 draws against ONE broadcast of i: the loop-carried chain (i -> broadcast -> compare -> movemask -> popcount
                    // -> i, ~15 cycles) is paid once per 32 draws instead of once per 8
                    while (k + 32 <= avail && i >= lo + 32) {
                        const __m256i vi = _mm256_set1_epi32((int)i), vlo = _mm256_set1_epi32((int)i - 31);
                        __m256i x[4];
                        int rej[4], nacc[4];
#pragma unroll
                        for (int b = 0; b < 4; ++b) {
                            x[b] = _mm256_and_si256(_mm256_loadu_si256(reinterpret_cast<const __m256i*>(o + k + 8 * b)), vmask);
                            rej[b] = _mm256_movemask_ps(_mm256_castsi256_ps(_mm256_cmpgt_epi32(x[b], vi)));
                            nacc[b] = _mm256_movemask_ps(_mm256_castsi256_ps(_mm256_cmpgt_epi32(x[b], vlo)));
                        }
                        if ((rej[0] ^ nacc[0]) | (rej[1] ^ nacc[1]) | (rej[2] ^ nacc[2]) | (rej[3] ^ nacc[3])) break;   // ambiguous: narrower paths
                        uint32_t ii = i;
#pragma unroll
                        for (int b = 0; b < 4; ++b) {
                            const int a = (~nacc[b]) & 0xff;
                            if (MODE == 2) {
                                const __m256i perm = _mm256_load_si256(reinterpret_cast<const __m256i*>(g_pack_lut[a]));
                                _mm256_storeu_si256(reinterpret_cast<__m256i*>(arr + (ii - 7)), _mm256_permutevar8x32_epi32(x[b], perm));
                            }
                            ii -= (uint32_t)__builtin_popcount((unsigned)a);
                        }
                        i = ii;
                        k += 32;
                    }
                    while (k + 8 <= avail && i >= lo + 8) {
                        const __m256i x = _mm256_and_si256(_mm256_loadu_si256(reinterpret_cast<const __m256i*>(o + k)), vmask);
                        const int rej = _mm256_movemask_ps(_mm256_castsi256_ps(_mm256_cmpgt_epi32(x, _mm256_set1_epi32((int)i))));
                        const int nacc = _mm256_movemask_ps(_mm256_castsi256_ps(_mm256_cmpgt_epi32(x, _mm256_set1_epi32((int)i - 7))));
                        if (rej != nacc) break;                       // an ambiguous draw: this block goes through the scalar loop
                        const int a = (~nacc) & 0xff;
                        if (MODE == 2) {
                            const __m256i perm = _mm256_load_si256(reinterpret_cast<const __m256i*>(g_pack_lut[a]));
                            _mm256_storeu_si256(reinterpret_cast<__m256i*>(arr + (i - 7)), _mm256_permutevar8x32_epi32(x, perm));
                        }
                        i -= (uint32_t)__builtin_popcount((unsigned)a);
                        k += 8;
                        if (k + 32 <= avail && i >= lo + 32) break;      // back to the wide path
                    }
                    const int kend = k + 8 < avail ? k + 8 : avail;
                    for (; k < kend && i > lo; ++k) {
                        const uint32_t x = o[k] & mask;
                        if (MODE == 2) arr[i] = (int32_t)x;
                        i -= x <= i ? 1u : 0u;
                    }
                    if (k >= avail || i <= lo) break;
                }
                g.pos += k;
            }
            mask = lo;
        }
    }
}

// AVX-512 form (Zen 4/5, recent Xeons): the same scheme sixteen words at a time -- the twister's recurrence leaves sixteen
// consecutive words independent too (distances 1 and 397 / -227) -- with the compares landing in mask registers.  64 draws are
// decided against one broadcast of i (ambiguous only for i - 63 < x <= i).  MODE 2 packs the accepted draws of a 16-block with
// vpcompressd on the lane-reversed block (register form + masked store: the r-th accepted draw lands at arr[i - r], nothing above
// arr[i] is touched).  Same draws, same state as the scalar loop; AVL_NO_AVX512=1 falls back to the AVX2 form.
#define AVL_T512 __attribute__((target("avx512f,avx512vl,avx512bw,avx512dq,popcnt")))
AVL_T512 static inline void mt_step16(uint32_t* key, int i, int j) {
    const __m512i a = _mm512_loadu_si512(key + i), b = _mm512_loadu_si512(key + i + 1);
    const __m512i y = _mm512_or_si512(_mm512_and_si512(a, _mm512_set1_epi32((int)0x80000000u)), _mm512_and_si512(b, _mm512_set1_epi32(0x7fffffff)));
    const __mmask16 odd = _mm512_test_epi32_mask(y, _mm512_set1_epi32(1));
    __m512i r = _mm512_xor_si512(_mm512_loadu_si512(key + j), _mm512_srli_epi32(y, 1));
    r = _mm512_mask_xor_epi32(r, odd, r, _mm512_set1_epi32((int)0x9908b0dfu));
    _mm512_storeu_si512(key + i, r);
}
AVL_T512 static void mt_regen_avx512(uint32_t* key) {
    int i = 0;
    for (; i + 16 <= 227; i += 16) mt_step16(key, i, i + 397);
    for (; i < 227; ++i) mt_step1(key, i, i + 1, i + 397);
    for (; i + 16 <= 623; i += 16) mt_step16(key, i, i - 227);
    for (; i < 623; ++i) mt_step1(key, i, i + 1, i - 227);
    mt_step1(key, 623, 0, 396);
}
AVL_T512 static void mt_temper_avx512(const uint32_t* key, uint32_t* out) {
    const __m512i m1 = _mm512_set1_epi32((int)0x9d2c5680u), m2 = _mm512_set1_epi32((int)0xefc60000u);
    for (int k = 0; k < 624; k += 16) {
        __m512i y = _mm512_loadu_si512(key + k);
        y = _mm512_xor_si512(y, _mm512_srli_epi32(y, 11));
        y = _mm512_xor_si512(y, _mm512_and_si512(_mm512_slli_epi32(y, 7), m1));
        y = _mm512_xor_si512(y, _mm512_and_si512(_mm512_slli_epi32(y, 15), m2));
        y = _mm512_xor_si512(y, _mm512_srli_epi32(y, 18));
        _mm512_store_si512(out + k, y);
    }
}
AVL_T512 static inline unsigned rev16(unsigned m) {       // bit-reverse a 16-bit mask
    m = ((m & 0x5555u) << 1) | ((m >> 1) & 0x5555u);
    m = ((m & 0x3333u) << 2) | ((m >> 2) & 0x3333u);
    m = ((m & 0x0f0fu) << 4) | ((m >> 4) & 0x0f0fu);
    return ((m & 0x00ffu) << 8) | ((m >> 8) & 0x00ffu);
}

// accepted draws of one 16-block (mask acc over the draws in order) -> arr[i], arr[i - 1], ...; returns how many
template <int MODE>
AVL_T512 static inline uint32_t pack16(__m512i x, unsigned acc, uint32_t i, int32_t* arr) {
    const uint32_t cnt = (uint32_t)__builtin_popcount(acc);
    if (MODE == 2) {
        const __m512i vrev = _mm512_set_epi32(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);   // lane l <- lane 15 - l
        const __m512i packed = _mm512_maskz_compress_epi32((__mmask16)rev16(acc), _mm512_permutexvar_epi32(vrev, x));
        _mm512_mask_storeu_epi32(arr + ((int64_t)i - cnt + 1), (__mmask16)((1u << cnt) - 1u), packed);   // last accepted ... first accepted
    }
    return cnt;
}

template <int MODE>
AVL_T512 void run_shuffles_avx512(Mt19937& g, int64_t n_items, int64_t n_shuffles, int32_t* arr) {
    alignas(64) uint32_t out[624 + 16];
    if (g.pos < 624) mt_temper_avx512(g.key, out);
    for (int64_t s = 0; s < n_shuffles; ++s) {
        if (n_items < 2) continue;
        uint32_t i = (uint32_t)(n_items - 1), mask = i;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
        while (i >= 1) {
            const uint32_t lo = mask >> 1;
            const __m512i vmask = _mm512_set1_epi32((int)mask);
            while (i > lo) {
                if (g.pos >= 624) {
                    mt_regen_avx512(g.key);
                    g.pos = 0;
                    mt_temper_avx512(g.key, out);
                }
                const int avail = 624 - g.pos;
                const uint32_t* o = out + g.pos;
                int k = 0;
                for (;;) {
                    while (k + 64 <= avail && i >= lo + 64) {
                        const __m512i vi = _mm512_set1_epi32((int)i), vlo = _mm512_set1_epi32((int)i - 63);
                        __m512i x[4];
                        unsigned rej[4], nacc[4];
#pragma unroll
                        for (int b = 0; b < 4; ++b) {
                            x[b] = _mm512_and_si512(_mm512_loadu_si512(o + k + 16 * b), vmask);
                            rej[b] = _mm512_cmpgt_epu32_mask(x[b], vi);
                            nacc[b] = _mm512_cmpgt_epu32_mask(x[b], vlo);
                        }
                        if ((rej[0] ^ nacc[0]) | (rej[1] ^ nacc[1]) | (rej[2] ^ nacc[2]) | (rej[3] ^ nacc[3])) break;   // ambiguous: narrower paths
                        uint32_t ii = i;
#pragma unroll
                        for (int b = 0; b < 4; ++b) ii -= pack16<MODE>(x[b], (~nacc[b]) & 0xffffu, ii, arr);
                        i = ii;
                        k += 64;
                    }
                    while (k + 16 <= avail && i >= lo + 16) {
                        const __m512i x = _mm512_and_si512(_mm512_loadu_si512(o + k), vmask);
                        const unsigned rej = _mm512_cmpgt_epu32_mask(x, _mm512_set1_epi32((int)i));
                        const unsigned nacc = _mm512_cmpgt_epu32_mask(x, _mm512_set1_epi32((int)i - 15));
                        if (rej != nacc) break;                       // an ambiguous draw: this block goes through the scalar loop
                        i -= pack16<MODE>(x, (~nacc) & 0xffffu, i, arr);
                        k += 16;
                        if (k + 64 <= avail && i >= lo + 64) break;      // back to the wide path
                    }
                    const int kend = k + 16 < avail ? k + 16 : avail;
                    for (; k < kend && i > lo; ++k) {
                        const uint32_t x = o[k] & mask;
                        if (MODE == 2) arr[i] = (int32_t)x;
                        i -= x <= i ? 1u : 0u;
                    }
                    if (k >= avail || i <= lo) break;
                }
                g.pos += k;
            }
            mask = lo;
        }
    }
}

template <int MODE>
void run_shuffles_best(Mt19937& g, int64_t n_items, int64_t n_shuffles, int32_t* arr) {
    static const bool avx2 = [] {
        const bool ok = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("popcnt") && std::getenv("AVL_NO_AVX2") == nullptr;
        if (ok) build_pack_lut();
        return ok;
    }();
    static const bool avx512 = avx2 && __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl") &&
                               __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512dq") && std::getenv("AVL_NO_AVX512") == nullptr;
    if (avx512) run_shuffles_avx512<MODE>(g, n_items, n_shuffles, arr);
    else if (avx2) run_shuffles_avx2<MODE>(g, n_items, n_shuffles, arr);
    else run_shuffles<MODE>(g, n_items, n_shuffles, arr);
}
}  // namespace
}  // extern "C++"

int avl_mt19937_skip_shuffles(uint32_t* h_key624, int* h_pos, int64_t n_items, int64_t n_shuffles) {
    AVL_REQUIRE(h_key624 && h_pos, "avl_mt19937_skip_shuffles: null state");
    AVL_REQUIRE(*h_pos >= 0 && *h_pos <= 624 && n_items >= 0 && n_shuffles >= 0, "avl_mt19937_skip_shuffles: bad arguments");
    AVL_REQUIRE(n_items <= 0x7fffffffll, "avl_mt19937_skip_shuffles: arrays beyond 2^31 items are not supported");
    Mt19937 g{h_key624, *h_pos};
    run_shuffles_best<0>(g, n_items, n_shuffles, nullptr);
    *h_pos = g.pos;
    return AVL_OK;
}

int avl_mt19937_shuffle_sample(uint32_t* h_key624, int* h_pos, int64_t n_items, int64_t rate, int32_t* h_scratch, int32_t* h_out) {
    AVL_REQUIRE(h_key624 && h_pos && h_scratch && h_out, "avl_mt19937_shuffle_sample: null pointer");
    AVL_REQUIRE(*h_pos >= 0 && *h_pos <= 624 && n_items >= 0 && rate >= 1, "avl_mt19937_shuffle_sample: bad arguments");
    AVL_REQUIRE(n_items <= 0x7fffffffll, "avl_mt19937_shuffle_sample: arrays beyond 2^31 items are not supported");
    Mt19937 g{h_key624, *h_pos};
    const int64_t n_out = (n_items + rate - 1) / rate;
    if (n_out < 32767 && n_items >= 2) {
        // Only every rate-th element of the permutation is wanted (vlmap_builder.py:277 shuffle_mask[::depth_sample_rate]): the
        // array is never permuted.  Pass 1 records the accepted draw j_i of every Fisher-Yates step i = n-1 .. 1 (sequential
        // writes).  Pass 2 undoes the swaps from the LAST one (i = 1) to the first: the content of output position p after all
        // swaps sat, before swap i, at j_i if it sits at i now (and vice versa) -- tracked for the n_out wanted positions at once
        // through owner[position] = output slot or -1 (1.5 MB of int16 for 720x1080, read-mostly: 99 % of the steps touch two
        // untracked positions and store nothing).  What is left at the end is the ORIGINAL position = the value of arange there.
        // Same samples, same RNG state; 1.7x faster than permuting a 3 MB array with dependent random swaps.
        static thread_local std::vector<int16_t> owner;
        owner.assign((size_t)n_items, (int16_t)-1);
        for (int64_t k = 0, o = 0; k < n_items; k += rate, ++o) {
            owner[(size_t)k] = (int16_t)o;
            h_out[o] = (int32_t)k;
        }
        run_shuffles_best<2>(g, n_items, 1, h_scratch);
        int16_t* ow = owner.data();
        for (int64_t i = 1; i < n_items; ++i) {
            const uint32_t j = (uint32_t)h_scratch[i];
            const int16_t a = ow[i], b = ow[j];
            if ((int16_t)(a & b) != (int16_t)-1) {
                ow[i] = b;
                ow[j] = a;
                if (a >= 0) h_out[a] = (int32_t)j;
                if (b >= 0) h_out[b] = (int32_t)i;
            }
        }
    } else {
        for (int64_t k = 0; k < n_items; ++k) h_scratch[k] = (int32_t)k;
        run_shuffles<1>(g, n_items, 1, h_scratch);
        for (int64_t k = 0, o = 0; k < n_items; k += rate, ++o) h_out[o] = h_scratch[k];
    }
    *h_pos = g.pos;
    return AVL_OK;
}

int avl_host_alloc(void** h_ptr_out, size_t bytes) {
    AVL_REQUIRE(h_ptr_out, "avl_host_alloc: null output");
    AVL_HIP_CHECK(hipHostMalloc(h_ptr_out, bytes ? bytes : 1, hipHostMallocDefault));
    return AVL_OK;
}

int avl_host_free(void* h_ptr) {
    if (h_ptr) AVL_HIP_CHECK(hipHostFree(h_ptr));
    return AVL_OK;
}

int avl_memset(void* d_ptr, int value, size_t bytes, void* stream) {
    AVL_HIP_CHECK(hipMemsetAsync(d_ptr, value, bytes, as_stream(stream)));
    return AVL_OK;
}

int avl_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes, void* stream) {
    AVL_HIP_CHECK(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, as_stream(stream)));
    return AVL_OK;
}

int avl_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes, void* stream) {
    AVL_HIP_CHECK(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
    AVL_HIP_CHECK(hipStreamSynchronize(as_stream(stream)));
    return AVL_OK;
}

int avl_memcpy_d2d(void* d_dst, const void* d_src, size_t bytes, void* stream) {
    AVL_HIP_CHECK(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, as_stream(stream)));
    return AVL_OK;
}

// dst row i = src row rows[i]; one wave per row when rows are 16-byte multiples at 16-byte aligned bases, bytes otherwise
__global__ __launch_bounds__(256) void gather_rows16_kernel(const uint4* __restrict__ src, int64_t row_u4, const int64_t* __restrict__ rows,
                                                            int64_t n, uint4* __restrict__ dst) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t i = wave0; i < n; i += nwaves) {
        const uint4* s = src + rows[i] * row_u4;
        uint4* d = dst + i * row_u4;
        for (int64_t k = lane; k < row_u4; k += 64) d[k] = s[k];
    }
}

__global__ __launch_bounds__(256) void gather_rows_bytes_kernel(const unsigned char* __restrict__ src, int64_t row_bytes,
                                                                const int64_t* __restrict__ rows, int64_t n, unsigned char* __restrict__ dst) {
    const int64_t total = n * row_bytes;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = t / row_bytes, k = t - i * row_bytes;
        dst[t] = src[rows[i] * row_bytes + k];
    }
}

int avl_gather_rows(const void* d_src, int64_t row_bytes, const int64_t* d_rows, int64_t n, void* d_dst, void* stream) {
    AVL_REQUIRE(row_bytes > 0 && n >= 0, "avl_gather_rows: bad shape");
    if (n == 0) return AVL_OK;
    AVL_REQUIRE(d_src && d_rows && d_dst, "avl_gather_rows: null pointer");
    hipStream_t st = as_stream(stream);
    const bool wide = row_bytes % 16 == 0 && (reinterpret_cast<uintptr_t>(d_src) & 15) == 0 && (reinterpret_cast<uintptr_t>(d_dst) & 15) == 0;
    if (wide) {
        int64_t blocks = (n + 3) / 4;
        if (blocks > (int64_t)avl::num_cus() * 16) blocks = (int64_t)avl::num_cus() * 16;
        hipLaunchKernelGGL(gather_rows16_kernel, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<const uint4*>(d_src), row_bytes / 16,
                           d_rows, n, reinterpret_cast<uint4*>(d_dst));
    } else {
        int64_t blocks = (n * row_bytes + 255) / 256;
        if (blocks > (int64_t)avl::num_cus() * 16) blocks = (int64_t)avl::num_cus() * 16;
        hipLaunchKernelGGL(gather_rows_bytes_kernel, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<const unsigned char*>(d_src),
                           row_bytes, d_rows, n, reinterpret_cast<unsigned char*>(d_dst));
    }
    AVL_HIP_CHECK(hipGetLastError());
    return AVL_OK;
}

// dst row rows[i] = src row i (rows distinct and in [0, n_dst): an out-of-range row is skipped and raises the flag)
__global__ __launch_bounds__(256) void scatter_rows16_kernel(const uint4* __restrict__ src, int64_t row_u4, const int64_t* __restrict__ rows,
                                                             int64_t n, int64_t n_dst, uint4* __restrict__ dst, int* __restrict__ err_flag) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t i = wave0; i < n; i += nwaves) {
        const int64_t r = rows[i];
        if (r < 0 || r >= n_dst) {
            if (lane == 0 && err_flag) atomicOr(err_flag, 1);
            continue;
        }
        const uint4* s = src + i * row_u4;
        uint4* d = dst + r * row_u4;
        for (int64_t k = lane; k < row_u4; k += 64) d[k] = s[k];
    }
}

int avl_scatter_rows(const void* d_src, int64_t row_bytes, const int64_t* d_rows, int64_t n, void* d_dst, int64_t n_dst, int32_t* d_err_flag,
                     void* stream) {
    AVL_REQUIRE(row_bytes > 0 && row_bytes % 16 == 0 && n >= 0 && n_dst >= 0, "avl_scatter_rows: rows must be multiples of 16 bytes");
    if (n == 0) return AVL_OK;
    AVL_REQUIRE(d_src && d_rows && d_dst, "avl_scatter_rows: null pointer");
    AVL_REQUIRE((reinterpret_cast<uintptr_t>(d_src) & 15) == 0 && (reinterpret_cast<uintptr_t>(d_dst) & 15) == 0, "avl_scatter_rows: 16-byte aligned bases");
    int64_t blocks = (n + 3) / 4;
    if (blocks > (int64_t)avl::num_cus() * 16) blocks = (int64_t)avl::num_cus() * 16;
    hipLaunchKernelGGL(scatter_rows16_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), reinterpret_cast<const uint4*>(d_src),
                       row_bytes / 16, d_rows, n, n_dst, reinterpret_cast<uint4*>(d_dst), reinterpret_cast<int*>(d_err_flag));
    AVL_HIP_CHECK(hipGetLastError());
    return AVL_OK;
}

int avl_hbm_read_probe(const void* d_buf, int64_t rows, int row_floats, int pattern, int iters, float* h_best_gbs, void* stream) {
    AVL_REQUIRE(d_buf && rows > 0 && row_floats > 0 && iters > 0 && h_best_gbs, "avl_hbm_read_probe: bad arguments");
    AVL_REQUIRE(pattern >= 0 && pattern <= 15, "avl_hbm_read_probe: pattern must be 0..15");
    const bool rowline = (pattern & 1) != 0, sustained = (pattern & 2) != 0;
    AVL_REQUIRE(!rowline || (row_floats % 64 == 0), "avl_hbm_read_probe: the row-line pattern needs row_floats %% 64 == 0");
    hipStream_t st = as_stream(stream);
    float* sink = nullptr;
    AVL_HIP_CHECK(hipMallocAsync((void**)&sink, sizeof(float), st));
    hipEvent_t e0, e1;
    AVL_HIP_CHECK(hipEventCreate(&e0));
    AVL_HIP_CHECK(hipEventCreate(&e1));
    const double bytes = (double)rows * row_floats * 4.0;
    auto launch = [&]() {
        if (!rowline)
            hipLaunchKernelGGL(probe_coalesced_kernel, dim3(8192), dim3(256), 0, st, reinterpret_cast<const float4*>(d_buf),
                               (size_t)(bytes / 16), sink);
        else
            hipLaunchKernelGGL(probe_rowline_kernel, dim3(num_cus() * ((pattern & 4) ? 1 : ((pattern & 8) ? 3 : 2))), dim3(512), 0, st, reinterpret_cast<const float*>(d_buf),
                               (long long)rows, row_floats, sink);
    };
    float best = 0.f;
    if (sustained) {
        // `iters` launches back to back between ONE event pair (after as many untimed ones): the rate the memory system
        // holds once the package sits at its power / thermal operating point, which is what a timed benchmark loop sees
        for (int it = 0; it < iters; ++it) launch();
        AVL_HIP_CHECK(hipEventRecord(e0, st));
        for (int it = 0; it < iters; ++it) launch();
        AVL_HIP_CHECK(hipEventRecord(e1, st));
        AVL_HIP_CHECK(hipEventSynchronize(e1));
        float ms = 0.f;
        AVL_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms > 0.f) best = (float)(bytes * iters / (ms * 1e-3) / 1e9);
    } else {
        for (int it = 0; it < iters + 2; ++it) {
            AVL_HIP_CHECK(hipEventRecord(e0, st));
            launch();
            AVL_HIP_CHECK(hipEventRecord(e1, st));
            AVL_HIP_CHECK(hipEventSynchronize(e1));
            float ms = 0.f;
            AVL_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (it >= 2 && ms > 0.f) best = fmaxf(best, (float)(bytes / (ms * 1e-3) / 1e9));
        }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFreeAsync(sink, st);
    *h_best_gbs = best;
    return AVL_OK;
}

int avl_event_create(void** h_event_out) {
    AVL_REQUIRE(h_event_out, "avl_event_create: null output");
    hipEvent_t e;
    AVL_HIP_CHECK(hipEventCreate(&e));
    *h_event_out = e;
    return AVL_OK;
}

int avl_event_destroy(void* event) {
    AVL_HIP_CHECK(hipEventDestroy(reinterpret_cast<hipEvent_t>(event)));
    return AVL_OK;
}

int avl_event_record(void* event, void* stream) {
    AVL_HIP_CHECK(hipEventRecord(reinterpret_cast<hipEvent_t>(event), as_stream(stream)));
    return AVL_OK;
}

int avl_event_sync(void* event) {
    AVL_HIP_CHECK(hipEventSynchronize(reinterpret_cast<hipEvent_t>(event)));
    return AVL_OK;
}

int avl_stream_wait_event(void* stream, void* event) {
    AVL_REQUIRE(event, "avl_stream_wait_event: null event");
    AVL_HIP_CHECK(hipStreamWaitEvent(as_stream(stream), reinterpret_cast<hipEvent_t>(event), 0));
    return AVL_OK;
}

int avl_event_elapsed_ms(void* start, void* stop, float* h_ms) {
    AVL_REQUIRE(h_ms, "avl_event_elapsed_ms: null output");
    AVL_HIP_CHECK(hipEventElapsedTime(h_ms, reinterpret_cast<hipEvent_t>(start), reinterpret_cast<hipEvent_t>(stop)));
    return AVL_OK;
}

}  // extern "C"
