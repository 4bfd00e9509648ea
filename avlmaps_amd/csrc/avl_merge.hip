// The local steps of the multi-GPU merge plan (avlmaps_amd/parallel.py: plan_merge_directory) as a few kernels each.
//
// What is merged: the ranks' voxel maps of the reference's builder loop (avlmaps/map/vlmap_builder.py:102-183), frames sharded
// contiguously over the ranks; the plan decides every voxel's final row = the reference's voxel id (vlmap_builder.py:163-170:
// ids in first-touch order).  The choreography -- which rank talks to which, and when -- stays in parallel.py (torch.distributed
// carries the collectives); between two collectives a rank's work used to be 20-40 small tensor operations, each a launch over
// 8-byte elements.  Here every such stretch is one entry point: a radix sort over just the bits the keys have and one or two
// kernels with LDS histograms, no host synchronisation inside, scratch owned by the caller.  The tensor code remains as the CPU
// twin (gloo tests) and is what the GPU tests compare these kernels with.
#include <algorithm>
#include <cstdint>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include "avl_common.h"

namespace avl {

constexpr int kMaxRanks = 64;

__device__ __forceinline__ int dir_owner(int32_t cell, int ws) {
    // parallel._dir_owner: (((cell * 2654435761) & 0xFFFFFFFF) >> 12) % ws  (a multiplicative hash: a map in one corner of the grid
    // still spreads evenly over the directory ranks)
    const unsigned long long h = ((unsigned long long)(long long)cell * 2654435761ull) & 0xFFFFFFFFull;
    return (int)((h >> 12) % (unsigned long long)ws);
}

// head = [sc (ws) | n | kmin | kmax]: cleared before the partition kernel accumulates into it
__global__ void merge_head_init_kernel(long long* __restrict__ head, int ws, long long n) {
    const int i = threadIdx.x;
    if (i < ws) head[i] = 0;
    if (i == 0) {
        head[ws] = n;
        head[ws + 1] = 0x7FFFFFFFFFFFFFFFll;
        head[ws + 2] = -1;
    }
}

// directory rank of every local voxel (sort key), its index (sort value), the per-rank counts and the range of the first-touch keys
__global__ __launch_bounds__(256) void merge_partition_kernel(long long n, const int32_t* __restrict__ cell, const long long* __restrict__ key,
                                                              int ws, uint32_t* __restrict__ dest, long long* __restrict__ iota,
                                                              long long* __restrict__ head) {
    __shared__ unsigned hist[kMaxRanks];
    __shared__ long long smin[4], smax[4];
    if (threadIdx.x < kMaxRanks) hist[threadIdx.x] = 0;
    __syncthreads();
    long long kmin = 0x7FFFFFFFFFFFFFFFll, kmax = -1;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int d = dir_owner(cell[i], ws);
        dest[i] = (uint32_t)d;
        iota[i] = i;
        atomicAdd(&hist[d], 1u);
        const long long k = key[i];
        kmin = k < kmin ? k : kmin;
        kmax = k > kmax ? k : kmax;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const long long a = __shfl_xor(kmin, off, 64), b = __shfl_xor(kmax, off, 64);
        kmin = a < kmin ? a : kmin;
        kmax = b > kmax ? b : kmax;
    }
    if ((threadIdx.x & 63) == 0) {
        smin[threadIdx.x >> 6] = kmin;
        smax[threadIdx.x >> 6] = kmax;
    }
    __syncthreads();
    if (threadIdx.x < ws && hist[threadIdx.x]) atomicAdd(reinterpret_cast<unsigned long long*>(&head[threadIdx.x]), (unsigned long long)hist[threadIdx.x]);
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) {
            kmin = smin[w] < kmin ? smin[w] : kmin;
            kmax = smax[w] > kmax ? smax[w] : kmax;
        }
        atomicMin(&head[ws + 1], kmin);
        atomicMax(&head[ws + 2], kmax);
    }
}

__global__ void merge_gather_cells_kernel(long long n, const int32_t* __restrict__ cell, const long long* __restrict__ ordd,
                                          int32_t* __restrict__ cell_sorted) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        cell_sorted[i] = cell[ordd[i]];
}

__global__ void merge_iota_kernel(long long n, long long* __restrict__ v) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) v[i] = i;
}

// Directory side: the arrivals sorted by (cell, source rank) -- cs = sorted cells, perm = their arrival positions -- answer every
// entry with the neighbouring contributors of its cell.  cnt = [n3r (ws) | n4r (ws) | number of distinct cells].
__global__ __launch_bounds__(256) void merge_dir_scan_kernel(long long R, const uint32_t* __restrict__ cs, const long long* __restrict__ perm,
                                                             const long long* __restrict__ rc, int ws, uint8_t* __restrict__ first,
                                                             long long* __restrict__ prev_r, long long* __restrict__ next_r,
                                                             int32_t* __restrict__ reply, uint8_t* __restrict__ m3r, uint8_t* __restrict__ m4r,
                                                             unsigned long long* __restrict__ cnt) {
    __shared__ long long ends[kMaxRanks];          // running sum of the per-source counts: arrival a came from the first rank with a < ends
    __shared__ unsigned h3[kMaxRanks], h4[kMaxRanks], hfirst;
    if (threadIdx.x == 0) {
        long long run = 0;
        for (int r = 0; r < ws; ++r) {
            run += rc[r];
            ends[r] = run;
        }
        hfirst = 0;
    }
    if (threadIdx.x < kMaxRanks) h3[threadIdx.x] = h4[threadIdx.x] = 0;
    __syncthreads();
    auto src_of = [&](long long a) {
        int lo = 0, hi = ws - 1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (a < ends[mid]) hi = mid;
            else lo = mid + 1;
        }
        return lo;
    };
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < R; p += (long long)gridDim.x * blockDim.x) {
        const uint32_t c = cs[p];
        const long long a = perm[p];
        const bool f = p == 0 || cs[p - 1] != c, l = p == R - 1 || cs[p + 1] != c;
        const long long pv = f ? -1 : (long long)src_of(perm[p - 1]), nx = l ? -1 : (long long)src_of(perm[p + 1]);
        first[p] = f ? 1 : 0;
        prev_r[a] = pv;
        next_r[a] = nx;
        reply[a] = (int32_t)((pv + 1) | ((nx + 1) << 16));
        const bool is3 = pv < 0 && nx >= 0, is4 = pv >= 0;
        m3r[a] = is3 ? 1 : 0;
        m4r[a] = is4 ? 1 : 0;
        const int s = src_of(a);
        if (is3) atomicAdd(&h3[s], 1u);
        if (is4) atomicAdd(&h4[s], 1u);
        if (f) atomicAdd(&hfirst, 1u);
    }
    __syncthreads();
    if (threadIdx.x < ws) {
        if (h3[threadIdx.x]) atomicAdd(&cnt[threadIdx.x], (unsigned long long)h3[threadIdx.x]);
        if (h4[threadIdx.x]) atomicAdd(&cnt[ws + threadIdx.x], (unsigned long long)h4[threadIdx.x]);
    }
    if (threadIdx.x == 0 && hfirst) atomicAdd(&cnt[2 * ws], (unsigned long long)hfirst);
}

// Back on the voxels' rank: the directory's answers (in sending order) -> prev / next per slot, the masks of the two lists of the
// second directory round trip (in sending order) and every count the next all_gather carries:
// cnt = [new voxels | n3 (ws) | n4 (ws) | n_prev (ws) | n_next (ws)].
__global__ __launch_bounds__(256) void merge_classify_kernel(long long n, const int32_t* __restrict__ back, const long long* __restrict__ ordd,
                                                             const int32_t* __restrict__ cell_sorted, int ws, long long* __restrict__ prev,
                                                             long long* __restrict__ nxt, uint8_t* __restrict__ is_new,
                                                             uint8_t* __restrict__ m3, uint8_t* __restrict__ m4,
                                                             unsigned long long* __restrict__ cnt) {
    __shared__ unsigned h3[kMaxRanks], h4[kMaxRanks], hp[kMaxRanks], hn[kMaxRanks], hnew;
    if (threadIdx.x < kMaxRanks) h3[threadIdx.x] = h4[threadIdx.x] = hp[threadIdx.x] = hn[threadIdx.x] = 0;
    if (threadIdx.x == 0) hnew = 0;
    __syncthreads();
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long b = (long long)back[i];
        const long long pv = (b & 0xFFFF) - 1, nx = (b >> 16) - 1;
        const long long slot = ordd[i];
        prev[slot] = pv;
        nxt[slot] = nx;
        const bool nw = pv < 0;
        is_new[slot] = nw ? 1 : 0;
        const bool is3 = nw && nx >= 0, is4 = !nw;
        m3[i] = is3 ? 1 : 0;
        m4[i] = is4 ? 1 : 0;
        const int d = dir_owner(cell_sorted[i], ws);
        if (is3) atomicAdd(&h3[d], 1u);
        if (is4) atomicAdd(&h4[d], 1u);
        if (pv >= 0 && pv < ws) atomicAdd(&hp[pv], 1u);
        if (nx >= 0 && nx < ws) atomicAdd(&hn[nx], 1u);
        if (nw) atomicAdd(&hnew, 1u);
    }
    __syncthreads();
    if (threadIdx.x < ws) {
        const int r = threadIdx.x;
        if (h3[r]) atomicAdd(&cnt[1 + r], (unsigned long long)h3[r]);
        if (h4[r]) atomicAdd(&cnt[1 + ws + r], (unsigned long long)h4[r]);
        if (hp[r]) atomicAdd(&cnt[1 + 2 * ws + r], (unsigned long long)hp[r]);
        if (hn[r]) atomicAdd(&cnt[1 + 3 * ws + r], (unsigned long long)hn[r]);
    }
    if (threadIdx.x == 0 && hnew) atomicAdd(&cnt[0], (unsigned long long)hnew);
}

// ---- second half of the plan: rows of the new voxels, and the two lists through which shared voxels learn theirs

struct FlagToCount {
    __device__ long long operator()(uint8_t v) const { return v ? 1 : 0; }
};

// row = -1 everywhere; the new voxels (is_new) compacted in slot order with their first-touch keys
__global__ void merge_new_compact_kernel(long long n, const uint8_t* __restrict__ is_new, const long long* __restrict__ off,
                                         const long long* __restrict__ key, long long* __restrict__ row, unsigned long long* __restrict__ ukeys,
                                         long long* __restrict__ uidx) {
    for (long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x; s < n; s += (long long)gridDim.x * blockDim.x) {
        row[s] = -1;
        if (is_new[s]) {
            const long long j = off[s];
            ukeys[j] = (unsigned long long)key[s];
            uidx[j] = s;
        }
    }
}

__global__ void merge_new_rows_kernel(long long c, const long long* __restrict__ idx_new, long long base, long long* __restrict__ row) {
    for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < c; j += (long long)gridDim.x * blockDim.x) row[idx_new[j]] = base + j;
}

// out[off[i]] = src[via ? via[i] : i] for every flagged i (a compaction that keeps the order)
__global__ void merge_compact_rows_kernel(long long n, const uint8_t* __restrict__ flag, const long long* __restrict__ off,
                                          const long long* __restrict__ via, const long long* __restrict__ src, long long* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        if (flag[i]) out[off[i]] = src[via ? via[i] : i];
}

// dst[via[i]] = src[off[i]] for every flagged i (the reverse: a list in flag order placed back)
__global__ void merge_place_rows_kernel(long long n, const uint8_t* __restrict__ flag, const long long* __restrict__ off,
                                        const long long* __restrict__ via, const long long* __restrict__ src, long long* __restrict__ dst) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        if (flag[i]) dst[via[i]] = src[off[i]];
}

__global__ void merge_head_pos_kernel(long long R, const uint8_t* __restrict__ first, long long* __restrict__ v) {
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < R; p += (long long)gridDim.x * blockDim.x) v[p] = first[p] ? p : 0;
}

// every directory entry gets the row its cell's FIRST contributor reported (-1 if that one has not: it is not a sender of list 3)
__global__ void merge_propagate_kernel(long long R, const long long* __restrict__ perm, const long long* __restrict__ head_pos,
                                       const uint8_t* __restrict__ m3r, const long long* __restrict__ off3r, const long long* __restrict__ recv3,
                                       long long* __restrict__ row_r) {
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < R; p += (long long)gridDim.x * blockDim.x) {
        const long long ah = perm[head_pos[p]];
        row_r[perm[p]] = m3r[ah] ? recv3[off3r[ah]] : -1;
    }
}

// MixedExchange's side record of every local voxel, in final-row order: 64 B = [row | cell << 32 | single << 63, sum_w4 (4 x f64),
// replay state (3 x i64), zeros unless next[voxel] < 0] -- four lanes per record, 16 B each (coalesced stores; the gathers by `order` are 32 / 24 B rows)
__global__ void merge_side_pack_kernel(long long n, const long long* __restrict__ order, const long long* __restrict__ rows_sorted,
                                       const uint8_t* __restrict__ single_sorted, const int32_t* __restrict__ cell,
                                       const long long* __restrict__ w4, const long long* __restrict__ state, const long long* __restrict__ next,
                                       long long* __restrict__ side) {
    using i64x2 = __attribute__((ext_vector_type(2))) long long;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < 4 * n; t += (long long)gridDim.x * blockDim.x) {
        const long long i = t >> 2, o = order[i];
        const int p = (int)(t & 3);
        const bool st = state && (!next || next[o] < 0);      // only a voxel's LAST contributor (no higher rank holds it) sends its state
        long long a, b;
        if (p == 0) {
            a = rows_sorted[i] | ((long long)cell[o] << 32) | (single_sorted[i] ? (long long)(1ull << 63) : 0ll);
            b = w4[4 * o];
        } else if (p == 1) {
            a = w4[4 * o + 1];
            b = w4[4 * o + 2];
        } else if (p == 2) {
            a = w4[4 * o + 3];
            b = st ? state[3 * o] : 0ll;
        } else {
            a = st ? state[3 * o + 1] : 0ll;
            b = st ? state[3 * o + 2] : 0ll;
        }
        *reinterpret_cast<i64x2*>(side + 2 * t) = i64x2{a, b};
    }
}

// owner side: the records that arrived -> row (relative to the block, clamped), the block's cells, the final replay states (only a
// voxel's LAST contributor sends one: `started` in the high half of the third word).  A row outside the block sets bit 0 of *err.
__global__ void merge_side_unpack_kernel(long long R, const long long* __restrict__ side, long long r0, long long n_own, long long* __restrict__ rows,
                                         int32_t* __restrict__ own_cell, long long* __restrict__ state, int* __restrict__ err) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < R; i += (long long)gridDim.x * blockDim.x) {
        const long long word = side[8 * i];
        long long row = (word & 0xFFFFFFFFll) - r0;
        if (row < 0 || row >= n_own) {
            if (err) atomicOr(err, 1);
            rows[i] = row < 0 ? 0 : (n_own > 0 ? n_own - 1 : 0);
            continue;
        }
        rows[i] = row;
        own_cell[row] = (int32_t)((word >> 32) & 0x7FFFFFFFll);
        const long long s2 = side[8 * i + 7];
        if (((unsigned long long)s2 >> 32) != 0) {
            state[3 * row] = side[8 * i + 5];
            state[3 * row + 1] = side[8 * i + 6];
            state[3 * row + 2] = s2;
        }
    }
}

static size_t al256(size_t b) { return (b + 255) / 256 * 256; }

static hipError_t scan_flags(void* tmp, size_t& tmp_bytes, const uint8_t* flags, long long* out, long long n, hipStream_t st) {
    auto it = rocprim::make_transform_iterator(flags, FlagToCount{});
    return rocprim::exclusive_scan(tmp, tmp_bytes, it, out, 0ll, (size_t)n, rocprim::plus<long long>(), st);
}

static hipError_t scan_max(void* tmp, size_t& tmp_bytes, const long long* in, long long* out, long long n, hipStream_t st) {
    return rocprim::inclusive_scan(tmp, tmp_bytes, in, out, (size_t)n, rocprim::maximum<long long>(), st);
}

static hipError_t sort_u64_i64(void* tmp, size_t& tmp_bytes, const unsigned long long* k, unsigned long long* ko, const long long* v, long long* vo,
                               long long n, int bits, hipStream_t st) {
    return rocprim::radix_sort_pairs(tmp, tmp_bytes, k, ko, v, vo, (size_t)n, 0, bits, st);
}

static hipError_t sort_u32_i64(void* tmp, size_t& tmp_bytes, const uint32_t* k, uint32_t* ko, const long long* v, long long* vo, long long n,
                               int bits, hipStream_t st) {
    return rocprim::radix_sort_pairs(tmp, tmp_bytes, k, ko, v, vo, (size_t)n, 0, bits, st);
}

static unsigned grid_for(long long n) { return (unsigned)std::min<long long>((n + 255) / 256, 2048); }

}  // namespace avl

using namespace avl;

extern "C" {

int avl_merge_work_bytes(int64_t n, size_t* h_bytes) {
    AVL_REQUIRE(h_bytes && n >= 0 && n < (1ll << 31), "avl_merge_work_bytes: bad arguments");
    size_t t1 = 0, t32 = 0;
    const long long m = n ? n : 1;
    AVL_HIP_CHECK(sort_u32_i64(nullptr, t1, nullptr, nullptr, nullptr, nullptr, m, 8, nullptr));
    AVL_HIP_CHECK(sort_u32_i64(nullptr, t32, nullptr, nullptr, nullptr, nullptr, m, 31, nullptr));
    // [keys u32 | sorted keys u32 | values i64 | rocPRIM's storage (the larger of the narrow and the wide sort)]
    *h_bytes = 2 * al256((size_t)m * 4) + al256((size_t)m * 8) + al256(std::max(t1, t32)) + 512;
    return AVL_OK;
}

struct MergeWork {
    uint32_t *keys, *keys_out;
    long long* iota;
    void* tmp;
    size_t tmp_bytes;
};

static int carve_work(int64_t n, void* d_work, size_t work_bytes, MergeWork& w, const char* who) {
    size_t need = 0;
    int rc = avl_merge_work_bytes(n, &need);
    if (rc != AVL_OK) return rc;
    AVL_REQUIRE(d_work && work_bytes >= need, "%s: work buffer of %zu bytes, %zu needed (avl_merge_work_bytes)", who, work_bytes, need);
    const long long m = n ? n : 1;
    char* p = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(d_work) + 255) / 256 * 256);
    w.keys = reinterpret_cast<uint32_t*>(p); p += al256((size_t)m * 4);
    w.keys_out = reinterpret_cast<uint32_t*>(p); p += al256((size_t)m * 4);
    w.iota = reinterpret_cast<long long*>(p); p += al256((size_t)m * 8);
    w.tmp = p;
    w.tmp_bytes = need - 512 - 2 * al256((size_t)m * 4) - al256((size_t)m * 8);
    return AVL_OK;
}

int avl_merge_partition(int64_t n, const int32_t* d_cell, const int64_t* d_key, int ws, int64_t* d_ordd, int32_t* d_cell_sorted,
                        int64_t* d_head, void* d_work, size_t work_bytes, void* stream) {
    AVL_REQUIRE(n >= 0 && n < (1ll << 31) && ws >= 1 && ws <= kMaxRanks, "avl_merge_partition: bad arguments (at most %d ranks)", kMaxRanks);
    AVL_REQUIRE(d_head, "avl_merge_partition: null head");
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(merge_head_init_kernel, dim3(1), dim3(64), 0, st, reinterpret_cast<long long*>(d_head), ws, (long long)n);
    if (n == 0) {
        AVL_HIP_CHECK(hipGetLastError());
        return AVL_OK;
    }
    AVL_REQUIRE(d_cell && d_key && d_ordd && d_cell_sorted, "avl_merge_partition: null pointer");
    MergeWork w;
    int rc = carve_work(n, d_work, work_bytes, w, "avl_merge_partition");
    if (rc != AVL_OK) return rc;
    hipLaunchKernelGGL(merge_partition_kernel, dim3(grid_for(n)), dim3(256), 0, st, (long long)n, d_cell, reinterpret_cast<const long long*>(d_key),
                       ws, w.keys, w.iota, reinterpret_cast<long long*>(d_head));
    int bits = 1;
    while ((1 << bits) < ws) ++bits;
    size_t tb = w.tmp_bytes;
    AVL_HIP_CHECK(sort_u32_i64(w.tmp, tb, w.keys, w.keys_out, w.iota, reinterpret_cast<long long*>(d_ordd), n, bits, st));
    hipLaunchKernelGGL(merge_gather_cells_kernel, dim3(grid_for(n)), dim3(256), 0, st, (long long)n, d_cell,
                       reinterpret_cast<const long long*>(d_ordd), d_cell_sorted);
    AVL_HIP_CHECK(hipGetLastError());
    return AVL_OK;
}

int avl_merge_dir_scan(int64_t R, const int32_t* d_recv, const int64_t* d_rc, int ws, int cell_bits, int64_t* d_perm, uint8_t* d_first,
                       int64_t* d_prev_r, int64_t* d_next_r, int32_t* d_reply, uint8_t* d_m3r, uint8_t* d_m4r, int64_t* d_cnt,
                       void* d_work, size_t work_bytes, void* stream) {
    AVL_REQUIRE(R >= 0 && R < (1ll << 31) && ws >= 1 && ws <= kMaxRanks && cell_bits >= 1 && cell_bits <= 31, "avl_merge_dir_scan: bad arguments");
    AVL_REQUIRE(d_cnt && d_rc, "avl_merge_dir_scan: null pointer");
    hipStream_t st = as_stream(stream);
    AVL_HIP_CHECK(hipMemsetAsync(d_cnt, 0, (size_t)(2 * ws + 1) * sizeof(int64_t), st));
    if (R == 0) return AVL_OK;
    AVL_REQUIRE(d_recv && d_perm && d_first && d_prev_r && d_next_r && d_reply && d_m3r && d_m4r, "avl_merge_dir_scan: null pointer");
    MergeWork w;
    int rc = carve_work(R, d_work, work_bytes, w, "avl_merge_dir_scan");
    if (rc != AVL_OK) return rc;
    hipLaunchKernelGGL(merge_iota_kernel, dim3(grid_for(R)), dim3(256), 0, st, (long long)R, w.iota);
    size_t tb = w.tmp_bytes;
    // arrivals are grouped by source rank, so the STABLE sort by cell is the (cell, source rank) order
    AVL_HIP_CHECK(sort_u32_i64(w.tmp, tb, reinterpret_cast<const uint32_t*>(d_recv), w.keys_out, w.iota, reinterpret_cast<long long*>(d_perm), R,
                               cell_bits, st));
    hipLaunchKernelGGL(merge_dir_scan_kernel, dim3(grid_for(R)), dim3(256), 0, st, (long long)R, w.keys_out, reinterpret_cast<const long long*>(d_perm),
                       reinterpret_cast<const long long*>(d_rc), ws, d_first, reinterpret_cast<long long*>(d_prev_r),
                       reinterpret_cast<long long*>(d_next_r), d_reply, d_m3r, d_m4r, reinterpret_cast<unsigned long long*>(d_cnt));
    AVL_HIP_CHECK(hipGetLastError());
    return AVL_OK;
}

int avl_merge_classify(int64_t n, const int32_t* d_back, const int64_t* d_ordd, const int32_t* d_cell_sorted, int ws, int64_t* d_prev,
                       int64_t* d_next, uint8_t* d_is_new, uint8_t* d_m3, uint8_t* d_m4, int64_t* d_cnt, void* stream) {
    AVL_REQUIRE(n >= 0 && ws >= 1 && ws <= kMaxRanks && d_cnt, "avl_merge_classify: bad arguments");
    hipStream_t st = as_stream(stream);
    AVL_HIP_CHECK(hipMemsetAsync(d_cnt, 0, (size_t)(1 + 4 * ws) * sizeof(int64_t), st));
    if (n == 0) return AVL_OK;
    AVL_REQUIRE(d_back && d_ordd && d_cell_sorted && d_prev && d_next && d_is_new && d_m3 && d_m4, "avl_merge_classify: null pointer");
    hipLaunchKernelGGL(merge_classify_kernel, dim3(grid_for(n)), dim3(256), 0, st, (long long)n, d_back, reinterpret_cast<const long long*>(d_ordd),
                       d_cell_sorted, ws, reinterpret_cast<long long*>(d_prev), reinterpret_cast<long long*>(d_next), d_is_new, d_m3, d_m4,
                       reinterpret_cast<unsigned long long*>(d_cnt));
    AVL_HIP_CHECK(hipGetLastError());
    return AVL_OK;
}

int avl_merge_rows_work_bytes(int64_t n, size_t* h_bytes) {
    AVL_REQUIRE(h_bytes && n >= 0 && n < (1ll << 31), "avl_merge_rows_work_bytes: bad arguments");
    const long long m = n ? n : 1;
    size_t t_scan = 0, t_max = 0, t_sort = 0;
    AVL_HIP_CHECK(scan_flags(nullptr, t_scan, nullptr, nullptr, m, nullptr));
    AVL_HIP_CHECK(scan_max(nullptr, t_max, nullptr, nullptr, m, nullptr));
    AVL_HIP_CHECK(sort_u64_i64(nullptr, t_sort, nullptr, nullptr, nullptr, nullptr, m, 63, nullptr));
    // [offsets | keys | sorted keys | indices | a second offset / row vector | rocPRIM's storage]
    *h_bytes = 5 * al256((size_t)m * 8) + al256(std::max(std::max(t_scan, t_max), t_sort)) + 512;
    return AVL_OK;
}

struct RowsWork {
    long long *off, *idx, *aux;
    unsigned long long *keys, *keys_out;
    void* tmp;
    size_t tmp_bytes;
};

static int carve_rows_work(int64_t n, void* d_work, size_t work_bytes, RowsWork& w, const char* who) {
    size_t need = 0;
    int rc = avl_merge_rows_work_bytes(n, &need);
    if (rc != AVL_OK) return rc;
    AVL_REQUIRE(d_work && work_bytes >= need, "%s: work buffer of %zu bytes, %zu needed (avl_merge_rows_work_bytes)", who, work_bytes, need);
    const size_t b = al256((size_t)(n ? n : 1) * 8);
    char* p = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(d_work) + 255) / 256 * 256);
    w.off = reinterpret_cast<long long*>(p); p += b;
    w.keys = reinterpret_cast<unsigned long long*>(p); p += b;
    w.keys_out = reinterpret_cast<unsigned long long*>(p); p += b;
    w.idx = reinterpret_cast<long long*>(p); p += b;
    w.aux = reinterpret_cast<long long*>(p); p += b;
    w.tmp = p;
    w.tmp_bytes = need - 512 - 5 * b;
    return AVL_OK;
}

int avl_merge_rows_new(int64_t n, int64_t c, const uint8_t* d_is_new, const int64_t* d_key, int key_bits, int64_t base, const int64_t* d_ordd,
                       const uint8_t* d_m3, int64_t n3, int64_t* d_row, int64_t* d_idx_new, int64_t* d_send3, void* d_work, size_t work_bytes,
                       void* stream) {
    AVL_REQUIRE(n >= 0 && c >= 0 && c <= n && n3 >= 0 && n3 <= n && key_bits >= 1 && key_bits <= 63, "avl_merge_rows_new: bad arguments");
    if (n == 0) return AVL_OK;
    AVL_REQUIRE(d_is_new && d_key && d_ordd && d_m3 && d_row && (c == 0 || d_idx_new) && (n3 == 0 || d_send3), "avl_merge_rows_new: null pointer");
    hipStream_t st = as_stream(stream);
    RowsWork w;
    int rc = carve_rows_work(n, d_work, work_bytes, w, "avl_merge_rows_new");
    if (rc != AVL_OK) return rc;
    size_t tb = w.tmp_bytes;
    AVL_HIP_CHECK(scan_flags(w.tmp, tb, d_is_new, w.off, n, st));
    hipLaunchKernelGGL(merge_new_compact_kernel, dim3(grid_for(n)), dim3(256), 0, st, (long long)n, d_is_new, w.off,
                       reinterpret_cast<const long long*>(d_key), reinterpret_cast<long long*>(d_row), w.keys, w.idx);
    if (c > 0) {
        // the rank's new voxels in first-touch-key order = their order among the final rows
        tb = w.tmp_bytes;
        AVL_HIP_CHECK(sort_u64_i64(w.tmp, tb, w.keys, w.keys_out, w.idx, reinterpret_cast<long long*>(d_idx_new), c, key_bits, st));
        hipLaunchKernelGGL(merge_new_rows_kernel, dim3(grid_for(c)), dim3(256), 0, st, (long long)c, reinterpret_cast<const long long*>(d_idx_new),
                           (long long)base, reinterpret_cast<long long*>(d_row));
    }
    if (n3 > 0) {
        tb = w.tmp_bytes;
        AVL_HIP_CHECK(scan_flags(w.tmp, tb, d_m3, w.off, n, st));
        hipLaunchKernelGGL(merge_compact_rows_kernel, dim3(grid_for(n)), dim3(256), 0, st, (long long)n, d_m3, w.off,
                           reinterpret_cast<const long long*>(d_ordd), reinterpret_cast<const long long*>(d_row), reinterpret_cast<long long*>(d_send3));
    }
    AVL_HIP_CHECK(hipGetLastError());
    return AVL_OK;
}

int avl_merge_dir_rows(int64_t R, const int64_t* d_recv3, const uint8_t* d_m3r, const uint8_t* d_first, const int64_t* d_perm,
                       const uint8_t* d_m4r, int64_t n4, int64_t* d_send4, void* d_work, size_t work_bytes, void* stream) {
    AVL_REQUIRE(R >= 0 && n4 >= 0 && n4 <= R, "avl_merge_dir_rows: bad arguments");
    if (R == 0 || n4 == 0) return AVL_OK;
    AVL_REQUIRE(d_m3r && d_first && d_perm && d_m4r && d_send4, "avl_merge_dir_rows: null pointer");
    hipStream_t st = as_stream(stream);
    RowsWork w;
    int rc = carve_rows_work(R, d_work, work_bytes, w, "avl_merge_dir_rows");
    if (rc != AVL_OK) return rc;
    long long* off3r = w.off;
    long long* head_pos = w.idx;
    long long* row_r = w.aux;
    long long* off4r = reinterpret_cast<long long*>(w.keys);
    size_t tb = w.tmp_bytes;
    AVL_HIP_CHECK(scan_flags(w.tmp, tb, d_m3r, off3r, R, st));
    hipLaunchKernelGGL(merge_head_pos_kernel, dim3(grid_for(R)), dim3(256), 0, st, (long long)R, d_first, reinterpret_cast<long long*>(w.keys_out));
    tb = w.tmp_bytes;
    AVL_HIP_CHECK(scan_max(w.tmp, tb, reinterpret_cast<const long long*>(w.keys_out), head_pos, R, st));
    hipLaunchKernelGGL(merge_propagate_kernel, dim3(grid_for(R)), dim3(256), 0, st, (long long)R, reinterpret_cast<const long long*>(d_perm), head_pos,
                       d_m3r, off3r, reinterpret_cast<const long long*>(d_recv3), row_r);
    tb = w.tmp_bytes;
    AVL_HIP_CHECK(scan_flags(w.tmp, tb, d_m4r, off4r, R, st));
    hipLaunchKernelGGL(merge_compact_rows_kernel, dim3(grid_for(R)), dim3(256), 0, st, (long long)R, d_m4r, off4r, (const long long*)nullptr, row_r,
                       reinterpret_cast<long long*>(d_send4));
    AVL_HIP_CHECK(hipGetLastError());
    return AVL_OK;
}

int avl_merge_rows_other(int64_t n, const uint8_t* d_m4, const int64_t* d_ordd, const int64_t* d_recv4, int64_t n4, int64_t* d_row, void* d_work,
                         size_t work_bytes, void* stream) {
    AVL_REQUIRE(n >= 0 && n4 >= 0 && n4 <= n, "avl_merge_rows_other: bad arguments");
    if (n == 0 || n4 == 0) return AVL_OK;
    AVL_REQUIRE(d_m4 && d_ordd && d_recv4 && d_row, "avl_merge_rows_other: null pointer");
    hipStream_t st = as_stream(stream);
    RowsWork w;
    int rc = carve_rows_work(n, d_work, work_bytes, w, "avl_merge_rows_other");
    if (rc != AVL_OK) return rc;
    size_t tb = w.tmp_bytes;
    AVL_HIP_CHECK(scan_flags(w.tmp, tb, d_m4, w.off, n, st));
    hipLaunchKernelGGL(merge_place_rows_kernel, dim3(grid_for(n)), dim3(256), 0, st, (long long)n, d_m4, w.off, reinterpret_cast<const long long*>(d_ordd),
                       reinterpret_cast<const long long*>(d_recv4), reinterpret_cast<long long*>(d_row));
    AVL_HIP_CHECK(hipGetLastError());
    return AVL_OK;
}

int avl_merge_side_pack(int64_t n, const int64_t* d_order, const int64_t* d_rows_sorted, const uint8_t* d_single_sorted, const int32_t* d_cell,
                        const double* d_w4, const int64_t* d_state, const int64_t* d_next, int64_t* d_side, void* stream) {
    AVL_REQUIRE(n >= 0 && n < (1ll << 40), "avl_merge_side_pack: bad arguments");
    if (n == 0) return AVL_OK;
    AVL_REQUIRE(d_order && d_rows_sorted && d_single_sorted && d_cell && d_w4 && d_side, "avl_merge_side_pack: null pointer");
    hipLaunchKernelGGL(merge_side_pack_kernel, dim3(grid_for(4 * n)), dim3(256), 0, as_stream(stream), (long long)n,
                       reinterpret_cast<const long long*>(d_order), reinterpret_cast<const long long*>(d_rows_sorted), d_single_sorted, d_cell,
                       reinterpret_cast<const long long*>(d_w4), reinterpret_cast<const long long*>(d_state), reinterpret_cast<const long long*>(d_next),
                       reinterpret_cast<long long*>(d_side));
    AVL_HIP_CHECK(hipGetLastError());
    return AVL_OK;
}

int avl_merge_side_unpack(int64_t R, const int64_t* d_side, int64_t r0, int64_t n_own, int64_t* d_rows, int32_t* d_own_cell, int64_t* d_state,
                          int* d_err_flag, void* stream) {
    AVL_REQUIRE(R >= 0 && r0 >= 0 && n_own >= 0, "avl_merge_side_unpack: bad arguments");
    if (R == 0) return AVL_OK;
    AVL_REQUIRE(d_side && d_rows && d_own_cell && d_state, "avl_merge_side_unpack: null pointer");
    hipLaunchKernelGGL(merge_side_unpack_kernel, dim3(grid_for(R)), dim3(256), 0, as_stream(stream), (long long)R,
                       reinterpret_cast<const long long*>(d_side), (long long)r0, (long long)n_own, reinterpret_cast<long long*>(d_rows), d_own_cell,
                       reinterpret_cast<long long*>(d_state), d_err_flag);
    AVL_HIP_CHECK(hipGetLastError());
    return AVL_OK;
}

}  // extern "C"
