// The multi-GPU merge of the map build, second form ("gather plan"): everything a rank computes between two collectives is ONE
// entry point of this file (or avl_builder_m2_pack in avl_builder.hip, which needs the accumulators), with no host
// synchronisation inside except the single read-back of the plan's sizes.
//
// What is merged: the ranks' voxel maps of the reference's builder loop (avlmaps/map/vlmap_builder.py:102-183), frames sharded
// contiguously over the ranks.  A voxel's final row is the reference's voxel id = its position in first-touch order
// (vlmap_builder.py:163-170).  Protocol (avlmaps_amd/merge2.py carries the collectives through torch.distributed):
//   1. all_gather of a 4-word header per rank (voxel count, key range, flags)
//   2. all_gather of every rank's (first-touch key, cell) list, 12 B per voxel           -> avl_merge2_plan on EVERY rank:
//        the union of the cells (radix sort by cell; stable, so a cell's contributors stay in rank order), the first
//        contributor of every cell, the reference's row of every cell (radix sort of the first contributors' keys), and for the
//        rank's OWN voxels: final row, neighbouring contributors (prev / next rank), the send order, the replay lists; the
//        ws x ws tables of list sizes every later exchange needs are read back ONCE (the only host synchronisation)
//   3. the sequential weight / colour replay hops rank -> next rank for voxels several ranks touched (24 B of state each)
//   4. ONE all_to_all of the payload: per destination [side records 64 B | finished float32 rows | float64 partial rows]
//   5. avl_merge2_fold: wave per row of the rank's block -- contributors found by binary search in the peers' (row-sorted) side
//      lists, summed in rank order (reproducible float64 sums), divided, written; position / weight / colour from the side sums
//      or the replay state.
// The first plan (avl_merge.hip + parallel.plan_merge_directory: directory ranks, five small round trips, nothing O(sum n) per
// rank) stays as the fall-back; here every rank sorts all sum(n) entries (2.3 M entries at 8 ranks x 290 k voxels: two radix
// sorts, ~0.3 ms) and the round trips are gone.
#include <algorithm>
#include <cstdint>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include "avl_common.h"

namespace avl {

constexpr int kM2MaxRanks = 64;
constexpr unsigned long long kM2Single = 1ull << 63;   // side word: the voxel has ONE contributor (its feature row is finished float32)
constexpr unsigned long long kM2Direct = 1ull << 62;   // ... and that contributor owns the row: the row is already in the block

struct M2Offsets {
    long long off[kM2MaxRanks + 1];   // entry index of rank p's first voxel; off[ws] = E
};

// header of a rank: [voxel count, smallest first-touch key, largest, flags] -- one block; the keys of a rank are 12 B x 300 k voxels
__global__ __launch_bounds__(1024) void m2_header_kernel(long long n, const long long* __restrict__ key, long long flags, long long* __restrict__ hdr) {
    __shared__ long long smin[16], smax[16];
    long long kmin = 0x7FFFFFFFFFFFFFFFll, kmax = -1;
    for (long long i = threadIdx.x; i < n; i += blockDim.x) {
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
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) {
            kmin = smin[w] < kmin ? smin[w] : kmin;
            kmax = smax[w] > kmax ? smax[w] : kmax;
        }
        hdr[0] = n;
        hdr[1] = kmin;
        hdr[2] = kmax;
        hdr[3] = flags;
    }
}

// entry e = (rank p, slot s): its cell (sort key), its index (sort value), its rank
__global__ __launch_bounds__(256) void m2_compact_kernel(long long E, int ws, M2Offsets o, long long stride, long long nmax,
                                                         const long long* __restrict__ g, uint32_t* __restrict__ ecell,
                                                         uint32_t* __restrict__ eidx, uint8_t* __restrict__ erank) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (long long)gridDim.x * blockDim.x) {
        int p = 0;
        while (p + 1 < ws && e >= o.off[p + 1]) ++p;
        const long long s = e - o.off[p];
        const int32_t* cells = reinterpret_cast<const int32_t*>(g + p * stride + nmax);
        ecell[e] = (uint32_t)cells[s];
        eidx[e] = (uint32_t)e;
        erank[e] = (uint8_t)p;
    }
}

// sorted position i (by cell, contributors of a cell in rank order): head position of its cell's run, neighbouring contributors,
// and the key the rows are sorted by: the first contributor's first-touch key, every other entry a sentinel above all keys
__global__ __launch_bounds__(256) void m2_segments_kernel(long long E, M2Offsets o, long long stride, const long long* __restrict__ g,
                                                          const uint32_t* __restrict__ scell, const uint32_t* __restrict__ se,
                                                          const uint8_t* __restrict__ erank, unsigned long long sentinel,
                                                          uint32_t* __restrict__ hp, uint16_t* __restrict__ pn,
                                                          unsigned long long* __restrict__ k2, uint32_t* __restrict__ v2,
                                                          unsigned long long* __restrict__ res) {
    unsigned heads = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < E; i += (long long)gridDim.x * blockDim.x) {
        const uint32_t c = scell[i];
        const bool head = i == 0 || scell[i - 1] != c;
        const bool tail = i + 1 == E || scell[i + 1] != c;
        long long h = i;
        while (h > 0 && scell[h - 1] == c) --h;                         // a cell has at most ws contributors
        hp[i] = (uint32_t)h;
        const int prev = head ? -1 : (int)erank[se[i - 1]];
        const int next = tail ? -1 : (int)erank[se[i + 1]];
        pn[i] = (uint16_t)((prev + 1) | ((next + 1) << 8));
        const uint32_t e = se[i];
        const int p = erank[e];
        k2[i] = head ? (unsigned long long)g[p * stride + (e - o.off[p])] : sentinel;
        v2[i] = (uint32_t)i;
        heads += head ? 1u : 0u;
    }
    for (int off = 32; off > 0; off >>= 1) heads += __shfl_xor(heads, off, 64);
    if ((threadIdx.x & 63) == 0 && heads) atomicAdd(&res[0], (unsigned long long)heads);      // res[0] = M
}

// row j of the merged map: the j-th smallest first-touch key
__global__ void m2_rows_kernel(long long E, const unsigned long long* __restrict__ k2s, const uint32_t* __restrict__ v2s,
                               unsigned long long sentinel, const uint32_t* __restrict__ scell, long long grow_row,
                               uint32_t* __restrict__ rowofhead, int32_t* __restrict__ rowcell, unsigned long long* __restrict__ res) {
    for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < E; j += (long long)gridDim.x * blockDim.x) {
        const unsigned long long k = k2s[j];
        if (k >= sentinel) continue;
        const uint32_t i = v2s[j];
        rowofhead[i] = (uint32_t)j;
        rowcell[j] = (int32_t)scell[i];
        if (j == grow_row) res[1] = k;
    }
}

// every entry: its final row and destination; the ws x ws tables [sender][owner] of list sizes; this rank's own voxels by slot
__global__ __launch_bounds__(256) void m2_entries_kernel(long long E, int ws, int rank, M2Offsets o, const uint32_t* __restrict__ se,
                                                         const uint8_t* __restrict__ erank, const uint32_t* __restrict__ hp,
                                                         const uint16_t* __restrict__ pn, const uint32_t* __restrict__ rowofhead,
                                                         unsigned long long* __restrict__ res, int32_t* __restrict__ row_s,
                                                         int32_t* __restrict__ prev_s, int32_t* __restrict__ next_s,
                                                         uint32_t* __restrict__ krow, uint32_t* __restrict__ vslot) {
    extern __shared__ unsigned m2_hist[];                 // [all | done | hop] x ws x ws
    const int W2 = ws * ws;
    for (int t = threadIdx.x; t < 3 * W2; t += blockDim.x) m2_hist[t] = 0;
    __syncthreads();
    const long long M = (long long)res[0];
    const long long per = M > 0 ? (M + ws - 1) / ws : 1;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < E; i += (long long)gridDim.x * blockDim.x) {
        const uint32_t e = se[i];
        const int p = erank[e];
        const long long row = rowofhead[hp[i]];
        int q = (int)(row / per);
        q = q < ws - 1 ? q : ws - 1;
        const int prev = (int)(pn[i] & 0xFF) - 1, next = (int)(pn[i] >> 8) - 1;
        atomicAdd(&m2_hist[p * ws + q], 1u);
        if (prev < 0 && next < 0) atomicAdd(&m2_hist[W2 + p * ws + q], 1u);
        if (prev >= 0) atomicAdd(&m2_hist[2 * W2 + prev * ws + p], 1u);
        if (p == rank) {
            const long long s = e - o.off[p];
            row_s[s] = (int32_t)row;
            prev_s[s] = prev;
            next_s[s] = next;
            krow[s] = (uint32_t)row;
            vslot[s] = (uint32_t)s;
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 3 * W2; t += blockDim.x)
        if (m2_hist[t]) atomicAdd(&res[2 + t], (unsigned long long)m2_hist[t]);
}

// own voxels in final-row order: the flags of the mixed payload, the replay selections, the keys of the hop lists
__global__ void m2_own_kernel(long long n, int ws, const int32_t* __restrict__ order, const int32_t* __restrict__ prev_s,
                              const int32_t* __restrict__ next_s, uint8_t* __restrict__ single, long long* __restrict__ selA,
                              long long* __restrict__ selB, uint32_t* __restrict__ kp, uint32_t* __restrict__ kn) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int32_t s = order[i];
        const int prev = prev_s[s], next = next_s[s];
        single[i] = (prev < 0 && next < 0) ? 1 : 0;
        selA[s] = prev < 0 ? (long long)s : -1ll;
        selB[s] = prev < 0 ? -1ll : (long long)s;
        kp[i] = (uint32_t)(prev < 0 ? ws : prev);
        kn[i] = (uint32_t)(next < 0 ? ws : next);
    }
}

struct FlagToI32 {
    __device__ int32_t operator()(uint8_t f) const { return f ? 1 : 0; }
};

// the replay state of the selected voxels, in list order (24 B = 3 words each)
__global__ void m2_state_gather_kernel(long long k, const int32_t* __restrict__ idx, const long long* __restrict__ state,
                                       long long* __restrict__ out) {
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < 3 * k; t += (long long)gridDim.x * blockDim.x) {
        const long long i = t / 3;
        out[t] = state[3ll * idx[i] + (t - 3 * i)];
    }
}

__global__ void m2_state_scatter_kernel(long long k, const int32_t* __restrict__ idx, const long long* __restrict__ in,
                                        long long* __restrict__ state) {
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < 3 * k; t += (long long)gridDim.x * blockDim.x) {
        const long long i = t / 3;
        state[3ll * idx[i] + (t - 3 * i)] = in[t];
    }
}

struct M2Seg {
    long long start[kM2MaxRanks + 1];   // own voxels [start[q], start[q + 1]) of the final-row order go to rank q
    long long side_off[kM2MaxRanks];    // word offset of destination q's side records in the send buffer
};

// words 5..7 of every side record: the replay state where this rank is the voxel's LAST contributor, zeros elsewhere
__global__ void m2_side_state_kernel(long long n, int ws, M2Seg sg, const int32_t* __restrict__ order, const int32_t* __restrict__ next_s,
                                     const long long* __restrict__ state, long long* __restrict__ send) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        int q = 0;
        while (q + 1 < ws && i >= sg.start[q + 1]) ++q;
        const int32_t s = order[i];
        long long* rec = send + sg.side_off[q] + 8 * (i - sg.start[q]);
        const bool last = state && next_s[s] < 0;
        rec[5] = last ? state[3ll * s] : 0ll;
        rec[6] = last ? state[3ll * s + 1] : 0ll;
        rec[7] = last ? state[3ll * s + 2] : 0ll;
    }
}

struct M2Peers {
    const long long* side[kM2MaxRanks];   // peer p's side records for this rank (64 B each, rows ascending)
    const float* done[kM2MaxRanks];       // ... its finished float32 rows
    const double* part[kM2MaxRanks];      // ... its float64 partial rows
    long long count[kM2MaxRanks];
};

struct ReplayState24 {
    double w;
    float c[3];
    uint32_t started;
};

// Owner side.  Wave per row r of the block: lane p looks r up in peer p's side list (sorted by row); the contributors are summed
// in rank order -- [sum alpha, sum alpha rgb] and, for a voxel several ranks touched, the float64 feature partials -- and the row
// is finished with finalize_kernel's expressions (avl_builder.hip).  A voxel with one contributor arrives as a finished float32
// row (or is in place already: kM2Direct).
__global__ __launch_bounds__(256) void m2_fold_kernel(long long n_own, long long r0, int ws, int D, long long ldf, long long ldp, int gs, int vh,
                                                      M2Peers pe, const int32_t* __restrict__ rowcell, int have_log,
                                                      float* __restrict__ grid_feat, int32_t* __restrict__ grid_pos,
                                                      float* __restrict__ weight, uint8_t* __restrict__ grid_rgb,
                                                      int32_t* __restrict__ cell_out, int* __restrict__ err) {
    const int lane = threadIdx.x & 63;
    const long long wave0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long r = wave0; r < n_own; r += nwaves) {
        long long found = -1;
        if (lane < ws) {
            const long long* sd = pe.side[lane];
            long long lo = 0, hi = pe.count[lane];
            while (lo < hi) {
                const long long mid = (lo + hi) >> 1;
                const long long v = sd[8 * mid] & 0xFFFFFFFFll;
                if (v < r) lo = mid + 1;
                else hi = mid;
            }
            if (lo < pe.count[lane] && (sd[8 * lo] & 0xFFFFFFFFll) == r) found = lo;
        }
        unsigned long long mask = __ballot(found >= 0);
        if (mask == 0) {
            if (lane == 0 && err) atomicOr(err, 2);          // a row of the block nobody sent: a plan / exchange bug
            continue;
        }
        const int ncontrib = __popcll(mask);
        double w4[4] = {0.0, 0.0, 0.0, 0.0};
        ReplayState24 st{0.0, {0.f, 0.f, 0.f}, 0u};
        unsigned long long word0 = 0;
        int p0 = -1;
        for (unsigned long long m = mask; m; m &= m - 1) {
            const int p = __ffsll((long long)m) - 1;
            const long long idx = __shfl(found, p, 64);
            const long long* rec = pe.side[p] + 8 * idx;
            if (p0 < 0) { p0 = p; word0 = (unsigned long long)rec[0]; }
            for (int k = 0; k < 4; ++k) w4[k] += __longlong_as_double(rec[1 + k]);
            const unsigned long long s2 = (unsigned long long)rec[7];
            if ((s2 >> 32) != 0) {
                st.w = __longlong_as_double(rec[5]);
                const unsigned long long s1 = (unsigned long long)rec[6];
                st.c[0] = __uint_as_float((unsigned)(s1 & 0xFFFFFFFFull));
                st.c[1] = __uint_as_float((unsigned)(s1 >> 32));
                st.c[2] = __uint_as_float((unsigned)(s2 & 0xFFFFFFFFull));
                st.started = (uint32_t)(s2 >> 32);
            }
        }
        const double w = w4[0];
        float* o = grid_feat + r * D;
        if (ncontrib == 1 && (word0 & kM2Single)) {
            if (!(word0 & kM2Direct)) {
                const float* src = pe.done[p0] + (long long)((word0 >> 32) & 0x3FFFFFFFull) * ldf;
                for (int c = lane; c < D; c += 64) o[c] = src[c];
            }
        } else {
            if (lane == 0 && err && (word0 & kM2Single)) atomicOr(err, 4);
            for (int c0 = 0; c0 < D; c0 += 64) {          // (whole-wave trips: the shuffles below need every lane)
                const int c = c0 + lane;
                double acc = 0.0;
                for (unsigned long long m = mask; m; m &= m - 1) {
                    const int p = __ffsll((long long)m) - 1;
                    const long long idx = __shfl(found, p, 64);
                    const unsigned long long wd = (unsigned long long)pe.side[p][8 * idx];
                    if (c < D) acc += pe.part[p][(long long)((wd >> 32) & 0x3FFFFFFFull) * ldp + c];
                }
                if (c < D) o[c] = (float)(acc / w);
            }
        }
        if (lane == 0) {
            const int32_t cl = rowcell[r0 + r];
            if (cell_out) cell_out[r] = cl;
            grid_pos[r * 3 + 0] = cl / (gs * vh);
            grid_pos[r * 3 + 1] = (cl / vh) % gs;
            grid_pos[r * 3 + 2] = cl % vh;
            float wt = (float)w;
            uint8_t c3[3];
            for (int k = 0; k < 3; ++k) {
                double m = w4[1 + k] / w + 1e-9;          // as finalize_kernel
                m = fmin(fmax(m, 0.0), 255.0);
                c3[k] = (uint8_t)m;
            }
            if (have_log && st.started) {               // replay_apply_kernel
                wt = (float)st.w;
                for (int k = 0; k < 3; ++k) c3[k] = (uint8_t)fminf(fmaxf(st.c[k], 0.f), 255.f);
            }
            weight[r] = wt;
            for (int k = 0; k < 3; ++k) grid_rgb[r * 3 + k] = c3[k];
        }
    }
}

static size_t m2_al(size_t b) { return (b + 255) / 256 * 256; }
static unsigned m2_grid(long long n) { return (unsigned)std::min<long long>((n + 255) / 256, 4096); }
static int bit_length(unsigned long long v) {
    int b = 0;
    while (v) { ++b; v >>= 1; }
    return b;
}

// layout of the work buffer of avl_merge2_plan (byte offsets from the 256-aligned base)
struct M2Layout {
    size_t row, prev, next, order, sidx, selA, selB, idx_prev, idx_next, rowcell, res;   // results (see avl_merge2_plan)
    size_t ecell, eidx, erank, scell, se, hp, pn, k2, v2, k2s, v2s, rowofhead, krow, vslot, krow_s, single, kp, kn, kps, tmp, total;
    size_t tmp_bytes;
};

static int m2_layout(long long E, long long n, int ws, M2Layout& L) {
    const size_t e = (size_t)(E > 0 ? E : 1), m = (size_t)(n > 0 ? n : 1);
    size_t t_cell = 0, t_key = 0, t_row = 0, t_small = 0, t_scan = 0;
    AVL_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, t_cell, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, e, 0, 32, nullptr));
    AVL_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, t_key, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (uint32_t*)nullptr,
                                            (uint32_t*)nullptr, e, 0, 64, nullptr));
    AVL_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, t_row, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, m, 0, 32, nullptr));
    AVL_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, t_small, (uint32_t*)nullptr, (uint32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr, m, 0, 8, nullptr));
    {
        auto it = rocprim::make_transform_iterator((const uint8_t*)nullptr, FlagToI32{});
        AVL_HIP_CHECK(rocprim::exclusive_scan(nullptr, t_scan, it, (int32_t*)nullptr, 0, m, rocprim::plus<int32_t>(), nullptr));
    }
    size_t p = 0;
    auto take = [&](size_t bytes) { const size_t at = p; p += m2_al(bytes); return at; };
    L.row = take(m * 4); L.prev = take(m * 4); L.next = take(m * 4); L.order = take(m * 4); L.sidx = take(m * 4);
    L.selA = take(m * 8); L.selB = take(m * 8); L.idx_prev = take(m * 4); L.idx_next = take(m * 4);
    L.rowcell = take(e * 4);
    L.res = take((size_t)(2 + 3 * ws * ws) * 8);
    L.ecell = take(e * 4); L.eidx = take(e * 4); L.erank = take(e); L.scell = take(e * 4); L.se = take(e * 4); L.hp = take(e * 4);
    L.pn = take(e * 2); L.k2 = take(e * 8); L.v2 = take(e * 4); L.k2s = take(e * 8); L.v2s = take(e * 4); L.rowofhead = take(e * 4);
    L.krow = take(m * 4); L.vslot = take(m * 4); L.krow_s = take(m * 4); L.single = take(m); L.kp = take(m * 4); L.kn = take(m * 4);
    L.kps = take(m * 4);
    L.tmp_bytes = std::max(std::max(t_cell, t_key), std::max(std::max(t_row, t_small), t_scan));
    L.tmp = take(L.tmp_bytes ? L.tmp_bytes : 16);
    L.total = p + 256;
    return AVL_OK;
}

}  // namespace avl

using namespace avl;

extern "C" {

int avl_merge2_header(int64_t n, const int64_t* d_key, int64_t flags, int64_t* d_hdr, void* stream) {
    AVL_REQUIRE(n >= 0 && d_hdr && (n == 0 || d_key), "avl_merge2_header: bad arguments");
    hipLaunchKernelGGL(m2_header_kernel, dim3(1), dim3(1024), 0, as_stream(stream), (long long)n, reinterpret_cast<const long long*>(d_key),
                       (long long)flags, reinterpret_cast<long long*>(d_hdr));
    AVL_HIP_CHECK(hipGetLastError());
    return AVL_OK;
}

int avl_merge2_work_bytes(int64_t E, int64_t n, int ws, size_t* h_bytes) {
    AVL_REQUIRE(h_bytes && E >= 0 && E < (1ll << 31) && n >= 0 && n <= E && ws >= 1 && ws <= kM2MaxRanks,
                "avl_merge2_work_bytes: bad arguments (at most %d ranks, 2^31 entries)", kM2MaxRanks);
    M2Layout L;
    int rc = m2_layout(E, n, ws, L);
    if (rc != AVL_OK) return rc;
    *h_bytes = L.total;
    return AVL_OK;
}

int avl_merge2_plan(int ws, int rank, const int64_t* h_n_all, int64_t nmax, const int64_t* d_gathered, int cell_bits, int key_bits,
                    int64_t grow_row, int want_replay_lists, void* d_work, size_t work_bytes, int64_t* h_off, int64_t* h_res, void* stream) {
    AVL_REQUIRE(ws >= 1 && ws <= kM2MaxRanks && rank >= 0 && rank < ws && h_n_all && h_off && h_res, "avl_merge2_plan: bad arguments");
    AVL_REQUIRE(cell_bits >= 1 && cell_bits <= 31 && key_bits >= 1 && key_bits <= 63, "avl_merge2_plan: cell_bits in [1, 31], key_bits in [1, 63]");
    M2Offsets o;
    long long E = 0;
    for (int p = 0; p < ws; ++p) {
        AVL_REQUIRE(h_n_all[p] >= 0 && h_n_all[p] <= nmax, "avl_merge2_plan: rank %d holds %lld voxels, chunk size %lld", p, (long long)h_n_all[p],
                    (long long)nmax);
        o.off[p] = E;
        E += h_n_all[p];
    }
    for (int p = ws; p <= kM2MaxRanks; ++p) o.off[p] = E;
    AVL_REQUIRE(E < (1ll << 31), "avl_merge2_plan: more than 2^31 entries");
    const long long n = h_n_all[rank];
    const long long stride = nmax + (nmax + 1) / 2;
    M2Layout L;
    int rc = m2_layout(E, n, ws, L);
    if (rc != AVL_OK) return rc;
    AVL_REQUIRE(d_work && work_bytes >= L.total, "avl_merge2_plan: work buffer of %zu bytes, %zu needed (avl_merge2_work_bytes)", work_bytes, L.total);
    char* base = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(d_work) + 255) / 256 * 256);
    const int64_t shift = base - reinterpret_cast<char*>(d_work);
    const size_t offs[11] = {L.row, L.prev, L.next, L.order, L.sidx, L.selA, L.selB, L.idx_prev, L.idx_next, L.rowcell, L.res};
    for (int k = 0; k < 11; ++k) h_off[k] = (int64_t)offs[k] + shift;
    hipStream_t st = as_stream(stream);
    const int nres = 2 + 3 * ws * ws;
    unsigned long long* res = reinterpret_cast<unsigned long long*>(base + L.res);
    AVL_HIP_CHECK(hipMemsetAsync(res, 0, (size_t)nres * 8, st));
    const unsigned long long sentinel = 1ull << key_bits;
    if (E > 0) {
        AVL_REQUIRE(d_gathered, "avl_merge2_plan: null gathered lists");
        auto U32 = [&](size_t off) { return reinterpret_cast<uint32_t*>(base + off); };
        auto I32 = [&](size_t off) { return reinterpret_cast<int32_t*>(base + off); };
        auto U64 = [&](size_t off) { return reinterpret_cast<unsigned long long*>(base + off); };
        const long long* g = reinterpret_cast<const long long*>(d_gathered);
        uint8_t* erank = reinterpret_cast<uint8_t*>(base + L.erank);
        hipLaunchKernelGGL(m2_compact_kernel, dim3(m2_grid(E)), dim3(256), 0, st, E, ws, o, stride, (long long)nmax, g, U32(L.ecell), U32(L.eidx), erank);
        size_t tb = L.tmp_bytes;
        AVL_HIP_CHECK(rocprim::radix_sort_pairs(base + L.tmp, tb, U32(L.ecell), U32(L.scell), U32(L.eidx), U32(L.se), (size_t)E, 0, cell_bits, st));
        hipLaunchKernelGGL(m2_segments_kernel, dim3(m2_grid(E)), dim3(256), 0, st, E, o, stride, g, U32(L.scell), U32(L.se), erank, sentinel, U32(L.hp),
                           reinterpret_cast<uint16_t*>(base + L.pn), U64(L.k2), U32(L.v2), res);
        tb = L.tmp_bytes;
        AVL_HIP_CHECK(rocprim::radix_sort_pairs(base + L.tmp, tb, U64(L.k2), U64(L.k2s), U32(L.v2), U32(L.v2s), (size_t)E, 0, key_bits + 1, st));
        hipLaunchKernelGGL(m2_rows_kernel, dim3(m2_grid(E)), dim3(256), 0, st, E, U64(L.k2s), U32(L.v2s), sentinel, U32(L.scell), (long long)grow_row,
                           U32(L.rowofhead), I32(L.rowcell), res);
        hipLaunchKernelGGL(m2_entries_kernel, dim3(std::min(m2_grid(E), 512u)), dim3(256), (size_t)(3 * ws * ws) * sizeof(unsigned), st, E, ws, rank, o,
                           U32(L.se), erank, U32(L.hp), reinterpret_cast<const uint16_t*>(base + L.pn), U32(L.rowofhead), res, I32(L.row), I32(L.prev),
                           I32(L.next), U32(L.krow), U32(L.vslot));
        if (n > 0) {
            tb = L.tmp_bytes;
            AVL_HIP_CHECK(rocprim::radix_sort_pairs(base + L.tmp, tb, U32(L.krow), U32(L.krow_s), U32(L.vslot), U32(L.order), (size_t)n, 0,
                                                    std::max(1, bit_length((unsigned long long)E)), st));
            uint8_t* single = reinterpret_cast<uint8_t*>(base + L.single);
            hipLaunchKernelGGL(m2_own_kernel, dim3(m2_grid(n)), dim3(256), 0, st, n, ws, I32(L.order), I32(L.prev), I32(L.next), single,
                               reinterpret_cast<long long*>(base + L.selA), reinterpret_cast<long long*>(base + L.selB), U32(L.kp), U32(L.kn));
            tb = L.tmp_bytes;
            auto it = rocprim::make_transform_iterator(reinterpret_cast<const uint8_t*>(single), FlagToI32{});
            AVL_HIP_CHECK(rocprim::exclusive_scan(base + L.tmp, tb, it, I32(L.sidx), 0, (size_t)n, rocprim::plus<int32_t>(), st));
            if (want_replay_lists && ws > 1) {
                const int rb = std::max(1, bit_length((unsigned long long)ws));
                tb = L.tmp_bytes;
                AVL_HIP_CHECK(rocprim::radix_sort_pairs(base + L.tmp, tb, U32(L.kp), U32(L.kps), I32(L.order), I32(L.idx_prev), (size_t)n, 0, rb, st));
                tb = L.tmp_bytes;
                AVL_HIP_CHECK(rocprim::radix_sort_pairs(base + L.tmp, tb, U32(L.kn), U32(L.kps), I32(L.order), I32(L.idx_next), (size_t)n, 0, rb, st));
            }
        }
        AVL_HIP_CHECK(hipGetLastError());
    }
    // the ONE read-back of the merge: M, the growth key and the three ws x ws size tables
    AVL_HIP_CHECK(hipMemcpyAsync(h_res, res, (size_t)nres * 8, hipMemcpyDeviceToHost, st));
    AVL_HIP_CHECK(hipStreamSynchronize(st));
    if ((long long)h_res[0] <= grow_row || grow_row < 0) h_res[1] = -1;       // all ones: no voxel with that id
    return AVL_OK;
}

int avl_merge2_state_gather(int64_t k, const int32_t* d_idx, const int64_t* d_state, int64_t* d_out, void* stream) {
    AVL_REQUIRE(k >= 0, "avl_merge2_state_gather: bad k");
    if (k == 0) return AVL_OK;
    AVL_REQUIRE(d_idx && d_state && d_out, "avl_merge2_state_gather: null pointer");
    hipLaunchKernelGGL(m2_state_gather_kernel, dim3(m2_grid(3 * k)), dim3(256), 0, as_stream(stream), (long long)k, d_idx,
                       reinterpret_cast<const long long*>(d_state), reinterpret_cast<long long*>(d_out));
    AVL_HIP_CHECK(hipGetLastError());
    return AVL_OK;
}

int avl_merge2_state_scatter(int64_t k, const int32_t* d_idx, const int64_t* d_in, int64_t* d_state, void* stream) {
    AVL_REQUIRE(k >= 0, "avl_merge2_state_scatter: bad k");
    if (k == 0) return AVL_OK;
    AVL_REQUIRE(d_idx && d_in && d_state, "avl_merge2_state_scatter: null pointer");
    hipLaunchKernelGGL(m2_state_scatter_kernel, dim3(m2_grid(3 * k)), dim3(256), 0, as_stream(stream), (long long)k, d_idx,
                       reinterpret_cast<const long long*>(d_in), reinterpret_cast<long long*>(d_state));
    AVL_HIP_CHECK(hipGetLastError());
    return AVL_OK;
}

int avl_merge2_side_state(int64_t n, int ws, const int64_t* h_start, const int64_t* h_side_off, const int32_t* d_order, const int32_t* d_next,
                          const int64_t* d_state, int64_t* d_send, void* stream) {
    AVL_REQUIRE(n >= 0 && ws >= 1 && ws <= kM2MaxRanks && h_start && h_side_off, "avl_merge2_side_state: bad arguments");
    if (n == 0) return AVL_OK;
    AVL_REQUIRE(d_order && d_next && d_send, "avl_merge2_side_state: null pointer");
    M2Seg sg;
    for (int q = 0; q <= kM2MaxRanks; ++q) sg.start[q] = h_start[q < ws ? q : ws];
    for (int q = 0; q < kM2MaxRanks; ++q) sg.side_off[q] = q < ws ? h_side_off[q] : 0;
    hipLaunchKernelGGL(m2_side_state_kernel, dim3(m2_grid(n)), dim3(256), 0, as_stream(stream), (long long)n, ws, sg, d_order, d_next,
                       reinterpret_cast<const long long*>(d_state), reinterpret_cast<long long*>(d_send));
    AVL_HIP_CHECK(hipGetLastError());
    return AVL_OK;
}

int avl_merge2_fold(int64_t n_own, int64_t r0, int ws, int D, int gs, int vh, const void* const* h_side, const void* const* h_done,
                    const void* const* h_part, const int64_t* h_count, const int32_t* d_rowcell, int have_log, float* d_grid_feat,
                    int32_t* d_grid_pos, float* d_weight, uint8_t* d_grid_rgb, int32_t* d_cell, int32_t* d_err_flag, void* stream) {
    AVL_REQUIRE(n_own >= 0 && r0 >= 0 && ws >= 1 && ws <= kM2MaxRanks && D > 0 && gs > 0 && vh > 0, "avl_merge2_fold: bad shape");
    if (n_own == 0) return AVL_OK;
    AVL_REQUIRE(h_side && h_done && h_part && h_count && d_rowcell && d_grid_feat && d_grid_pos && d_weight && d_grid_rgb, "avl_merge2_fold: null pointer");
    M2Peers pe{};
    for (int p = 0; p < ws; ++p) {
        pe.side[p] = reinterpret_cast<const long long*>(h_side[p]);
        pe.done[p] = reinterpret_cast<const float*>(h_done[p]);
        pe.part[p] = reinterpret_cast<const double*>(h_part[p]);
        pe.count[p] = h_count[p];
        AVL_REQUIRE(h_count[p] == 0 || h_side[p], "avl_merge2_fold: peer %d has records but no buffer", p);
    }
    const long long ldf = (D + 1) / 2 * 2;      // float32 rows are padded to whole 8-byte words
    int64_t blocks = (n_own + 3) / 4;
    const int64_t maxb = (int64_t)num_cus() * 16;
    if (blocks > maxb) blocks = maxb;
    hipLaunchKernelGGL(m2_fold_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), (long long)n_own, (long long)r0, ws, D, ldf, (long long)D,
                       gs, vh, pe, d_rowcell, have_log, d_grid_feat, d_grid_pos, d_weight, d_grid_rgb, d_cell, reinterpret_cast<int*>(d_err_flag));
    AVL_HIP_CHECK(hipGetLastError());
    return AVL_OK;
}

}  // extern "C"
