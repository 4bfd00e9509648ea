// The multi-GPU merge of the map build, second form ("gather plan"): everything a rank computes between two collectives is ONE
// entry point of this file (or avl_builder_m2_pack in avl_builder.hip, which needs the accumulators), with no host
// synchronisation inside except the single read-back of the plan's sizes.
//
// What is merged: the ranks' voxel maps of the reference's builder loop (avlmaps/map/vlmap_builder.py:102-183), frames sharded
// contiguously over the ranks.  A voxel's final row is the reference's voxel id = its position in first-touch order
// (vlmap_builder.py:163-170).  Protocol (avlmaps_amd/merge2.py carries the collectives through torch.distributed):
//   0. avl_merge2_prepare: the rank's own (first-touch key, cell) list sorted by key (n entries); the header comes out of it
//   1. all_gather of a 4-word header per rank (voxel count, key range, flags)
//   2. all_gather of every rank's key-sorted (first-touch key, cell) list, 12 B per voxel    -> avl_merge2_plan on EVERY rank:
//        the union of the cells (ONE radix sort of all sum(n) entries by cell; stable, so a cell's contributors stay in rank
//        order), the first contributor of every cell; the keys are ordered by rank and every list is in key order, so ENTRY
//        order is key order and the reference's row of a cell is the number of first contributors before its own (a scan, no
//        second sort); for the rank's OWN voxels: final row, neighbouring contributors (prev / next rank), the send order, the
//        replay lists; the ws x ws tables of list sizes every later exchange needs are read back ONCE (the only host
//        synchronisation)
//   3. the sequential weight / colour replay hops rank -> next rank for voxels several ranks touched (24 B of state each)
//   4. ONE all_to_all of the payload: per destination [side records 64 B | finished float32 rows | float64 partial rows]
//   5. avl_merge2_fold: a scatter builds the table "which record of peer p belongs to row r" (the side lists carry their rows);
//      thread per row: the contributors' sums in rank order (reproducible float64), position / weight / colour from them or the
//      replay state; wave per row that still needs features: copy of the finished float32 row, or (sum of the float64 partials in
//      rank order) / sum alpha.
// The first plan (avl_merge.hip + parallel.plan_merge_directory: directory ranks, five small round trips, nothing O(sum n) per
// rank) stays selectable (AVLMAPS_MERGE_PLAN=directory); here every rank sorts all sum(n) entries by cell (2.3 M entries at 8 ranks
// x 290 k voxels: ~0.15 ms) and the round trips are gone: 3.7-4.1 ms of compute per rank against 7.9-12.3 (profiles/r06_*).
#include <algorithm>
#include <cstdint>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include "avl_common.h"

// rocPRIM's radix sort falls back to a merge sort below 2^20 items: ~18 block-merge launches of 5 us each for a rank's 290 k voxels, four
// times per merge.  Onesweep from 4 096 items on (a histogram + one pass per 8 key bits): fewer, fuller launches
// (profiles/r06_merge2_8ranks_one_process_kernel_stats.csv has the merge-sort form).
#ifndef AVL_M2_MERGE_SORT_LIMIT
#define AVL_M2_MERGE_SORT_LIMIT 4096
#endif
using M2SortCfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, AVL_M2_MERGE_SORT_LIMIT>;

namespace avl {

constexpr int kM2MaxRanks = 64;
constexpr unsigned long long kM2Single = 1ull << 63;   // side word: the voxel has ONE contributor (its feature row is finished float32)
constexpr unsigned long long kM2Direct = 1ull << 62;   // ... and that contributor owns the row: the row is already in the block

struct M2Offsets {
    long long off[kM2MaxRanks + 1];   // entry index of rank p's first voxel; off[ws] = E
};

// a rank's (first-touch key, cell) list sorted by key goes out: iota for the sort's values, then the cells in that order + the header
__global__ void m2_iota_kernel(long long n, int32_t* __restrict__ v) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) v[i] = (int32_t)i;
}

__global__ void m2_presorted_kernel(long long n, const int32_t* __restrict__ cell, const int32_t* __restrict__ perm, const long long* __restrict__ key_sorted,
                                    long long flags, int32_t* __restrict__ cell_sorted, long long* __restrict__ hdr) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) cell_sorted[i] = cell[perm[i]];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        hdr[0] = n;
        hdr[1] = n ? key_sorted[0] : 0x7FFFFFFFFFFFFFFFll;
        hdr[2] = n ? key_sorted[n - 1] : -1;
        hdr[3] = flags;
    }
}

// entry e = (rank p, position j in p's key-sorted list): its cell (sort key), its index (sort value), its rank
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
// and the flag "first contributor of its cell" back in entry order.  The ranks' lists are sorted by key and the keys ordered by rank,
// so ENTRY order is key order: a first contributor's row is the number of first contributors before it (one scan, no second sort).
__global__ __launch_bounds__(256) void m2_segments_kernel(long long E, const uint32_t* __restrict__ scell, const uint32_t* __restrict__ se,
                                                          const uint8_t* __restrict__ erank, uint32_t* __restrict__ hp, uint16_t* __restrict__ pn,
                                                          uint8_t* __restrict__ headflag, unsigned long long* __restrict__ res) {
    __shared__ unsigned sh_heads[4];
    unsigned heads = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < E; i += (long long)gridDim.x * blockDim.x) {
        const uint32_t c = scell[i];
        const bool head = i == 0 || scell[i - 1] != c;
        const bool tail = i + 1 == E || scell[i + 1] != c;
        long long h = i;
        if (!head) {
            --h;
            while (h > 0 && scell[h - 1] == c) --h;                     // a cell has at most ws contributors
        }
        hp[i] = (uint32_t)h;
        const int prev = head ? -1 : (int)erank[se[i - 1]];
        const int next = tail ? -1 : (int)erank[se[i + 1]];
        pn[i] = (uint16_t)((prev + 1) | ((next + 1) << 8));
        headflag[se[i]] = head ? 1 : 0;
        heads += head ? 1u : 0u;
    }
    for (int off = 32; off > 0; off >>= 1) heads += __shfl_xor(heads, off, 64);
    if ((threadIdx.x & 63) == 0) sh_heads[threadIdx.x >> 6] = heads;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = sh_heads[0] + sh_heads[1] + sh_heads[2] + sh_heads[3];
        if (t) atomicAdd(&res[0], (unsigned long long)t);                 // res[0] = M
    }
}

// every entry: its final row and destination; the ws x ws tables [sender][owner] of list sizes; the cell of every row; this rank's
// own voxels by slot (perm: position in the rank's key-sorted list -> slot)
__global__ __launch_bounds__(256) void m2_entries_kernel(long long E, int ws, int rank, M2Offsets o, long long stride, const long long* __restrict__ g,
                                                         const uint32_t* __restrict__ scell, const uint32_t* __restrict__ se,
                                                         const uint8_t* __restrict__ erank, const uint32_t* __restrict__ hp,
                                                         const uint16_t* __restrict__ pn, const int32_t* __restrict__ rowscan,
                                                         const int32_t* __restrict__ perm, long long grow_row, long long chunk_rows, int nchunk,
                                                         unsigned long long* __restrict__ res,
                                                         int32_t* __restrict__ rowcell, int32_t* __restrict__ row_s, int32_t* __restrict__ prev_s,
                                                         int32_t* __restrict__ next_s, uint32_t* __restrict__ krow, uint32_t* __restrict__ vslot) {
    extern __shared__ unsigned m2_hist[];                 // [all | done | hop] x ws x ws, then per chunk of a block's rows [all | done] x ws x ws
    const int W2 = ws * ws;
    const int nbins = (3 + 2 * nchunk) * W2;
    for (int t = threadIdx.x; t < nbins; t += blockDim.x) m2_hist[t] = 0;
    __syncthreads();
    const long long M = (long long)res[0];
    const long long per = M > 0 ? (M + ws - 1) / ws : 1;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < E; i += (long long)gridDim.x * blockDim.x) {
        const uint32_t e = se[i];
        const int p = erank[e];
        const uint32_t h = hp[i];
        const long long row = rowscan[se[h]];
        if (h == (uint32_t)i) {
            rowcell[row] = (int32_t)scell[i];
            if (row == grow_row) res[1] = (unsigned long long)g[p * stride + (e - o.off[p])];
        }
        int q = (int)(row / per);
        q = q < ws - 1 ? q : ws - 1;
        const int prev = (int)(pn[i] & 0xFF) - 1, next = (int)(pn[i] >> 8) - 1;
        // (one rank, or a stretch of a single rank's voxels: the whole wave counts into the same word -- one lane adds the wave's total
        // instead of 64 serialised LDS atomics; m2_entries_kernel 210 -> 60 us at 2.25 M entries of one rank)
        const int bin = p * ws + q;
        const bool single = prev < 0 && next < 0;
        const unsigned long long act = __ballot(1);
        const int bin0 = __builtin_amdgcn_readfirstlane(bin);
        if (__ballot(bin != bin0) == 0) {
            const unsigned long long sm = __ballot(single);
            if ((threadIdx.x & 63) == __ffsll((long long)act) - 1) {
                atomicAdd(&m2_hist[bin0], (unsigned)__popcll(act));
                if (sm) atomicAdd(&m2_hist[W2 + bin0], (unsigned)__popcll(sm));
            }
        } else {
            atomicAdd(&m2_hist[bin], 1u);
            if (single) atomicAdd(&m2_hist[W2 + bin], 1u);
        }
        if (prev >= 0) atomicAdd(&m2_hist[2 * W2 + prev * ws + p], 1u);
        if (nchunk > 0) {      // the exchange in chunks of chunk_rows rows of every owner's block: the same two tables per chunk
            const int c = (int)((row - (long long)q * per) / chunk_rows);
            const int cb = (3 + 2 * (c < nchunk ? c : nchunk - 1)) * W2 + bin;
            atomicAdd(&m2_hist[cb], 1u);
            if (single) atomicAdd(&m2_hist[cb + W2], 1u);
        }
        if (p == rank) {
            const long long j = e - o.off[p];
            const int32_t s = perm[j];
            row_s[s] = (int32_t)row;
            prev_s[s] = prev;
            next_s[s] = next;
            krow[j] = (uint32_t)row;
            vslot[j] = (uint32_t)s;
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < nbins; t += blockDim.x)
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
    long long cum[kM2MaxRanks + 1];     // as M2PackSeg (avl_builder.hip): record t serves rank q with cum[q] <= t < cum[q + 1]
    long long lo[kM2MaxRanks];          // ... and is voxel lo[q] + (t - cum[q]) of the rank's final-row order
    long long side_off[kM2MaxRanks];    // word offset of destination q's side records in the send buffer
};

// words 5..7 of every side record: the replay state where this rank is the voxel's LAST contributor, zeros elsewhere
__global__ void m2_side_state_kernel(long long n, int ws, M2Seg sg, const int32_t* __restrict__ order, const int32_t* __restrict__ next_s,
                                     const long long* __restrict__ state, long long* __restrict__ send) {
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
        int q = 0;
        while (q + 1 < ws && t >= sg.cum[q + 1]) ++q;
        const long long j = t - sg.cum[q];
        const int32_t s = order[sg.lo[q] + j];
        long long* rec = send + sg.side_off[q] + 8 * j;
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

// Owner side, step 1: record i of peer p belongs to row (word & 0xFFFFFFFF) of the block: table[row * ws + p] = i (the table starts at -1).
// A row outside the block sets bit 1 of *err.
__global__ void m2_fold_index_kernel(long long R, int ws, M2Peers pe, long long n_own, int32_t* __restrict__ table, int* __restrict__ err) {
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < R; t += (long long)gridDim.x * blockDim.x) {
        int p = 0;
        long long i = t;
        while (p + 1 < ws && i >= pe.count[p]) { i -= pe.count[p]; ++p; }
        const long long row = pe.side[p][8 * i] & 0xFFFFFFFFll;
        if (row >= n_own) {
            if (err) atomicOr(err, 1);
            continue;
        }
        table[row * ws + p] = (int32_t)i;
    }
}

// Owner side, step 2.  Thread per row r of the block: the contributors' [sum alpha, sum alpha rgb] summed in rank order, the replay
// state of the last one, position / weight / colour written (finalize_kernel's lane-0 part); need[r] = the row's features still
// have to be produced (anything but a single-rank voxel that is in place already).
__global__ __launch_bounds__(256) void m2_fold_scalar_kernel(long long n_own, long long r0, int ws, int gs, int vh, M2Peers pe,
                                                             const int32_t* __restrict__ table, const int32_t* __restrict__ rowcell, int have_log,
                                                             int32_t* __restrict__ grid_pos, float* __restrict__ weight,
                                                             uint8_t* __restrict__ grid_rgb, int32_t* __restrict__ cell_out,
                                                             double* __restrict__ wsum, uint8_t* __restrict__ need, int* __restrict__ err) {
    for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < n_own; r += (long long)gridDim.x * blockDim.x) {
        double w4[4] = {0.0, 0.0, 0.0, 0.0};
        ReplayState24 st{0.0, {0.f, 0.f, 0.f}, 0u};
        unsigned long long word0 = 0;
        int ncontrib = 0;
        for (int p = 0; p < ws; ++p) {
            const int32_t idx = table[r * ws + p];
            if (idx < 0) continue;
            const long long* rec = pe.side[p] + 8ll * idx;
            if (ncontrib++ == 0) word0 = (unsigned long long)rec[0];
            for (int k = 0; k < 4; ++k) w4[k] += __longlong_as_double(rec[1 + k]);
            const unsigned long long s2 = (unsigned long long)rec[7];
            if ((s2 >> 32) != 0) {
                const unsigned long long s1 = (unsigned long long)rec[6];
                st.w = __longlong_as_double(rec[5]);
                st.c[0] = __uint_as_float((unsigned)(s1 & 0xFFFFFFFFull));
                st.c[1] = __uint_as_float((unsigned)(s1 >> 32));
                st.c[2] = __uint_as_float((unsigned)(s2 & 0xFFFFFFFFull));
                st.started = (uint32_t)(s2 >> 32);
            }
        }
        if (ncontrib == 0) {
            if (err) atomicOr(err, 2);                       // a row of the block nobody sent: a plan / exchange bug
            need[r] = 0;
            continue;
        }
        const bool single = (word0 & kM2Single) != 0;
        if (single && ncontrib > 1 && err) atomicOr(err, 4);
        need[r] = (single && ncontrib == 1 && (word0 & kM2Direct)) ? 0 : 1;
        const double w = w4[0];
        wsum[r] = w;
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

// Owner side, step 3.  Wave per group of 16 rows, the rows that still need their features one after the other: a single-rank voxel's
// finished float32 row is copied, a voxel several ranks touched gets (sum over the contributors, in rank order, of their float64
// partial rows) / sum alpha -- finalize_kernel's expression (avl_builder.hip).
__global__ __launch_bounds__(256) void m2_fold_feat_kernel(long long n_own, int ws, int D, long long ldf, long long ldp, M2Peers pe,
                                                           const int32_t* __restrict__ table, const double* __restrict__ wsum,
                                                           const uint8_t* __restrict__ need, float* __restrict__ grid_feat) {
    const int lane = threadIdx.x & 63;
    const long long wave0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    const long long ngroups = (n_own + 15) / 16;
    for (long long gq = wave0; gq < ngroups; gq += nwaves) {
        const long long rl = gq * 16 + lane;
        unsigned long long todo = __ballot(lane < 16 && rl < n_own && need[rl] != 0);
        for (; todo; todo &= todo - 1) {
            const long long r = gq * 16 + (__ffsll((long long)todo) - 1);
            const int32_t found = lane < ws ? table[r * ws + lane] : -1;
            const unsigned long long mask = __ballot(found >= 0);
            const unsigned long long wd = found >= 0 ? (unsigned long long)pe.side[lane][8ll * found] : 0ull;
            const int p0 = __ffsll((long long)mask) - 1;
            const unsigned long long word0 = (unsigned long long)__shfl((long long)wd, p0, 64);
            float* o = grid_feat + r * D;
            if (__popcll(mask) == 1 && (word0 & kM2Single)) {
                const float* src = pe.done[p0] + (long long)((word0 >> 32) & 0x3FFFFFFFull) * ldf;
                for (int c = lane; c < D; c += 64) o[c] = src[c];
                continue;
            }
            const double w = wsum[r];
            for (int c0 = 0; c0 < D; c0 += 64) {          // (whole-wave trips: the shuffles below need every lane)
                const int c = c0 + lane;
                double acc = 0.0;
                for (unsigned long long m = mask; m; m &= m - 1) {
                    const int p = __ffsll((long long)m) - 1;
                    const unsigned long long wp = (unsigned long long)__shfl((long long)wd, p, 64);
                    if (c < D) acc += pe.part[p][(long long)((wp >> 32) & 0x3FFFFFFFull) * ldp + c];
                }
                if (c < D) o[c] = (float)(acc / w);
            }
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
    size_t ecell, eidx, erank, scell, se, hp, pn, headflag, rowscan, krow, vslot, krow_s, single, kp, kn, kps, tmp, total;
    size_t tmp_bytes;
};

static int m2_layout(long long E, long long n, int ws, int nchunk, M2Layout& L) {
    const size_t e = (size_t)(E > 0 ? E : 1), m = (size_t)(n > 0 ? n : 1);
    size_t t_cell = 0, t_row = 0, t_small = 0, t_scan = 0, t_scan_e = 0;
    AVL_HIP_CHECK(rocprim::radix_sort_pairs<M2SortCfg>(nullptr, t_cell, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, e, 0, 32, nullptr));
    AVL_HIP_CHECK(rocprim::radix_sort_pairs<M2SortCfg>(nullptr, t_row, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, m, 0, 32, nullptr));
    AVL_HIP_CHECK(rocprim::radix_sort_pairs<M2SortCfg>(nullptr, t_small, (uint32_t*)nullptr, (uint32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr, m, 0, 8, nullptr));
    {
        auto it = rocprim::make_transform_iterator((const uint8_t*)nullptr, FlagToI32{});
        AVL_HIP_CHECK(rocprim::exclusive_scan(nullptr, t_scan, it, (int32_t*)nullptr, 0, m, rocprim::plus<int32_t>(), nullptr));
        AVL_HIP_CHECK(rocprim::exclusive_scan(nullptr, t_scan_e, it, (int32_t*)nullptr, 0, e, rocprim::plus<int32_t>(), nullptr));
    }
    size_t p = 0;
    auto take = [&](size_t bytes) { const size_t at = p; p += m2_al(bytes); return at; };
    L.row = take(m * 4); L.prev = take(m * 4); L.next = take(m * 4); L.order = take(m * 4); L.sidx = take(m * 4);
    L.selA = take(m * 8); L.selB = take(m * 8); L.idx_prev = take(m * 4); L.idx_next = take(m * 4);
    L.rowcell = take(e * 4);
    L.res = take((size_t)(2 + (3 + 2 * nchunk) * ws * ws) * 8);
    L.ecell = take(e * 4); L.eidx = take(e * 4); L.erank = take(e); L.scell = take(e * 4); L.se = take(e * 4); L.hp = take(e * 4);
    L.pn = take(e * 2); L.headflag = take(e); L.rowscan = take(e * 4);
    L.krow = take(m * 4); L.vslot = take(m * 4); L.krow_s = take(m * 4); L.single = take(m); L.kp = take(m * 4); L.kn = take(m * 4);
    L.kps = take(m * 4);
    L.tmp_bytes = std::max(std::max(t_cell, t_scan_e), std::max(std::max(t_row, t_small), t_scan));
    L.tmp = take(L.tmp_bytes ? L.tmp_bytes : 16);
    L.total = p + 256;
    return AVL_OK;
}

}  // namespace avl

using namespace avl;

extern "C" {

// HIP loads a code object when one of its kernels is first used (15 ms for this file's: the radix sort and scan instantiations): asking
// for a kernel's attributes does it, once per process
int avl_merge2_load(void) {
    static bool done = false;
    if (done) return AVL_OK;
    hipFuncAttributes attr;
    AVL_HIP_CHECK(hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(m2_compact_kernel)));
    done = true;
    return AVL_OK;
}

int avl_merge2_prepare_work_bytes(int64_t n, size_t* h_bytes) {
    AVL_REQUIRE(h_bytes && n >= 0 && n < (1ll << 31), "avl_merge2_prepare_work_bytes: bad arguments");
    size_t t = 0;
    const size_t m = (size_t)(n > 0 ? n : 1);
    AVL_HIP_CHECK(rocprim::radix_sort_pairs<M2SortCfg>(nullptr, t, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr, m, 0, 64,
                                            nullptr));
    *h_bytes = m2_al(m * 4) + m2_al(t ? t : 16) + 256;
    return AVL_OK;
}

int avl_merge2_prepare(int64_t n, const int64_t* d_key, const int32_t* d_cell, int key_bits, int64_t flags, int64_t* d_key_sorted,
                       int32_t* d_cell_sorted, int32_t* d_perm, int64_t* d_hdr, void* d_work, size_t work_bytes, void* stream) {
    AVL_REQUIRE(n >= 0 && n < (1ll << 31) && d_hdr && key_bits >= 1 && key_bits <= 63, "avl_merge2_prepare: bad arguments");
    hipStream_t st = as_stream(stream);
    if (n > 0) {
        AVL_REQUIRE(d_key && d_cell && d_key_sorted && d_cell_sorted && d_perm, "avl_merge2_prepare: null pointer");
        size_t need = 0;
        int rc = avl_merge2_prepare_work_bytes(n, &need);
        if (rc != AVL_OK) return rc;
        AVL_REQUIRE(d_work && work_bytes >= need, "avl_merge2_prepare: work buffer of %zu bytes, %zu needed", work_bytes, need);
        char* base = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(d_work) + 255) / 256 * 256);
        int32_t* iota = reinterpret_cast<int32_t*>(base);
        void* tmp = base + m2_al((size_t)n * 4);
        size_t tb = need - 256 - m2_al((size_t)n * 4);
        hipLaunchKernelGGL(m2_iota_kernel, dim3(m2_grid(n)), dim3(256), 0, st, (long long)n, iota);
        AVL_HIP_CHECK(rocprim::radix_sort_pairs<M2SortCfg>(tmp, tb, reinterpret_cast<const unsigned long long*>(d_key), reinterpret_cast<unsigned long long*>(d_key_sorted),
                                                iota, d_perm, (size_t)n, 0, key_bits, st));
    }
    hipLaunchKernelGGL(m2_presorted_kernel, dim3(n > 0 ? m2_grid(n) : 1), dim3(256), 0, st, (long long)n, d_cell, d_perm,
                       reinterpret_cast<const long long*>(d_key_sorted), (long long)flags, d_cell_sorted, reinterpret_cast<long long*>(d_hdr));
    AVL_HIP_CHECK(hipGetLastError());
    return AVL_OK;
}

int avl_merge2_fold_work_bytes(int64_t n_own, int ws, size_t* h_bytes) {
    AVL_REQUIRE(h_bytes && n_own >= 0 && ws >= 1 && ws <= kM2MaxRanks, "avl_merge2_fold_work_bytes: bad arguments");
    const size_t m = (size_t)(n_own > 0 ? n_own : 1);
    *h_bytes = m2_al(m * ws * sizeof(int32_t)) + m2_al(m * 8) + m2_al(m) + 256;
    return AVL_OK;
}

// the per-chunk tables live in the entries kernel's LDS histogram next to the three whole-exchange ones: at most this many chunks
static int m2_max_chunks(int ws) { return std::max(0, (12288 / (ws * ws) - 3) / 2); }

int avl_merge2_max_chunks(int ws, int* h_max) {
    AVL_REQUIRE(h_max && ws >= 1 && ws <= kM2MaxRanks, "avl_merge2_max_chunks: bad arguments");
    *h_max = m2_max_chunks(ws);
    return AVL_OK;
}

int avl_merge2_work_bytes(int64_t E, int64_t n, int ws, int nchunk, size_t* h_bytes) {
    AVL_REQUIRE(h_bytes && E >= 0 && E < (1ll << 31) && n >= 0 && n <= E && ws >= 1 && ws <= kM2MaxRanks && nchunk >= 0 && nchunk <= m2_max_chunks(ws),
                "avl_merge2_work_bytes: bad arguments (at most %d ranks, 2^31 entries, avl_merge2_max_chunks chunks)", kM2MaxRanks);
    M2Layout L;
    int rc = m2_layout(E, n, ws, nchunk, L);
    if (rc != AVL_OK) return rc;
    *h_bytes = L.total;
    return AVL_OK;
}

int avl_merge2_plan(int ws, int rank, const int64_t* h_n_all, int64_t nmax, const int64_t* d_gathered, const int32_t* d_perm, int cell_bits,
                    int64_t grow_row, int want_replay_lists, int64_t chunk_rows, int nchunk, void* d_work, size_t work_bytes, int64_t* h_off,
                    int64_t* h_res, void* stream) {
    AVL_REQUIRE(ws >= 1 && ws <= kM2MaxRanks && rank >= 0 && rank < ws && h_n_all && h_off && h_res, "avl_merge2_plan: bad arguments");
    AVL_REQUIRE(nchunk >= 0 && nchunk <= m2_max_chunks(ws) && (nchunk == 0 || chunk_rows >= 1), "avl_merge2_plan: bad chunking (avl_merge2_max_chunks)");
    AVL_REQUIRE(cell_bits >= 1 && cell_bits <= 31, "avl_merge2_plan: cell_bits in [1, 31]");
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
    int rc = m2_layout(E, n, ws, nchunk, L);
    if (rc != AVL_OK) return rc;
    AVL_REQUIRE(d_work && work_bytes >= L.total, "avl_merge2_plan: work buffer of %zu bytes, %zu needed (avl_merge2_work_bytes)", work_bytes, L.total);
    char* base = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(d_work) + 255) / 256 * 256);
    const int64_t shift = base - reinterpret_cast<char*>(d_work);
    const size_t offs[11] = {L.row, L.prev, L.next, L.order, L.sidx, L.selA, L.selB, L.idx_prev, L.idx_next, L.rowcell, L.res};
    for (int k = 0; k < 11; ++k) h_off[k] = (int64_t)offs[k] + shift;
    hipStream_t st = as_stream(stream);
    const int nres = 2 + (3 + 2 * nchunk) * ws * ws;
    unsigned long long* res = reinterpret_cast<unsigned long long*>(base + L.res);
    AVL_HIP_CHECK(hipMemsetAsync(res, 0, (size_t)nres * 8, st));
    if (E > 0) {
        AVL_REQUIRE(d_gathered && (n == 0 || d_perm), "avl_merge2_plan: null lists");
        auto U32 = [&](size_t off) { return reinterpret_cast<uint32_t*>(base + off); };
        auto I32 = [&](size_t off) { return reinterpret_cast<int32_t*>(base + off); };
        const long long* g = reinterpret_cast<const long long*>(d_gathered);
        uint8_t* erank = reinterpret_cast<uint8_t*>(base + L.erank);
        uint8_t* headflag = reinterpret_cast<uint8_t*>(base + L.headflag);
        hipLaunchKernelGGL(m2_compact_kernel, dim3(m2_grid(E)), dim3(256), 0, st, E, ws, o, stride, (long long)nmax, g, U32(L.ecell), U32(L.eidx), erank);
        size_t tb = L.tmp_bytes;
        AVL_HIP_CHECK(rocprim::radix_sort_pairs<M2SortCfg>(base + L.tmp, tb, U32(L.ecell), U32(L.scell), U32(L.eidx), U32(L.se), (size_t)E, 0, cell_bits, st));
        hipLaunchKernelGGL(m2_segments_kernel, dim3(m2_grid(E)), dim3(256), 0, st, E, U32(L.scell), U32(L.se), erank, U32(L.hp),
                           reinterpret_cast<uint16_t*>(base + L.pn), headflag, res);
        {
            tb = L.tmp_bytes;
            auto it = rocprim::make_transform_iterator(reinterpret_cast<const uint8_t*>(headflag), FlagToI32{});
            AVL_HIP_CHECK(rocprim::exclusive_scan(base + L.tmp, tb, it, I32(L.rowscan), 0, (size_t)E, rocprim::plus<int32_t>(), st));
        }
        hipLaunchKernelGGL(m2_entries_kernel, dim3(std::min(m2_grid(E), 1024u)), dim3(256), (size_t)((3 + 2 * nchunk) * ws * ws) * sizeof(unsigned), st, E, ws,
                           rank, o, stride, g, U32(L.scell), U32(L.se), erank, U32(L.hp), reinterpret_cast<const uint16_t*>(base + L.pn), I32(L.rowscan),
                           d_perm, (long long)grow_row, (long long)(nchunk ? chunk_rows : 1), nchunk, res, I32(L.rowcell), I32(L.row), I32(L.prev), I32(L.next), U32(L.krow), U32(L.vslot));
        if (n > 0) {
            tb = L.tmp_bytes;
            AVL_HIP_CHECK(rocprim::radix_sort_pairs<M2SortCfg>(base + L.tmp, tb, U32(L.krow), U32(L.krow_s), U32(L.vslot), U32(L.order), (size_t)n, 0,
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
                AVL_HIP_CHECK(rocprim::radix_sort_pairs<M2SortCfg>(base + L.tmp, tb, U32(L.kp), U32(L.kps), I32(L.order), I32(L.idx_prev), (size_t)n, 0, rb, st));
                tb = L.tmp_bytes;
                AVL_HIP_CHECK(rocprim::radix_sort_pairs<M2SortCfg>(base + L.tmp, tb, U32(L.kn), U32(L.kps), I32(L.order), I32(L.idx_next), (size_t)n, 0, rb, st));
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

int avl_merge2_side_state(int64_t n, int ws, const int64_t* h_cum, const int64_t* h_lo, const int64_t* h_side_off, const int32_t* d_order,
                          const int32_t* d_next, const int64_t* d_state, int64_t* d_send, void* stream) {
    AVL_REQUIRE(n >= 0 && ws >= 1 && ws <= kM2MaxRanks && h_cum && h_lo && h_side_off, "avl_merge2_side_state: bad arguments");
    if (n == 0) return AVL_OK;
    AVL_REQUIRE(d_order && d_next && d_send, "avl_merge2_side_state: null pointer");
    M2Seg sg;
    for (int q = 0; q <= kM2MaxRanks; ++q) sg.cum[q] = h_cum[q < ws ? q : ws];
    for (int q = 0; q < kM2MaxRanks; ++q) {
        sg.lo[q] = q < ws ? h_lo[q] : 0;
        sg.side_off[q] = q < ws ? h_side_off[q] : 0;
    }
    AVL_REQUIRE(sg.cum[ws] == n, "avl_merge2_side_state: the destination ranges cover %lld voxels, n = %lld", sg.cum[ws], (long long)n);
    hipLaunchKernelGGL(m2_side_state_kernel, dim3(m2_grid(n)), dim3(256), 0, as_stream(stream), (long long)n, ws, sg, d_order, d_next,
                       reinterpret_cast<const long long*>(d_state), reinterpret_cast<long long*>(d_send));
    AVL_HIP_CHECK(hipGetLastError());
    return AVL_OK;
}

int avl_merge2_fold(int64_t n_own, int64_t r0, int ws, int D, int gs, int vh, const void* const* h_side, const void* const* h_done,
                    const void* const* h_part, const int64_t* h_count, const int32_t* d_rowcell, int have_log, float* d_grid_feat,
                    int32_t* d_grid_pos, float* d_weight, uint8_t* d_grid_rgb, int32_t* d_cell, void* d_work, size_t work_bytes, int32_t* d_err_flag,
                    void* stream) {
    AVL_REQUIRE(n_own >= 0 && r0 >= 0 && ws >= 1 && ws <= kM2MaxRanks && D > 0 && gs > 0 && vh > 0, "avl_merge2_fold: bad shape");
    if (n_own == 0) return AVL_OK;
    AVL_REQUIRE(h_side && h_done && h_part && h_count && d_rowcell && d_grid_feat && d_grid_pos && d_weight && d_grid_rgb && d_work,
                "avl_merge2_fold: null pointer");
    size_t need_bytes = 0;
    (void)avl_merge2_fold_work_bytes(n_own, ws, &need_bytes);
    AVL_REQUIRE(work_bytes >= need_bytes && (reinterpret_cast<uintptr_t>(d_work) & 255) == 0, "avl_merge2_fold: %zu bytes of 256-aligned work needed (avl_merge2_fold_work_bytes)",
                need_bytes);
    int32_t* d_table = reinterpret_cast<int32_t*>(d_work);
    M2Peers pe{};
    long long R = 0;
    for (int p = 0; p < ws; ++p) {
        pe.side[p] = reinterpret_cast<const long long*>(h_side[p]);
        pe.done[p] = reinterpret_cast<const float*>(h_done[p]);
        pe.part[p] = reinterpret_cast<const double*>(h_part[p]);
        pe.count[p] = h_count[p];
        R += h_count[p];
        AVL_REQUIRE(h_count[p] == 0 || h_side[p], "avl_merge2_fold: peer %d has records but no buffer", p);
    }
    hipStream_t st = as_stream(stream);
    AVL_HIP_CHECK(hipMemsetAsync(d_table, 0xFF, (size_t)n_own * ws * sizeof(int32_t), st));
    if (R > 0)
        hipLaunchKernelGGL(m2_fold_index_kernel, dim3(m2_grid(R)), dim3(256), 0, st, R, ws, pe, (long long)n_own, d_table, reinterpret_cast<int*>(d_err_flag));
    const long long ldf = (D + 1) / 2 * 2;      // float32 rows are padded to whole 8-byte words
    // scratch behind the table: sum alpha (f64) and the "features still to do" flag of every row
    char* sb = reinterpret_cast<char*>(d_table) + m2_al((size_t)n_own * ws * sizeof(int32_t));
    double* wsum = reinterpret_cast<double*>(sb);
    uint8_t* need = reinterpret_cast<uint8_t*>(sb + m2_al((size_t)n_own * 8));
    hipLaunchKernelGGL(m2_fold_scalar_kernel, dim3(m2_grid(n_own)), dim3(256), 0, st, (long long)n_own, (long long)r0, ws, gs, vh, pe, d_table, d_rowcell,
                       have_log, d_grid_pos, d_weight, d_grid_rgb, d_cell, wsum, need, reinterpret_cast<int*>(d_err_flag));
    int64_t blocks = ((n_own + 15) / 16 + 3) / 4;
    const int64_t maxb = (int64_t)num_cus() * 32;
    if (blocks > maxb) blocks = maxb;
    hipLaunchKernelGGL(m2_fold_feat_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (long long)n_own, ws, D, ldf, (long long)D, pe, d_table, wsum, need,
                       d_grid_feat);
    AVL_HIP_CHECK(hipGetLastError());
    return AVL_OK;
}

}  // extern "C"
