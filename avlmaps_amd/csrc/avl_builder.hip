// Map builder for gfx950: depth back-projection + voxelisation + distance-weighted feature fusion.
//
// Replaces the per-frame body of the upstream reference's builder loop (path:line, upstream repo root):
//   avlmaps/map/vlmap_builder.py:129-178          per-point Python loop
//   avlmaps/utils/mapping_utils.py:226-251        depth2pc            (only for the sampled pixels)
//   avlmaps/utils/mapping_utils.py:305-315        transform_pc
//   avlmaps/utils/mapping_utils.py:345-349        base_pos2grid_id_3d (fp64 divide, trunc toward zero)
//   avlmaps/utils/mapping_utils.py:599-605        project_point       (rgb pixel, LSeg pixel)
//
// This file is compiled with -ffp-contract=off: the ONLY fused multiply-adds in the index math are the
// explicit fma() calls that mirror the reference's BLAS dgemm / dgemv summation (see oracle/avl_oracle.c).
//
// Device state (all in HBM, sized for a 288 GB part -- dense tables instead of a probing hash):
//   cell_slot  int32 [gs*gs*vh]   cell -> slot (-1 empty)
//   slot_cell  int32 [cap]        slot -> linear cell
//   slot_key   uint64[cap]        first-touch key (frame << 32 | position in the frame's sample list), ~0 until known
//   sum_feat   f64   [cap, D]     sum_i alpha_i * f_i
//   sum_w4     f64   [cap, 4]     sum alpha, sum alpha*(r,g,b)
//   first_feat f32   [cap, D]     feature of the first-touch point, first_alpha f64 [cap]
//   head       int32 [cap]        per-frame list head of the samples that hit the voxel (-1 between frames)
// The reference's order-dependent result has the closed form
//   grid_feat = (sum_feat - a1*(1-a1)*first_feat) / sum_alpha,  weight = sum_alpha
// so the accumulation itself is commutative; only the first-touch point (smallest key) is special.
//
// Per launch (one frame, or the B frames of a batch) two kernels on the caller's stream, none of them serial:
//   voxelize_link_kernel, thread per sampled pixel:
//     K1 bp_voxelize  fp64 geometry; an empty cell is created by the CAS winner (slot = atomicAdd on the voxel counter)
//     K2 link         push the sample on its voxel's list (atomicExch on head); the first pusher becomes the voxel's owner
//                     for this launch.  A sample whose cell is being created by another workgroup polls for the slot
//                     (bounded; the creator needs one atomic and one store) -- K1 and K2 used to be two launches
//   fuse_kernel:
//   K3 fuse         wave per sampled pixel, owners only: walk the list, gather each member's channels-last
//                   feature row (2 KB contiguous at D=512), accumulate alpha*f in fp64 REGISTERS, then ONE plain
//                   read-modify-write of the voxel row (store-only for a voxel born this frame).  No floating-point
//                   atomics: an earlier version used 512 fp64 atomics per point and was bound by the L2 atomic rate
//                   (36 us/frame, 47 MB of write traffic for 11 MB of algorithmic RMW).  The owner also finds the
//                   voxel's first-touch sample (smallest position) when the voxel is new.
// Deferred fuse (avl_builder_set_deferred_fuse): pipe_kernel = voxelize_link of frame i + fuse of frame i - 1 in one launch.
// The frame loop in C (avl_builder_integrate_frames) additionally runs the map-independent half of K1 for frame i + 1 -- everything
// in front of the cell_slot lookup -- in workgroups of its own at the front of frame i's launch (struct PreGather,
// voxelize_link_next_kernel / pipe_kernel), so that a frame's dependent chain starts at the voxel-hash lookup.
// K1 and K3 are written as STRAIGHT-LINE code around their loads (flags and selects, full-width rows): a load inside a branch makes
// the compiler wait for everything outstanding where the branches merge, and these kernels live on having several loads in flight.
// Slots are handed out in arrival order, so finalisation sorts the first-touch keys (rocPRIM radix sort) to emit
// rows in the reference's voxel-id order; the sort is a once-per-save cost.
#include <algorithm>
#include <climits>
#include <cstring>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#include "avl_common.h"

namespace avl {

#ifdef AVL_PROBE_CHAIN
// tools/probe_chain.py: per-work-item timestamps (s_memrealtime, 100 MHz) of the dependent-load chains of K1 + K2 and K3, taken
// AFTER the loads before them have landed.  Never compiled into the shipped library (tools/build_variant.py -DAVL_PROBE_CHAIN).
__device__ unsigned long long* g_probe = nullptr;
constexpr int kProbeK12Row0 = 16384;
#define AVL_STAMP(var)                                                                                             \
    unsigned long long var;                                                                                        \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory")
#define AVL_STAMP_DECL(var) unsigned long long var = 0
#define AVL_RESTAMP(var) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory")
#else
#define AVL_STAMP(var)
#define AVL_STAMP_DECL(var)
#define AVL_RESTAMP(var)
#endif

// one frame of a batch (avl_builder_integrate_batch): what differs between the frames of one launch
struct BatchEntry {
    double t[16];                 // pc_transform of the frame
    const float* depth;
    const int32_t* samples;
    const uint8_t* rgb;
    const float* feat;
    unsigned long long frame_key; // key_bias | frame_idx << 32
};

struct FrameParams {
    double kinv[9];   // inv(calib)            (mapping_utils.py:237)
    double k[9];      // calib                 (vlmap_builder.py:98)
    double kf[9];     // get_sim_cam_mat(Hf,Wf)(mapping_utils.py:591-596)
    double t[16];     // pc_transform          (vlmap_builder.py:133)
    double min_depth, max_depth, two_sigma_sq;
    double cs, half_gs;
    double pcd_min[3];   // global mode: lower corner of the pass-1 bounding box (vlmap_builder_multi_floor.py:117)
    double depth_div;    // uint16 depth images: metres = value / depth_div (multi-floor: / 1000.0, :105)
    int H, W, Hf, Wf, n0, n1, n2, P;   // grid: n0 rows x n1 cols x n2 heights (mobile-base mode: gs, gs, vh)
    const BatchEntry* batch;   // nullptr: single frame (pointers / transform passed directly); else P = B * P_frame samples
    int P_frame;
    int mode;            // 0 = mobile-base map (vlmap_builder.py), 1 = global multi-floor map (vlmap_builder_multi_floor.py)
    int depth_u16;
    long long capacity;
    const struct PreRec* pre;   // the stateless half of K1 for THIS frame, computed by the previous launch (PreGather); nullptr: compute here
};

// The frame loop in C (avl_builder_integrate_frames) knows frame i + 1 while it launches frame i, and the first half of K1 --
// sample index -> depth -> back-projection, pose, cell, the two pinhole projections, the colour gather, the weight -- depends on
// nothing the builder holds: only from the cell's slot onwards does a sample touch the map.  A few workgroups in FRONT of frame
// i's launch run that half for frame i + 1 (bp_voxelize_body<.., 2>) and leave 24 bytes per sample; K1 of frame i + 1
// (bp_voxelize_body<.., 1>) then starts its chain with one coalesced load and goes straight to the slot: four of the chain's
// hops (sample index, depth, geometry, colour) move out of the dependent path into the shadow of the previous frame.
// Single float32-depth frames of one avl_builder_integrate_frames call; the same arithmetic, operation for operation.
struct PreRec {
    double alpha;     // 0 if the sample dropped out
    int32_t cell;     // -1 if the sample dropped out
    int32_t fpix;
    uint32_t rgbv;
    uint32_t flags;   // bit 0: outside the pass-1 bounding box (global mode), bit 1: projected outside the RGB image
};
static_assert(sizeof(PreRec) == 24, "PreRec");

struct PreGather {
    PreRec* out;              // nullptr: nothing to prepare
    const float* depth;
    const int32_t* samples;
    const uint8_t* rgb;
};

// per-frame sample records (structure of arrays, sized for the largest P seen)
struct Recs {
    double* alpha;
    int32_t* slot;    // -1 = the sample updates no voxel
    int32_t* fpix;    // py*Wf + px into the (Hf, Wf, D) feature map
    uint32_t* rgb;    // r | g<<8 | b<<16
    int32_t* next;    // next sample of the same voxel in this frame, -1 = end
    uint8_t* owner;   // 1 = this sample found its voxel's list empty: it is the list's TAIL and its wave fuses the list
};
// The owners of a BATCHED launch compacted per K2 workgroup (no atomics: a ballot and four LDS words): entry 256 b + k is owner k of
// workgroup b, ocnt[b] of them.  K3 then runs kFuseWaves waves per K2 workgroup over them instead of one wave per SAMPLE, most of
// which load a flag and leave: half the workgroups to dispatch, +3 % / +6 % at 16 / 64 frames per launch.  Single-frame launches keep
// the wave-per-sample form (the compacted one measured 11.2 -> 12.0 us there), and these pointers stay out of their kernel arguments.
struct OwnerList {
    double* o_alpha;
    int32_t* o_s;
    int32_t* o_slot;
    int32_t* o_fpix;
    uint32_t* o_rgb;
    int32_t* ocnt;
};
#ifndef AVL_K3_THREADS
#define AVL_K3_THREADS 256      // wave-per-sample K3 launches (single frames): threads per workgroup
#endif
constexpr int kFuseWaves = 128;   // K3 waves per K2 workgroup of 256 samples (more owners than that: a wave takes several)

constexpr int kEmpty = -1, kPending = -2;
constexpr int kAggregateSamples = 32768;   // launches at least this large allocate slots per workgroup instead of per wave
constexpr unsigned long long kNoKey = ~0ull;
constexpr int kMaxPendingSpins = 1 << 18;   // link step of a deferred-fuse launch: polls of a cell that is being created

// The kernel-argument block of these kernels is 450-1 040 bytes (FrameParams alone: three 3 x 3 matrices and the pose), the
// compiler fetches it piecemeal, each piece where it is first needed and behind the branch that needs it, and every piece is a
// scalar-cache miss that goes to memory: several dependent misses at the head of every work item's chain.  Touch every line of the
// block at once, first thing in the kernels that read most of it (LINES x 64 bytes, LINES in {4, 8, 12, 16} and not beyond the segment): one miss time for all of them.
template <int LINES>
__device__ __forceinline__ void warm_kernel_arguments() {
#ifndef AVL_NO_KERNARG_WARMUP
    static_assert(LINES == 4 || LINES == 8 || LINES == 12 || LINES == 16, "four lines per step");
    const auto p = __builtin_amdgcn_kernarg_segment_ptr();     // (an address_space(4) pointer: two scalar registers)
    int t0, t1, t2, t3;
    if constexpr (LINES == 4)
        asm volatile("s_load_dword %0, %4, 0x0\n s_load_dword %1, %4, 0x40\n s_load_dword %2, %4, 0x80\n s_load_dword %3, %4, 0xc0\n"
                     "s_waitcnt lgkmcnt(0)" : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3) : "s"(p));
    if constexpr (LINES == 8)
        asm volatile("s_load_dword %0, %4, 0x0\n s_load_dword %1, %4, 0x40\n s_load_dword %2, %4, 0x80\n s_load_dword %3, %4, 0xc0\n"
                     "s_load_dword %0, %4, 0x100\n s_load_dword %1, %4, 0x140\n s_load_dword %2, %4, 0x180\n s_load_dword %3, %4, 0x1c0\n"
                     "s_waitcnt lgkmcnt(0)" : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3) : "s"(p));
    if constexpr (LINES == 12)
        asm volatile("s_load_dword %0, %4, 0x0\n s_load_dword %1, %4, 0x40\n s_load_dword %2, %4, 0x80\n s_load_dword %3, %4, 0xc0\n"
                     "s_load_dword %0, %4, 0x100\n s_load_dword %1, %4, 0x140\n s_load_dword %2, %4, 0x180\n s_load_dword %3, %4, 0x1c0\n"
                     "s_load_dword %0, %4, 0x200\n s_load_dword %1, %4, 0x240\n s_load_dword %2, %4, 0x280\n s_load_dword %3, %4, 0x2c0\n"
                     "s_waitcnt lgkmcnt(0)" : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3) : "s"(p));
    if constexpr (LINES == 16)
        asm volatile("s_load_dword %0, %4, 0x0\n s_load_dword %1, %4, 0x40\n s_load_dword %2, %4, 0x80\n s_load_dword %3, %4, 0xc0\n"
                     "s_load_dword %0, %4, 0x100\n s_load_dword %1, %4, 0x140\n s_load_dword %2, %4, 0x180\n s_load_dword %3, %4, 0x1c0\n"
                     "s_load_dword %0, %4, 0x200\n s_load_dword %1, %4, 0x240\n s_load_dword %2, %4, 0x280\n s_load_dword %3, %4, 0x2c0\n"
                     "s_load_dword %0, %4, 0x300\n s_load_dword %1, %4, 0x340\n s_load_dword %2, %4, 0x380\n s_load_dword %3, %4, 0x3c0\n"
                     "s_waitcnt lgkmcnt(0)" : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3) : "s"(p));
    (void)t0; (void)t1; (void)t2; (void)t3;
    __builtin_amdgcn_sched_barrier(0);     // (nothing of the kernel is scheduled in front of the touches)
#endif
}

__device__ __forceinline__ int py_int(double v) {
    // Python int(): truncate toward zero.  Far-out values saturate at +-2e9 and NaN goes to -2e9 (all of them fail the range
    // tests that follow: grid, image and feature-image bounds are far below that), which lets the conversion be ONE instruction
    // (v_cvt_i32_f64) instead of the ~15 of a float64 -> int64 conversion -- five of those sat on every sample's chain in K1.
    return (int)fmin(fmax(v, -2.0e9), 2.0e9);
}

__device__ __forceinline__ double gemv3(const double* a, double x0, double x1, double x2) {
    // numpy (3,3)@(3,1) -> OpenBLAS dgemv tail:  fma(a2,x2, fma(a0,x0, a1*x1))   (oracle/avl_oracle.c)
    return fma(a[2], x2, fma(a[0], x0, a[1] * x1));
}

// what K1 leaves in registers for a fused K2 (deferred-fuse launches run both in one kernel)
struct SampleRec {
    double alpha;
    int32_t cell, fpix;
    uint32_t rgbv;
    int32_t slot;   // >= 0: the voxel's slot is already known (cell seen occupied, or created by this sample); else -1
};

// K1 body: block `blk` of a launch over fp.P samples.  Slots are published with agent-scope atomic stores so that a
// K2 running in the SAME kernel (other workgroups, other XCDs) can wait for them.
// (THREADS = the workgroup size, a compile-time constant: blockDim.x is a hidden kernel argument on a cache line of its own,
// i.e. one more scalar miss at the head of every chain)
// MODE 0: all of K1.  MODE 2: its stateless half only, for the NEXT frame, result to pre_out (PreGather).  MODE 1: the stateful half,
// starting from what a MODE-2 workgroup of the previous launch left in fp.pre.
template <int THREADS, int MODE = 0>
__device__ __forceinline__ SampleRec bp_voxelize_body(int blk, const FrameParams& fp, const float* depth,
                                                      const int32_t* __restrict__ sample_idx, const uint8_t* rgb,
                                                      int32_t* __restrict__ cell_slot, int32_t* __restrict__ slot_cell, const Recs& recs,
                                                      unsigned long long* __restrict__ counters, int* __restrict__ err_flags,
                                                      PreRec* __restrict__ pre_out = nullptr) {
    // slots are allocated per workgroup: creators are counted in LDS and ONE device atomic per 256 samples reserves the
    // block's range (a single hot device word only sustains ~90 atomics/us)
    __shared__ unsigned blk_new;
    __shared__ unsigned long long blk_base;
    if constexpr (MODE != 2) {
        if (threadIdx.x == 0) blk_new = 0;
        __syncthreads();
    }
    const int s = blk * THREADS + threadIdx.x;   // global sample index: frame-major within a batch
    const bool valid = s < fp.P;
    AVL_STAMP(pt0);
    AVL_STAMP_DECL(pt1);
    AVL_STAMP_DECL(pt2);
    AVL_STAMP_DECL(pt3);
    double alpha = 0.0;
    int32_t cell = -1, fpix = 0;
    uint32_t rgbv = 0;
    int32_t seen = kEmpty;
    bool ok = false, err_box = false, err_img = false;
    uint8_t c0 = 0, c1 = 0, c2 = 0;
    if constexpr (MODE == 1) {
        // 24 bytes per sample from the previous launch, then straight to the cell's slot
        const PreRec r = valid ? fp.pre[s] : PreRec{0.0, -1, 0, 0u, 0u};
        alpha = r.alpha;
        cell = r.cell;
        fpix = r.fpix;
        rgbv = r.rgbv;
        ok = cell >= 0;
        err_box = (r.flags & 1u) != 0;
        err_img = (r.flags & 2u) != 0;
        AVL_RESTAMP(pt1);
        seen = cell_slot[ok ? cell : 0];
        AVL_RESTAMP(pt2);
        AVL_RESTAMP(pt3);
    } else {

    // (kernel-uniform null test: a single-frame launch reads its pose from the kernel arguments, i.e. scalar registers -- selecting
    // per lane between be->t and fp.t made twelve vector loads behind twelve branches, issued only after the depth had arrived:
    // one more memory round trip on every sample's chain)
    const BatchEntry* be = fp.batch ? fp.batch + (valid ? s / fp.P_frame : 0) : nullptr;
    double T[12];
    if (be) {
#pragma unroll
        for (int k = 0; k < 12; ++k) T[k] = be->t[k];
        depth = be->depth;
        rgb = be->rgb;
    } else {
#pragma unroll
        for (int k = 0; k < 12; ++k) T[k] = fp.t[k];
    }
    const int pix = !valid ? -1 : (be ? be->samples[s % fp.P_frame] : sample_idx[s]);
    const bool ok0 = pix >= 0 && pix < fp.H * fp.W;
    AVL_RESTAMP(pt1);
    // From here to the CAS the code is STRAIGHT-LINE (flags and selects, no branch around a load): the sample's chain is depth ->
    // cell -> cell_slot, and the cell's slot is requested as soon as the cell is known; the two pinhole projections (four of the
    // seven fp64 divides), the colour gather and the weight's exp() are computed while that load is in flight.  They used to sit
    // in `if (ok)` blocks in FRONT of it: ~1.4-1.7 us of dependent fp64 arithmetic on every sample's chain (one wave per SIMD in
    // these workgroups: nothing hides it).  A sample that dropped out keeps computing on harmless values and reads element 0.
    // The arithmetic of a surviving sample is unchanged, operation for operation.
    // depth2pc: p_2d = (u + 0.5, v + 0.5, 1); pc = Kinv @ p_2d (dgemm: FMA chain over k); pc *= z
    const int pixc = ok0 ? pix : 0;
    const float* dsrc = ok0 ? depth : reinterpret_cast<const float*>(cell_slot);   // (batched launches carry no frame-level pointers)
    const double x = (double)(pixc % fp.W) + 0.5, y = (double)(pixc / fp.W) + 0.5;
    const double z = fp.depth_u16 ? (double)reinterpret_cast<const uint16_t*>(dsrc)[pixc] / fp.depth_div : (double)dsrc[pixc];
    const double pl0 = fma(fp.kinv[2], 1.0, fma(fp.kinv[1], y, fp.kinv[0] * x)) * z;
    const double pl1 = fma(fp.kinv[5], 1.0, fma(fp.kinv[4], y, fp.kinv[3] * x)) * z;
    const double pl2 = fma(fp.kinv[8], 1.0, fma(fp.kinv[7], y, fp.kinv[6] * x)) * z;
    const bool ok1 = ok0 && (pl2 > fp.min_depth) && (pl2 < fp.max_depth);  // strict on both sides, NaN fails
    AVL_RESTAMP(pt2);
    // transform_pc: pose @ [pc; 1]  (dgemm FMA chain k = 0..3)
    const double g0 = fma(T[3], 1.0, fma(T[2], pl2, fma(T[1], pl1, T[0] * pl0)));
    const double g1 = fma(T[7], 1.0, fma(T[6], pl2, fma(T[5], pl1, T[4] * pl0)));
    const double g2 = fma(T[11], 1.0, fma(T[10], pl2, fma(T[9], pl1, T[8] * pl0)));
    int row, col, h;          // (32-bit from here on: saturated values fail the range tests, in-range ones are small)
    if (fp.mode == 0) {       // (kernel-uniform)
        // base_pos2grid_id_3d: int(gs/2 - int(x/cs)) with a true fp64 divide
        row = py_int(fp.half_gs - (double)py_int(g0 / fp.cs));
        col = py_int(fp.half_gs - (double)py_int(g1 / fp.cs));
        h = py_int(g2 / fp.cs);
    } else {
        // row, height, col = np.round((p - pcd_min) / cs).astype(int)   (vlmap_builder_multi_floor.py:146; half-to-even)
        row = py_int(rint((g0 - fp.pcd_min[0]) / fp.cs));
        h = py_int(rint((g1 - fp.pcd_min[1]) / fp.cs));
        col = py_int(rint((g2 - fp.pcd_min[2]) / fp.cs));
    }
    const bool in_grid = !(col >= fp.n1 || row >= fp.n0 || h >= fp.n2 || col < 0 || row < 0 || h < 0);
    const bool ok2 = ok1 && in_grid;
    // global mode: the reference only tests the upper row/col bounds and otherwise wraps or raises; a point outside
    // the pass-1 bounding box is dropped here and reported
    err_box = ok1 && !in_grid && fp.mode == 1;
    const int32_t cell_try = ok2 ? (int32_t)(((unsigned)row * (unsigned)fp.n1 + (unsigned)col) * (unsigned)fp.n2 + (unsigned)h) : 0;
    // cell states only move forward (empty -> pending -> slot), so a slot read here is final even if the line is old
    if constexpr (MODE == 0) seen = cell_slot[cell_try];
    __builtin_amdgcn_sched_barrier(0);      // (the scheduler otherwise sinks the request below the arithmetic that follows)
    // project_point(calib, p_local) -> rgb[py, px] with numpy's negative-index wrap
    double q0 = gemv3(fp.k + 0, pl0, pl1, pl2), q1 = gemv3(fp.k + 3, pl0, pl1, pl2), q2 = gemv3(fp.k + 6, pl0, pl1, pl2);
    int px = py_int(q0 / q2 - 0.5), py = py_int(q1 / q2 - 0.5);
    px += px < 0 ? fp.W : 0;
    py += py < 0 ? fp.H : 0;
    const bool in_img = !(px < 0 || px >= fp.W || py < 0 || py >= fp.H);
    err_img = ok2 && !in_img;    // the reference raises IndexError here; we drop the point and flag it
    const bool ok3 = ok2 && in_img;
    const size_t rgb_off = ok3 ? ((size_t)py * fp.W + px) * 3 : 0;
    // the colour gather (a sample that dropped out reads element 0 of cell_slot: batched launches carry no frame-level rgb pointer)
    // (global address space stated: a FLAT load -- batched launches read the pointer from memory -- may return out of order with the
    // cell_slot load, so the wait before the CAS would have to be for both)
    using gbyte_ptr = const __attribute__((address_space(1))) uint8_t*;
    const gbyte_ptr c = (gbyte_ptr)(ok3 ? rgb + rgb_off : reinterpret_cast<const uint8_t*>(cell_slot));
    c0 = c[0];
    c1 = c[1];
    c2 = c[2];
    __builtin_amdgcn_sched_barrier(0);
    // project_point(get_sim_cam_mat(Hf, Wf), p_local) -> feature pixel, bounds-checked (vlmap_builder.py:161)
    q0 = gemv3(fp.kf + 0, pl0, pl1, pl2);
    q1 = gemv3(fp.kf + 3, pl0, pl1, pl2);
    q2 = gemv3(fp.kf + 6, pl0, pl1, pl2);
    px = py_int(q0 / q2 - 0.5);
    py = py_int(q1 / q2 - 0.5);
    ok = ok3 && !(px < 0 || py < 0 || px >= fp.Wf || py >= fp.Hf);
    if (ok2) fpix = (int32_t)((unsigned)py * (unsigned)fp.Wf + (unsigned)px);
    const double radial = (pl0 * pl0 + pl1 * pl1) + pl2 * pl2;  // np.sum(np.square(p_local))
    const double alpha_try = exp(-radial / fp.two_sigma_sq);
    if (ok) {
        alpha = alpha_try;
        cell = cell_try;
    }
    AVL_RESTAMP(pt3);
    }   // (MODE != 1)
    if constexpr (MODE == 2) {
        if (valid)
            pre_out[s] = PreRec{alpha, cell, fpix, ok ? ((uint32_t)c0 | ((uint32_t)c1 << 8) | ((uint32_t)c2 << 16)) : 0u,
                                (err_box ? 1u : 0u) | (err_img ? 2u : 0u)};
        return SampleRec{};
    }
    // create the voxel if the cell is empty: the CAS winner takes the next slot.  One counter atomic per wave:
    // winners are ranked with a ballot (a single hot word only sustains ~90 atomics/us).
    int32_t known = -1;
    bool creator = false;
    if (ok) {
        if (seen == kEmpty) creator = atomicCAS(&cell_slot[cell], kEmpty, kPending) == kEmpty;
        else if (seen >= 0) known = seen;
    }
    AVL_STAMP(pt4);
    const unsigned long long cmask = __ballot(creator);
    const int lane = threadIdx.x & 63;
    unsigned long long base = 0;
#ifndef AVL_K1_AGG_MIN
#define AVL_K1_AGG_MIN kAggregateSamples
#endif
    if (fp.P >= AVL_K1_AGG_MIN) {   // kernel-uniform: batched launches take the two barriers, single frames do not
        unsigned woff = 0;
        if (cmask) {
            const int leader = __ffsll((long long)cmask) - 1;
            if (lane == leader) woff = atomicAdd(&blk_new, (unsigned)__popcll(cmask));   // LDS: the wave's offset in the block
            woff = __shfl(woff, leader, 64);
        }
        __syncthreads();
        if (threadIdx.x == 0 && blk_new) blk_base = atomicAdd(&counters[0], (unsigned long long)blk_new);
        __syncthreads();
        base = blk_base + woff;
    } else if (cmask) {
        const int leader = __ffsll((long long)cmask) - 1;
        if (lane == leader) base = atomicAdd(&counters[0], (unsigned long long)__popcll(cmask));
        base = __shfl(base, leader, 64);
    }
    if (creator) {
        const unsigned long long slot = base + __popcll(cmask & ((1ull << lane) - 1ull));
        if ((long long)slot >= fp.capacity) {
            atomicOr(err_flags, 1);
            // give the cell back; its samples are dropped in K2
            __hip_atomic_store(&cell_slot[cell], kEmpty, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            slot_cell[slot] = cell;
            __hip_atomic_store(&cell_slot[cell], (int32_t)slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            known = (int32_t)slot;
        }
    }
    // (the colour bytes are first looked at HERE: the CAS above only waits for the cell's slot, the gather is still in flight then)
    if (MODE == 0 && ok) rgbv = (uint32_t)c0 | ((uint32_t)c1 << 8) | ((uint32_t)c2 << 16);
    if (valid) {
        recs.alpha[s] = alpha;
        recs.fpix[s] = fpix;
        recs.rgb[s] = rgbv;
    }
    if (err_box) atomicOr(err_flags, 8);
    if (err_img) atomicOr(err_flags, 2);
#ifdef AVL_PROBE_CHAIN
    {
        AVL_STAMP(pt5);
        if (g_probe && valid && fp.P < kProbeK12Row0) {
            unsigned long long* q = g_probe + (size_t)(kProbeK12Row0 + s) * 8;
            q[0] = pt0; q[1] = pt1; q[2] = pt2; q[3] = pt3; q[4] = pt4; q[5] = pt5;
        }
    }
#endif
    return SampleRec{alpha, cell, fpix, rgbv, known};
}

// optional per-sample log for the exact sequential replay of weight / grid_rgb at finalisation (position = key order)
// One 32-byte record per sample = one memory sector: the replay walks a voxel's samples through an index list, i.e. every entry is
// a random access -- with alpha / key / colour in three arrays that was three sectors per entry (replay_chain_kernel 1.63 ms for the
// 27 M active samples of a 10 000-frame build).  The slots stay in an array of their own: the compaction and the sort read only them.
struct alignas(32) LogRec {
    double alpha;
    unsigned long long key;
    uint32_t rgb;
    uint32_t pad[3];
};
static_assert(sizeof(LogRec) == 32, "one sector per replay-log record");
struct ReplayLog {
    uint32_t* slot;            // 0xFFFFFFFF = sample did not update a voxel
    LogRec* rec;
};

// K2 body.  Runs right behind K1 in the same kernel: the sample comes in registers, and a cell another workgroup is still
// creating (kPending) is waited for.  That cannot deadlock: a creator publishes its slot without waiting for anybody but
// its own workgroup's barrier (batched launches), which every wave of the workgroup reaches before it spins.
template <bool MAY_COMPACT = true, int THREADS = 256>
__device__ __forceinline__ void link_body(int blk, int P, const int32_t* __restrict__ cell_slot, int32_t* __restrict__ head,
                                          const Recs& recs, unsigned long long* __restrict__ counters, const ReplayLog& log,
                                          long long log_base, unsigned long long frame_key, const BatchEntry* __restrict__ batch,
                                          int P_frame, const SampleRec& in, int* __restrict__ err_flags, const OwnerList& ol = OwnerList{}) {
    // statistics counters are aggregated per workgroup in LDS: a single hot device word sustains only ~90 atomics/us,
    // which at one atomic per wave was most of this kernel's time in batched launches
    __shared__ unsigned blk_cnt[2];
    __shared__ unsigned sh_own[4];
    if (threadIdx.x < 2) blk_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int s = blk * THREADS + threadIdx.x;
    const bool valid = s < P;
    int32_t slot = -1, next = -1;
    uint8_t owner = 0;
    if (valid) {
        const int32_t cell = in.cell;
        if (cell >= 0) {
            {
                // most samples hit a voxel of an earlier frame: K1 has read its slot already
                slot = in.slot >= 0 ? in.slot : __hip_atomic_load(&cell_slot[cell], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // the creator needs one counter atomic and one store (~1-2 us); the cap (~0.2 s) turns a wait that should be
                // impossible into AVL_ERR_STATE at the next flag check instead of a hung device
                for (int spins = 0; slot == kPending; ++spins) {
                    if (spins >= kMaxPendingSpins) {
                        atomicOr(err_flags, 16);
                        slot = -1;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                    slot = __hip_atomic_load(&cell_slot[cell], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (slot >= 0) {
                next = atomicExch(&head[slot], s);  // LIFO push; whoever finds the list empty owns it this launch
                owner = next == -1;
            }
        }
    }
    const unsigned long long amask = __ballot(slot >= 0);
    const unsigned long long omask = __ballot(owner != 0);
    if (amask && (threadIdx.x & 63) == __ffsll((long long)amask) - 1) {
        atomicAdd(&blk_cnt[0], (unsigned)__popcll(amask));
        if (omask) atomicAdd(&blk_cnt[1], (unsigned)__popcll(omask));
    }
    if (MAY_COMPACT && (threadIdx.x & 63) == 0) sh_own[threadIdx.x >> 6] = (unsigned)__popcll(omask);
    if (valid) {
        recs.slot[s] = slot;
        recs.next[s] = next;
        recs.owner[s] = owner;
        if (log.slot) {
            const long long i = log_base + s;
            log.slot[i] = slot >= 0 ? (uint32_t)slot : 0xFFFFFFFFu;
#ifdef AVL_ABL_NOLOGREC   // timing ablation only (WRONG replay): what the record store costs a frame
            if (false) {
#elif defined(AVL_LOG_UNCOND)
            {
#else
            if (slot >= 0) {     // (the replay only ever reads the records of samples that updated a voxel)
#endif
                const unsigned long long k = batch ? (batch[s / P_frame].frame_key | (unsigned)(s % P_frame)) : (frame_key | (unsigned)s);
                using u64x2 = __attribute__((ext_vector_type(2))) unsigned long long;
                u64x2* r = reinterpret_cast<u64x2*>(log.rec + i);
                r[0] = u64x2{(unsigned long long)__double_as_longlong(in.alpha), k};
                r[1] = u64x2{(unsigned long long)in.rgbv, 0ull};
            }
        }
    }
#ifdef AVL_PROBE_CHAIN
    {
        AVL_STAMP(pt7);
        if (g_probe && valid && P < kProbeK12Row0) g_probe[(size_t)(kProbeK12Row0 + s) * 8 + 7] = pt7 | ((unsigned long long)(slot >= 0) << 63);
    }
#endif
    __syncthreads();
    if (MAY_COMPACT && owner && P >= kAggregateSamples) {   // (kernel-uniform: only batched launches fuse from the compacted owners, see fuse_body)
        unsigned pos = (unsigned)__popcll(omask & ((1ull << (threadIdx.x & 63)) - 1ull));
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) pos += sh_own[w];
        const size_t o = (size_t)blk * 256 + pos;
        ol.o_s[o] = s;
        ol.o_slot[o] = slot;
        ol.o_alpha[o] = in.alpha;
        ol.o_fpix[o] = in.fpix;
        ol.o_rgb[o] = in.rgbv;
    }
    if (MAY_COMPACT && P >= kAggregateSamples && threadIdx.x == 0) ol.ocnt[blk] = (int32_t)(sh_own[0] + sh_own[1] + sh_own[2] + sh_own[3]);
    if (threadIdx.x < 2 && blk_cnt[threadIdx.x]) atomicAdd(&counters[1 + threadIdx.x], (unsigned long long)blk_cnt[threadIdx.x]);
}

// K1 + K2 of one launch (a frame, or the B frames of a batch) in ONE kernel: the record stays in registers between the two
// steps, and a sample whose cell is being created by another workgroup waits for the slot instead of for a kernel boundary
__global__ __launch_bounds__(256) void voxelize_link_kernel(FrameParams fp, const float* depth, const int32_t* __restrict__ sample_idx,
                                                            const uint8_t* rgb, int32_t* __restrict__ cell_slot,
                                                            int32_t* __restrict__ slot_cell, Recs recs, int32_t* __restrict__ head,
                                                            unsigned long long* __restrict__ counters, int* __restrict__ err_flags,
                                                            ReplayLog log, long long log_base, unsigned long long frame_key, OwnerList ol) {
    warm_kernel_arguments<12>();       // (944 bytes with the hidden arguments)
    const SampleRec r = bp_voxelize_body<256>(blockIdx.x, fp, depth, sample_idx, rgb, cell_slot, slot_cell, recs, counters, err_flags);
    link_body<true, 256>(blockIdx.x, fp.P, cell_slot, head, recs, counters, log, log_base, frame_key, fp.batch, fp.P_frame, r, err_flags, ol);
}

// The same for one frame of the C frame loop (avl_builder_integrate_frames): gb workgroups in FRONT run the stateless half of K1 for
// the NEXT frame (fpn / next, see PreGather), and this frame's K1 starts from what the previous launch prepared when fp.pre is set.
__global__ __launch_bounds__(256) void voxelize_link_next_kernel(FrameParams fp, const float* depth, const int32_t* __restrict__ sample_idx,
                                                                 const uint8_t* rgb, int32_t* __restrict__ cell_slot,
                                                                 int32_t* __restrict__ slot_cell, Recs recs, int32_t* __restrict__ head,
                                                                 unsigned long long* __restrict__ counters, int* __restrict__ err_flags,
                                                                 ReplayLog log, long long log_base, unsigned long long frame_key,
                                                                 FrameParams fpn, PreGather next, int gb) {
    if ((int)blockIdx.x < gb) {
        bp_voxelize_body<256, 2>((int)blockIdx.x, fpn, next.depth, next.samples, next.rgb, cell_slot, slot_cell, recs, counters, err_flags, next.out);
        return;
    }
    const int blk = (int)blockIdx.x - gb;
    warm_kernel_arguments<12>();
    SampleRec r;
    if (fp.pre) r = bp_voxelize_body<256, 1>(blk, fp, depth, sample_idx, rgb, cell_slot, slot_cell, recs, counters, err_flags);
    else r = bp_voxelize_body<256, 0>(blk, fp, depth, sample_idx, rgb, cell_slot, slot_cell, recs, counters, err_flags);
    link_body<false, 256>(blk, fp.P, cell_slot, head, recs, counters, log, log_base, frame_key, nullptr, fp.P_frame, r, err_flags);
}

// wave per sampled point; only owners work.  CH = number of 256-float chunks kept in registers (D <= 256*CH).
// The owner sample s0 is the wave's own index and the TAIL of the LIFO list, so its record is fetched (scalar loads, the
// index is wave-uniform) together with the owner flag, and its feature row, the list head and the accumulator row go out in
// the next round trip; the rest of the list (1.3 samples per group on average) is walked from the head down to s0.
// FULL: D == 256 CH exactly (D = 512 at CH = 2, the LSeg width): every row access is an unconditional 16-byte vector access.
// With a run-time width the per-lane "does my piece of the row exist" tests are branches around the loads, and the compiler's
// s_waitcnt bookkeeping falls back to vmcnt(0) where they merge: the accumulator row, the owner's feature row and every
// member's row then arrive one after the other instead of together.
template <int CH, bool FULL>
__device__ __forceinline__ void fuse_group_impl(int s0, int32_t slot, double alpha0, int32_t fpix0, uint32_t rgb0, int P, int D_rt, unsigned long long frame_key, const BatchEntry* __restrict__ batch,
                                                int P_frame, const Recs& recs, int32_t* __restrict__ head, const float* __restrict__ feat,
                                                double* __restrict__ sum_feat, double* __restrict__ sum_w4,
                                                float* __restrict__ first_feat, double* __restrict__ first_alpha,
                                                unsigned long long* __restrict__ slot_key, uint8_t* __restrict__ dirty) {
    const int D = FULL ? 256 * CH : D_rt;
    const int lane = threadIdx.x & 63;
    AVL_STAMP(pt0);
    AVL_STAMP(pt1);
    const bool is_new = slot_key[slot] == kNoKey;  // born in this launch: accumulators hold nothing yet
    const int h0 = head[slot];
    double* sf = sum_feat + (size_t)slot * D;
    AVL_STAMP(pt2);
#ifdef AVL_PROBE_CHAIN
    int probe_n = 1;
    unsigned long long pt3 = 0, pt4 = 0;
#endif

    double acc[CH][4], old[CH][4];
    float f1[CH][4];
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc[c][e] = 0.0; f1[c][e] = 0.f; old[c][e] = 0.0; }
    double w4 = 0.0, a1 = 0.0, w4_old = 0.0;
    int min_s = INT_MAX;
    // the voxel's accumulator row is requested NOW, together with the first feature row, instead of after the list walk: one
    // memory round trip less on the wave's dependent chain (owner flag -> slot -> head / rows -> write).  Requesting it
    // before slot_key is known (reading the row of a new voxel too, discarding it) was measured slower: 12.5 vs 12.1 us per
    // frame, 260 vs 236 us per 64-frame launch -- the scalar loads of slot_key / head are not what the wave waits for.
    if (!is_new) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int d = c * 256 + lane * 4;
            if (FULL || (d + 3 < D && (D & 1) == 0)) {   // 16-byte aligned rows
                const double2 q0 = *reinterpret_cast<const double2*>(sf + d), q1 = *reinterpret_cast<const double2*>(sf + d + 2);
                old[c][0] = q0.x; old[c][1] = q0.y; old[c][2] = q1.x; old[c][3] = q1.y;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (d + e < D) old[c][e] = sf[d + e];
            }
        }
    }
    if (lane < 4) w4_old = is_new ? 0.0 : sum_w4[(size_t)slot * 4 + lane];

    auto row_of = [&](int cur, int32_t fpix) { return (batch ? batch[cur / P_frame].feat : feat) + (size_t)fpix * D; };
    auto load_row = [&](float (&v)[CH][4], const float* f) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int d = c * 256 + lane * 4;
            if (FULL || (d + 3 < D && (D & 3) == 0)) {
                const float4 q = *reinterpret_cast<const float4*>(f + d);
                v[c][0] = q.x; v[c][1] = q.y; v[c][2] = q.z; v[c][3] = q.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[c][e] = d + e < D ? f[d + e] : 0.f;
            }
        }
    };
    auto add = [&](int cur, double alpha, uint32_t rgbv, const float (&v)[CH][4]) {
        // The first-touch sample of a voxel BORN in this launch is kept OUT of the sum: sum_feat = sum over all OTHER samples of
        // alpha f, and the finalisation adds a1^2 f1 (the reference stores feat * alpha with weight alpha for a new voxel and
        // treats it as a mean afterwards, vlmap_builder.py:166-174).  The earlier form sum(alpha f) - a1 (1 - a1) f1 cancels
        // catastrophically for a voxel touched once from far away (alpha = exp(-r^2 / 1.2) < 1e-13 beyond ~6 m: 1 - a1 rounds to
        // 1 and the row came out as ZERO instead of a1 f1).  Members arrive in ascending sample order (up to 64 per launch), so
        // `first` normally fires once; if a smaller sample index turns up later the previous candidate rejoins the sum.
        const bool first = cur < min_s;  // wave-uniform
        const bool hold = first && is_new;
        const bool rejoin = hold && min_s != INT_MAX;
        const double a_prev = a1;
        if (first) { min_s = cur; a1 = alpha; }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (FULL || c * 256 + lane * 4 < D) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (rejoin) acc[c][e] += a_prev * (double)f1[c][e];
                    if (!hold) acc[c][e] += alpha * (double)v[c][e];
                    if (first) f1[c][e] = v[c][e];
                }
            }
        }
        if (lane < 4) w4 += lane == 0 ? alpha : alpha * (double)((rgbv >> (8 * (lane - 1))) & 0xffu);
    };
    const float* feat0 = row_of(s0, fpix0);
#ifdef AVL_ABL_K3_SINGLE      // timing ablation only (WRONG maps): every group treated as a single-sample group -- what the long groups cost
    if (true) {
#else
    if (h0 == s0) {
#endif
        float v[CH][4];
        load_row(v, feat0);
        add(s0, alpha0, rgb0, v);   // one sample for this voxel in this launch: 70 % of the groups of a single frame
#ifdef AVL_PROBE_CHAIN
        { AVL_STAMP(ptx); pt3 = pt4 = ptx; }
#endif
    } else {
        // Several samples: sum them in ASCENDING SAMPLE ORDER (the reference's order), not in the order in which their atomics
        // happened to arrive -- fp64 addition is not associative, and with the arrival order two runs of the same build could
        // differ in the last bit of a feature (seen in 1 map of 10 000).  Lane i keeps the i-th member met on the walk (the
        // owner, the tail, is member 0) and fetches its record; ranks come from n wave-wide compares; up to 64 members are
        // ordered, what is beyond (all-pixel sampling into coarse cells) is added in list order.
        // The range test and the step caps never fire: a list is a simple chain of at most P samples; they only make sure
        // that no corrupted list can keep a wave (and with it the device) busy forever.
        int my = lane == 0 ? s0 : INT_MAX;
        int n = 1, cur = h0;
        while (cur != s0 && n < 64 && (unsigned)cur < (unsigned)P) {
            if (lane == n) my = cur;
            ++n;
            cur = recs.next[cur];
        }
#ifdef AVL_PROBE_CHAIN
        { AVL_STAMP(ptx); pt3 = ptx; probe_n = n; }
#endif
        double a_l = 0.0;
        int32_t fp_l = 0;
        uint32_t rgb_l = 0;
        if (lane < n) {
            a_l = recs.alpha[my];
            fp_l = recs.fpix[my];
            rgb_l = recs.rgb[my];
        }
#ifdef AVL_PROBE_CHAIN
        { AVL_STAMP(ptx); pt4 = ptx; }
#endif
        int rank = 0;
        for (int j = 0; j < n; ++j) rank += __builtin_amdgcn_readlane(my, j) < my ? 1 : 0;
        const int a_lo = __double2loint(a_l), a_hi = __double2hiint(a_l);
        // The rows of MR members (ranks k0 .. k0 + MR - 1) are requested TOGETHER, and unconditionally: until round 5 every member's
        // row was loaded inside its add -- one HBM round trip (1.3-1.7 us) per member -- and a load inside a branch would not do
        // either: the compiler's s_waitcnt bookkeeping falls back to vmcnt(0) where the branches merge.  Ranks beyond the group
        // re-read the owner's row (a cache hit) and are not added.  Two at a time: 73 % of the multi-sample groups of a frame are
        // pairs, and four in flight cost 16 more registers and 5-9 % in 16- / 64-frame launches, where most groups are large
        // (profiles/r05_ab_builder_k3.txt).
#ifndef AVL_K3_MR
#define AVL_K3_MR 2
#endif
        constexpr int MR = CH <= 4 ? AVL_K3_MR : 1;   // (CH = 6: 219 VGPRs already)
        for (int k0 = 0; k0 < n; k0 += MR) {
            float v[MR][CH][4];
            int cj[MR], lj[MR];
#pragma unroll
            for (int j = 0; j < MR; ++j) {
                const unsigned long long m = __ballot(lane < n && rank == k0 + j);
                lj[j] = m ? __builtin_amdgcn_readfirstlane(__ffsll((long long)m) - 1) : 0;
                cj[j] = m ? __builtin_amdgcn_readlane(my, lj[j]) : -1;
                load_row(v[j], cj[j] >= 0 ? row_of(cj[j], __builtin_amdgcn_readlane(fp_l, lj[j])) : feat0);
            }
#pragma unroll
            for (int j = 0; j < MR; ++j) {
                if (cj[j] >= 0) {
                    const double al = __hiloint2double(__builtin_amdgcn_readlane(a_hi, lj[j]), __builtin_amdgcn_readlane(a_lo, lj[j]));
                    add(cj[j], al, (uint32_t)__builtin_amdgcn_readlane((int)rgb_l, lj[j]), v[j]);
                }
            }
        }
        for (int steps = 0; cur != s0 && (unsigned)cur < (unsigned)P && steps < P; ++steps) {   // members beyond the 64th
            const int nxt = recs.next[cur];
            float v[CH][4];
            load_row(v, row_of(cur, recs.fpix[cur]));
            add(cur, recs.alpha[cur], recs.rgb[cur], v);
            cur = nxt;
        }
    }

    AVL_STAMP(pt5);
    float* ff = first_feat + (size_t)slot * D;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int d = c * 256 + lane * 4;
        double r[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = is_new ? acc[c][e] : old[c][e] + acc[c][e];
        if constexpr (FULL) {
            *reinterpret_cast<double2*>(sf + d) = double2{r[0], r[1]};
            *reinterpret_cast<double2*>(sf + d + 2) = double2{r[2], r[3]};
            if (is_new) *reinterpret_cast<float4*>(ff + d) = float4{f1[c][0], f1[c][1], f1[c][2], f1[c][3]};
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (d + e < D) {
                    sf[d + e] = r[e];
                    if (is_new) ff[d + e] = f1[c][e];
                }
        }
    }
    if (lane < 4) sum_w4[(size_t)slot * 4 + lane] = w4_old + w4;
    if (is_new) {
        if (lane == 0) {
            first_alpha[slot] = a1;
            slot_key[slot] = batch ? (batch[min_s / P_frame].frame_key | (unsigned)(min_s % P_frame)) : (frame_key | (unsigned)min_s);
        }
    }
    if (lane == 0) {
        head[slot] = -1;  // ready for the next launch
        dirty[slot] = 1;  // changed since the last checkpoint (avl_builder_finalize_ex)
    }
#ifdef AVL_PROBE_CHAIN
    {
        AVL_STAMP(pt6);
        if (g_probe && lane == 0 && P < kProbeK12Row0) {
            unsigned long long* q = g_probe + (size_t)s0 * 8;
            q[0] = pt0; q[1] = pt1; q[2] = pt2; q[3] = pt3; q[4] = pt4; q[5] = pt5; q[6] = pt6;
            q[7] = (unsigned long long)probe_n | (is_new ? 1ull << 32 : 0ull);
        }
    }
#endif
}

template <int CH, bool COMPACT = false>
__device__ __forceinline__ void fuse_body(int blk, int P, int D, unsigned long long frame_key, const BatchEntry* __restrict__ batch,
                                          int P_frame, const Recs& recs, int32_t* __restrict__ head, const float* __restrict__ feat,
                                          double* __restrict__ sum_feat, double* __restrict__ sum_w4,
                                          float* __restrict__ first_feat, double* __restrict__ first_alpha,
                                          unsigned long long* __restrict__ slot_key, uint8_t* __restrict__ dirty,
                                          const OwnerList& ol = OwnerList{}) {
    auto group = [&](int s0, int32_t slot, double alpha0, int32_t fpix0, uint32_t rgb0) {
        if (D == 256 * CH)
            fuse_group_impl<CH, true>(s0, slot, alpha0, fpix0, rgb0, P, D, frame_key, batch, P_frame, recs, head, feat, sum_feat, sum_w4, first_feat,
                                      first_alpha, slot_key, dirty);
        else
            fuse_group_impl<CH, false>(s0, slot, alpha0, fpix0, rgb0, P, D, frame_key, batch, P_frame, recs, head, feat, sum_feat, sum_w4, first_feat,
                                       first_alpha, slot_key, dirty);
    };
    if constexpr (!COMPACT) {
        // a single frame (and every launch below kAggregateSamples samples): wave per SAMPLE, the owner flag fetched with the record (72 % of the waves leave here).  The compacted
        // form below measured slower at this size -- pipe_kernel 11.2 -> 12.0 us, fuse_kernel 10.35 -> 10.7 (profiles/r05_ab_builder_k3.txt s27)
        const int s0 = __builtin_amdgcn_readfirstlane((int)((blk * AVL_K3_THREADS + threadIdx.x) >> 6));   // (not blockDim.x: see bp_voxelize_body)
        if (s0 >= P) return;
        const uint8_t own = recs.owner[s0];
        const int32_t slot = recs.slot[s0];
        const double alpha0 = recs.alpha[s0];
        const int32_t fpix0 = recs.fpix[s0];
        const uint32_t rgb0 = recs.rgb[s0];
        if (!own) return;
        group(s0, slot, alpha0, fpix0, rgb0);
        return;
    } else {
    // Batched launches (a kernel of their own, so that the single-frame kernels keep their 80 registers): K3 runs over the owners K2 compacted -- half the workgroups, and +7 % at 64 frames per launch.  Wave i of
    // workgroup j takes owner entry t = i * nblk + j of the launch's nb * kFuseWaves entries (K2 workgroup t / kFuseWaves, its owner
    // t % kFuseWaves): the owners of a K2 workgroup are a PREFIX of its entries and consecutive entries go to different workgroups,
    // so every workgroup -- every CU -- gets the same mix of live and idle waves.
    const int nb = (P + 255) / 256;
    const int nblk = nb * (kFuseWaves / 4);
    const int t = (int)(threadIdx.x >> 6) * nblk + blk;
    const int b = __builtin_amdgcn_readfirstlane(t / kFuseWaves);
    if (b >= nb) return;
    const int n_own = __builtin_amdgcn_readfirstlane(ol.ocnt[b]);
    for (int k = t % kFuseWaves; k < n_own; k += kFuseWaves) {
        const int oi = __builtin_amdgcn_readfirstlane(b * 256 + k);
        // (oi is wave-uniform: scalar registers whatever kind of load the compiler picks)
        const double a0 = ol.o_alpha[oi];
        group(__builtin_amdgcn_readfirstlane(ol.o_s[oi]), __builtin_amdgcn_readfirstlane(ol.o_slot[oi]),
              __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(a0)), __builtin_amdgcn_readfirstlane(__double2loint(a0))),
              __builtin_amdgcn_readfirstlane(ol.o_fpix[oi]), (uint32_t)__builtin_amdgcn_readfirstlane((int)ol.o_rgb[oi]));
    }
    }
}

#ifndef AVL_K3_WPE
#define AVL_K3_WPE 0
#endif
#if AVL_K3_WPE
#define AVL_K3_OCC(CH) __attribute__((amdgpu_waves_per_eu((CH) <= 2 ? AVL_K3_WPE : 1)))
#else
#define AVL_K3_OCC(CH)
#endif

template <int CH, bool COMPACT>
__global__ __launch_bounds__(COMPACT ? 256 : AVL_K3_THREADS) AVL_K3_OCC(CH) void fuse_kernel(int P, int D, unsigned long long frame_key, const BatchEntry* __restrict__ batch,
                                                   int P_frame, Recs recs, int32_t* __restrict__ head, const float* __restrict__ feat,
                                                   double* __restrict__ sum_feat, double* __restrict__ sum_w4,
                                                   float* __restrict__ first_feat, double* __restrict__ first_alpha,
                                                   unsigned long long* __restrict__ slot_key, uint8_t* __restrict__ dirty, OwnerList ol) {
    fuse_body<CH, COMPACT>(blockIdx.x, P, D, frame_key, batch, P_frame, recs, head, feat, sum_feat, sum_w4, first_feat, first_alpha, slot_key, dirty, ol);
}

// Deferred-fuse launch (avl_builder_set_deferred_fuse): ONE kernel per frame.  Workgroups [0, pb) run K1 + K2 of the NEW frame
// (records into recs / lists into head), the others run K3 of the PREVIOUS frame (prev_recs / prev_head, the other halves of
// the double buffers).  The two touch disjoint state: K1/K2 use cell_slot, slot_cell, the slot counter and the new frame's
// buffers; K3 uses the accumulators, slot_key, dirty and the previous frame's buffers.
struct FusePrev {
    int P;                       // 0: nothing pending
    unsigned long long frame_key;
    Recs recs;
    int32_t* head;
    const float* feat;
};

template <int CH>
__global__ __launch_bounds__(AVL_K3_THREADS) AVL_K3_OCC(CH) void pipe_kernel(FrameParams fp, int pb, const float* depth, const int32_t* __restrict__ sample_idx,
                                                   const uint8_t* rgb, int32_t* __restrict__ cell_slot, int32_t* __restrict__ slot_cell,
                                                   Recs recs, int32_t* __restrict__ head, unsigned long long* __restrict__ counters,
                                                   int* __restrict__ err_flags, ReplayLog log, long long log_base,
                                                   unsigned long long frame_key, FusePrev prev, int D, double* __restrict__ sum_feat,
                                                   double* __restrict__ sum_w4, float* __restrict__ first_feat,
                                                   double* __restrict__ first_alpha, unsigned long long* __restrict__ slot_key,
                                                   uint8_t* __restrict__ dirty, FrameParams fpn, PreGather next, int gb) {
    if ((int)blockIdx.x < gb) {                   // (in front of K1 + K2's and K3's workgroups: the stateless half of the NEXT frame's K1)
        bp_voxelize_body<AVL_K3_THREADS, 2>((int)blockIdx.x, fpn, next.depth, next.samples, next.rgb, cell_slot, slot_cell, recs, counters, err_flags,
                                            next.out);
        return;
    }
    const int blk = (int)blockIdx.x - gb;
    if (blk < pb) {
        warm_kernel_arguments<12>();   // (K1 + K2 read FrameParams and a dozen pointers; a K3 wave needs two lines, and touching more
                                       // costs it: fuse_kernel 10.4-10.7 -> 10.9-11.1 us, 64 frames per launch 230 -> 255 us)
        SampleRec r;
        if (fp.pre) r = bp_voxelize_body<AVL_K3_THREADS, 1>(blk, fp, depth, sample_idx, rgb, cell_slot, slot_cell, recs, counters, err_flags);
        else r = bp_voxelize_body<AVL_K3_THREADS, 0>(blk, fp, depth, sample_idx, rgb, cell_slot, slot_cell, recs, counters, err_flags);
        link_body<false, AVL_K3_THREADS>(blk, fp.P, cell_slot, head, recs, counters, log, log_base, frame_key, nullptr, fp.P_frame, r, err_flags);
    } else {
        fuse_body<CH>(blk - pb, prev.P, D, prev.frame_key, nullptr, prev.P, prev.recs, prev.head, prev.feat, sum_feat, sum_w4,
                      first_feat, first_alpha, slot_key, dirty);
    }
}

// generic feature width (D > 1536).  A voxel's samples are summed in ASCENDING SAMPLE ORDER like fuse_group_impl does, so that two runs
// of a build give the same bits: the wave orders up to 64 members at a time (ranks from wave-wide compares, the order parked in LDS)
// and adds them 64 columns x kGenTile column groups at a time from registers; a list of more than 64 members is taken in ROUNDS -- every
// round re-walks it and keeps the 64 smallest sample indices above the previous round's largest -- whose partial sums are added to the
// row one after the other (a fixed association, whatever order the atomics of K2 arrived in).
constexpr int kGenTile = 8;
__global__ __launch_bounds__(256) void fuse_generic_kernel(int P, int D, unsigned long long frame_key,
                                                           const BatchEntry* __restrict__ batch, int P_frame, Recs recs,
                                                           int32_t* __restrict__ head, const float* __restrict__ feat,
                                                           double* __restrict__ sum_feat, double* __restrict__ sum_w4,
                                                           float* __restrict__ first_feat, double* __restrict__ first_alpha,
                                                           unsigned long long* __restrict__ slot_key, uint8_t* __restrict__ dirty) {
    __shared__ int ord_s[4][64];
    const int lane = threadIdx.x & 63;
    int* ord = ord_s[(threadIdx.x >> 6) & 3];
    const int s0 = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (s0 >= P) return;
    if (!recs.owner[s0]) return;
    const int32_t slot = recs.slot[s0];
    const bool is_new = slot_key[slot] == kNoKey;
    const int h0 = head[slot];
    auto wave_max = [&](int v) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v = max(v, __shfl_xor(v, o));
        return v;
    };
    // one walk: the size of the list, its smallest sample (the first touch) and -- all a short list needs -- its first 64 members
    int n_total = 0, min_s = INT_MAX, held = INT_MAX;
    for (int cur = h0; cur >= 0 && (unsigned)cur < (unsigned)P && n_total < P; cur = recs.next[cur]) {
        if (n_total < 64 && lane == n_total) held = cur;
        min_s = min(min_s, cur);
        ++n_total;
    }
    if (n_total == 0) return;      // (a corrupted list: never -- the owner is on its own list)
    const double a1 = recs.alpha[min_s];
    const float* row1 = (batch ? batch[min_s / P_frame].feat : feat) + (size_t)recs.fpix[min_s] * D;
    double* sf = sum_feat + (size_t)slot * D;
    float* ff = first_feat + (size_t)slot * D;
    if (is_new)
        for (int d = lane; d < D; d += 64) ff[d] = row1[d];
    double w4 = 0.0;
    int prev = -1;
    bool first_round = true;
    for (int done = 0; done < n_total;) {
        int m;
        if (n_total <= 64) {
            m = n_total;
        } else {
            // the 64 smallest sample indices above `prev`: lanes fill up, then a smaller newcomer replaces the largest one held
            held = INT_MAX;
            int cnt = 0, curmax = -1, steps = 0;
            for (int cur = h0; cur >= 0 && (unsigned)cur < (unsigned)P && steps < P; cur = recs.next[cur], ++steps) {
                if (cur <= prev) continue;
                if (cnt < 64) {
                    if (lane == cnt) held = cur;
                    if (++cnt == 64) curmax = wave_max(held);
                } else if (cur < curmax) {
                    const unsigned long long at = __ballot(held == curmax);
                    if (lane == __ffsll((long long)at) - 1) held = cur;
                    curmax = wave_max(held);
                }
            }
            m = cnt;
            if (m == 0) break;     // (a corrupted list: never)
        }
        int rank = 0;
        for (int j = 0; j < m; ++j) rank += __builtin_amdgcn_readlane(held, j) < held ? 1 : 0;
        if (lane < m) ord[rank] = held;
        __builtin_amdgcn_wave_barrier();
        prev = ord[m - 1];
        for (int k = 0; k < m; ++k) {
            const int cur = ord[k];
            const double alpha = recs.alpha[cur];
            const uint32_t rgbv = recs.rgb[cur];
            if (lane < 4) w4 += lane == 0 ? alpha : alpha * (double)((rgbv >> (8 * (lane - 1))) & 0xffu);
        }
        for (int d0 = 0; d0 < D; d0 += 64 * kGenTile) {
            double acc[kGenTile];
#pragma unroll
            for (int t = 0; t < kGenTile; ++t) acc[t] = 0.0;
            for (int k = 0; k < m; ++k) {
                const int cur = ord[k];
                if (is_new && cur == min_s) continue;            // the first touch of a new voxel stays out of the sum (fuse_group_impl)
                const double alpha = recs.alpha[cur];
                const float* row = (batch ? batch[cur / P_frame].feat : feat) + (size_t)recs.fpix[cur] * D;
#pragma unroll
                for (int t = 0; t < kGenTile; ++t) {
                    const int d = d0 + t * 64 + lane;
                    if (d < D) acc[t] += alpha * (double)row[d];
                }
            }
#pragma unroll
            for (int t = 0; t < kGenTile; ++t) {
                const int d = d0 + t * 64 + lane;
                if (d < D) sf[d] = (is_new && first_round) ? acc[t] : sf[d] + acc[t];
            }
        }
        __builtin_amdgcn_wave_barrier();
        done += m;
        first_round = false;
    }
    if (lane < 4) {
        double* w = sum_w4 + (size_t)slot * 4 + lane;
        *w = is_new ? w4 : *w + w4;
    }
    if (lane == 0) {
        if (is_new) {
            first_alpha[slot] = a1;
            slot_key[slot] = batch ? (batch[min_s / P_frame].frame_key | (unsigned)(min_s % P_frame)) : (frame_key | (unsigned)min_s);
        }
        head[slot] = -1;
        dirty[slot] = 1;
    }
}

// wave per output row r; the accumulators of row r live in slot perm[r] (perm == nullptr: identity).  sum_feat rows have
// stride ld_sf, the [sum alpha, sum alpha*rgb] quadruple of a row sits at sum_w4 + row * ld_w4.  first_feat == nullptr: the
// first-touch correction has already been folded into sum_feat (merged accumulators, scatter_merge_kernel).  Output row r
// is voxel id row0 + r (row0 != 0: one rank finalises one block of a reduce-scattered map).
__global__ __launch_bounds__(256) void finalize_kernel(int64_t n, int D, int gs, int vh, int64_t row0, const int32_t* __restrict__ perm,
                                                       const int32_t* __restrict__ cell, const double* __restrict__ sum_feat,
                                                       int64_t ld_sf, const double* __restrict__ sum_w4, int64_t ld_w4,
                                                       const float* __restrict__ first_feat, const double* __restrict__ first_alpha,
                                                       float* __restrict__ grid_feat, int32_t* __restrict__ grid_pos,
                                                       float* __restrict__ weight, uint8_t* __restrict__ grid_rgb,
                                                       int32_t* __restrict__ occupied) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r = wave0; r < n; r += nwaves) {
        const int64_t sl = perm ? perm[r] : r;
        const double w = sum_w4[sl * ld_w4];
        if (grid_feat) {
            const double* s = sum_feat + sl * ld_sf;
            float* o = grid_feat + r * D;
            if (first_feat) {
                // reference closed form (a1^2 f1 + sum_{i >= 2} alpha_i f_i) / sum alpha; sum_feat excludes the first touch
                const double a1 = first_alpha[sl];
                const double a1sq = a1 * a1;
                const float* f1 = first_feat + sl * D;
                for (int c = lane; c < D; c += 64) o[c] = (float)((a1sq * (double)f1[c] + s[c]) / w);
            } else {
                for (int c = lane; c < D; c += 64) o[c] = (float)(s[c] / w);
            }
        }
        if (lane == 0) {
            const int32_t cl = cell[sl];
            if (grid_pos) {
                grid_pos[r * 3 + 0] = cl / (gs * vh);
                grid_pos[r * 3 + 1] = (cl / vh) % gs;
                grid_pos[r * 3 + 2] = cl % vh;
            }
            if (weight) weight[r] = (float)w;
            if (occupied) occupied[cl] = (int32_t)(row0 + r);
        }
        if (grid_rgb && lane < 3) {
            // running mean stored into a uint8 array (truncating cast); we truncate the exact weighted mean.  (sum alpha c) / (sum
            // alpha) of samples that all have the colour c is c or c - 1 ulp: the 1e-9 keeps that from truncating to c - 1 (a voxel
            // touched once stores its pixel's colour exactly, vlmap_builder.py:167)
            double m = sum_w4[sl * ld_w4 + 1 + lane] / w + 1e-9;
            m = fmin(fmax(m, 0.0), 255.0);
            grid_rgb[r * 3 + lane] = (uint8_t)m;
        }
    }
}

// Multi-GPU merge, step "scatter" (avlmaps_amd/parallel.py): wave per local slot s.  The slot's accumulators go to row
// row_of_slot[s] of the dense (M, D + 4) float64 buffer every rank reduces -- straight from the builder's own arrays, no
// export copy.  The rank that OWNS the voxel's global first touch (its slot_key equals the all-reduced MIN key) subtracts the
// first touch with the reference's weight a1^2 (every other rank: a1; vlmap_builder.py:166-174 closed form), so that the
// reduced rows only need dividing by sum alpha: ONE sum-reduce carries the whole merge.
__global__ __launch_bounds__(256) void scatter_merge_kernel(int64_t n, int D, const int64_t* __restrict__ row_of_slot,
                                                            const unsigned long long* __restrict__ global_key,
                                                            const unsigned long long* __restrict__ slot_key,
                                                            const double* __restrict__ sum_feat, const double* __restrict__ sum_w4,
                                                            const float* __restrict__ first_feat, const double* __restrict__ first_alpha,
                                                            double* __restrict__ acc, int64_t ld) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t s = wave0; s < n; s += nwaves) {
        const int64_t row = row_of_slot[s];
        const bool owner = slot_key[s] == global_key[row];
        // sum_feat leaves the slot's LOCAL first touch out (fuse_body): the rank that holds the GLOBAL first touch contributes it
        // with the reference's a1^2, every other rank with its plain weight a1
        const double a1 = first_alpha[s];
        const double wf = owner ? a1 * a1 : a1;
        const double* sf = sum_feat + s * D;
        const float* f1 = first_feat + s * D;
        double* o = acc + row * ld;
        for (int c = lane; c < D; c += 64) o[c] = wf * (double)f1[c] + sf[c];
        if (lane < 4) o[D + lane] = sum_w4[s * 4 + lane];
    }
}

// Row-sharded merge with the mixed payload (avlmaps_amd/parallel.py, round 4).  A voxel that only ONE rank ever touched needs no
// float64 exchange: its finished float32 feature row (a1^2 f1 + sum) / sum alpha is computed where the accumulators live --
// the same float64 expression finalize_kernel evaluates, so the row is bit-identical to the single-process map -- and travels as
// 4 B per element.  Only voxels that several ranks touched ship float64 partial sums (own != 0: this rank holds the global first
// touch and folds the reference's first-touch term in).  Wave per listed slot; output row i belongs to slot slots[i].
__global__ __launch_bounds__(256) void export_rows_f32_kernel(int64_t k, int D, const int32_t* __restrict__ slots,
                                                              const double* __restrict__ sum_feat, const double* __restrict__ sum_w4,
                                                              const float* __restrict__ first_feat, const double* __restrict__ first_alpha,
                                                              float* __restrict__ out, int64_t ld) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t i = wave0; i < k; i += nwaves) {
        const int64_t sl = slots[i];
        const double w = sum_w4[sl * 4];
        const double a1 = first_alpha[sl];
        const double a1sq = a1 * a1;
        const double* s = sum_feat + sl * D;
        const float* f1 = first_feat + sl * D;
        float* o = out + i * ld;
        for (int c = lane; c < D; c += 64) o[c] = (float)((a1sq * (double)f1[c] + s[c]) / w);   // finalize_kernel's expression
    }
}

__global__ __launch_bounds__(256) void export_rows_f64_kernel(int64_t k, int D, const int32_t* __restrict__ slots,
                                                              const uint8_t* __restrict__ own, const double* __restrict__ sum_feat,
                                                              const float* __restrict__ first_feat, const double* __restrict__ first_alpha,
                                                              double* __restrict__ out, int64_t ld) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t i = wave0; i < k; i += nwaves) {
        const int64_t sl = slots[i];
        const double a1 = first_alpha[sl];
        const double wf = own[i] ? a1 * a1 : a1;   // global first touch: a1^2 f1 (reference closed form); else the sample's plain weight
        const double* s = sum_feat + sl * D;
        const float* f1 = first_feat + sl * D;
        double* o = out + i * ld;
        for (int c = lane; c < D; c += 64) o[c] = wf * (double)f1[c] + s[c];
    }
}

// Sender side of the gather-plan merge (avl_merge2.hip, avlmaps_amd/merge2.py): the rank's voxels in final-row order, straight from
// the accumulators into the send buffer of the ONE payload all_to_all.  Destination q's segment = [side records | finished float32
// rows | float64 partial rows].  Wave per voxel i: its 64-byte side record [row - first row of q's block | index in q's done / part
// list << 32 | flags, sum_w4 (4 x f64), 3 words of replay state (avl_merge2_side_state fills them after the replay)], and its
// feature row: export_rows_f32_kernel's expression for a voxel of this rank alone (straight into this rank's own block when it owns
// the row), export_rows_f64_kernel's for a voxel several ranks touched.
struct M2PackSeg {
    long long cum[65];        // cum[q] = voxels of this call for ranks < q (cum[ws] = all of them): wave w serves rank q with cum[q] <= w < cum[q + 1]
    long long lo[64];         // ... and is voxel lo[q] + (w - cum[q]) of the rank's final-row order
    long long dlo[64];        // single-rank voxels of that order before lo[q]
    long long row0[64];       // first final row this call covers at rank q (a chunk of q's block): side records carry row - row0[q]
    long long side_off[64], done_off[64], part_off[64];   // word (8 B) offsets of q's three lists in the send buffer
};

__global__ __launch_bounds__(256) void m2_pack_kernel(long long n, int ws, int rank, int D, long long own_r0, M2PackSeg sg,
                                                      const int32_t* __restrict__ order, const int32_t* __restrict__ row_s,
                                                      const int32_t* __restrict__ prev_s, const int32_t* __restrict__ next_s,
                                                      const int32_t* __restrict__ sidx, const double* __restrict__ sum_feat,
                                                      const double* __restrict__ sum_w4, const float* __restrict__ first_feat,
                                                      const double* __restrict__ first_alpha, long long* __restrict__ send,
                                                      float* __restrict__ own_feat) {
    const int lane = threadIdx.x & 63;
    const long long wave0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    const long long ldf = (D + 1) / 2 * 2;
    for (long long w = wave0; w < n; w += nwaves) {
        int q = 0;
        while (q + 1 < ws && w >= sg.cum[q + 1]) ++q;
        const long long j = w - sg.cum[q];
        const long long i = sg.lo[q] + j;
        const long long sl = order[i];
        const bool is_new = prev_s[sl] < 0, single = is_new && next_s[sl] < 0;
        const long long didx = (long long)sidx[i] - sg.dlo[q], pidx = j - didx;
        const bool direct = single && q == rank && own_feat != nullptr;
        const long long row = row_s[sl];
        const long long row_rel = row - sg.row0[q];
        const double a1 = first_alpha[sl];
        const double* s = sum_feat + sl * D;
        const float* f1 = first_feat + sl * D;
        if (single) {
            const double wsum = sum_w4[sl * 4];
            const double a1sq = a1 * a1;
            float* o = direct ? own_feat + (row - own_r0) * D : reinterpret_cast<float*>(send + sg.done_off[q]) + didx * ldf;
            for (int c = lane; c < D; c += 64) o[c] = (float)((a1sq * (double)f1[c] + s[c]) / wsum);     // finalize_kernel's expression
        } else {
            const double wf = is_new ? a1 * a1 : a1;   // global first touch: a1^2 f1 (reference closed form); else the sample's plain weight
            double* o = reinterpret_cast<double*>(send + sg.part_off[q]) + pidx * D;
            for (int c = lane; c < D; c += 64) o[c] = wf * (double)f1[c] + s[c];
        }
        long long* rec = send + sg.side_off[q] + 8 * j;
        if (lane == 0)
            rec[0] = (long long)((unsigned long long)row_rel | ((unsigned long long)(single ? didx : pidx) << 32) |
                                 (single ? (1ull << 63) : 0ull) | (direct ? (1ull << 62) : 0ull));
        else if (lane < 5)
            rec[lane] = __double_as_longlong(sum_w4[sl * 4 + lane - 1]);
        else if (lane < 8)
            rec[lane] = 0;
    }
}

// grid_pos / weight / grid_rgb / occupied_ids of n merged rows from their cells and [sum alpha, sum alpha rgb] quadruples alone
// (the feature rows of the mixed payload are finished elsewhere); same arithmetic as finalize_kernel's lane-0 part
__global__ __launch_bounds__(256) void finalize_side_kernel(int64_t n, int gs, int vh, int64_t row0, const int32_t* __restrict__ cell,
                                                            const double* __restrict__ w4, int32_t* __restrict__ grid_pos,
                                                            float* __restrict__ weight, uint8_t* __restrict__ grid_rgb,
                                                            int32_t* __restrict__ occupied) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x) {
        const int32_t cl = cell[r];
        const double w = w4[r * 4];
        if (grid_pos) {
            grid_pos[r * 3 + 0] = cl / (gs * vh);
            grid_pos[r * 3 + 1] = (cl / vh) % gs;
            grid_pos[r * 3 + 2] = cl % vh;
        }
        if (weight) weight[r] = (float)w;
        if (occupied) occupied[cl] = (int32_t)(row0 + r);
        if (grid_rgb)
            for (int k = 0; k < 3; ++k) {
                double m = w4[r * 4 + 1 + k] / w + 1e-9;   // as finalize_kernel
                m = fmin(fmax(m, 0.0), 255.0);
                grid_rgb[r * 3 + k] = (uint8_t)m;
            }
    }
}

// first / one-past-last position of every slot's run in the slot-sorted log
__global__ void log_segments_kernel(const uint32_t* __restrict__ sorted_slot, long long L, long long nslots,
                                    long long* __restrict__ seg_start, long long* __restrict__ seg_end) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < L; i += (long long)gridDim.x * blockDim.x) {
        const uint32_t sl = sorted_slot[i];
        if (sl >= (uint32_t)nslots) continue;
        if (i == 0 || sorted_slot[i - 1] != sl) seg_start[sl] = i;
        if (i == L - 1 || sorted_slot[i + 1] != sl) seg_end[sl] = i + 1;
    }
}

// One update of the reference's running weight / colour with the reference's dtypes (vlmap_builder.py:164-178; NumPy >= 2
// promotion, see oracle/avl_oracle.c avlo_integrate_frame):
//   until the first capacity doubling (_reserve_map_space, :286-311) weight is float32 and grid_rgb uint8 (truncating
//   store at every update); afterwards weight is float64 and grid_rgb float32.  The doubling happens right after the
//   voxel with id gs*gs - 1 was created, i.e. for every update whose key is greater than that voxel's first-touch key
//   (`grown`).  c[] holds uint8 or float32 values exactly.
__device__ __forceinline__ void replay_step(double& w, double (&c)[3], bool& started, double alpha, uint32_t rgbv, bool grown) {
    const double v[3] = {(double)(rgbv & 0xffu), (double)((rgbv >> 8) & 0xffu), (double)((rgbv >> 16) & 0xffu)};
    if (!started) {
        started = true;
        for (int k = 0; k < 3; ++k) c[k] = v[k];
        const double ww = 0.0 + alpha;
        w = grown ? ww : (double)(float)ww;
    } else {
        const double denom = w + alpha;
        if (!grown) {
            const float wf = (float)w;
            for (int k = 0; k < 3; ++k) {
                const float prod = (float)c[k] * wf;
                const double q = ((double)prod + v[k] * alpha) / denom;
                c[k] = (double)(uint8_t)q;
            }
            w = (double)(float)denom;
        } else {
            for (int k = 0; k < 3; ++k) c[k] = (double)(float)((c[k] * w + v[k] * alpha) / denom);
            w = denom;
        }
    }
}

// The updates [i0, i1) of one voxel, in order.  The state is a serial chain, the loads are not: kReplayAhead entries' index ->
// {alpha, rgb, key} gathers are requested together (unconditionally: positions past the end re-read the last entry), so a long
// segment pays one memory round trip per kReplayAhead entries instead of two per entry.
#ifndef AVL_REPLAY_AHEAD
#define AVL_REPLAY_AHEAD 4
#endif
constexpr int kReplayAhead = AVL_REPLAY_AHEAD;
__device__ __forceinline__ void replay_walk(double& w, double (&c)[3], bool& started, long long i0, long long i1, const int32_t* __restrict__ order,
                                            const ReplayLog& log, unsigned long long gkey) {
    for (long long i = i0; i < i1; i += kReplayAhead) {
        int32_t e[kReplayAhead];
#pragma unroll
        for (int k = 0; k < kReplayAhead; ++k) e[k] = order[i + k < i1 ? i + k : i1 - 1];
        double a[kReplayAhead];
        uint32_t v[kReplayAhead];
        unsigned long long ky[kReplayAhead];
#pragma unroll
        for (int k = 0; k < kReplayAhead; ++k) {
            using u64x2 = __attribute__((ext_vector_type(2))) unsigned long long;
            const u64x2* r = reinterpret_cast<const u64x2*>(log.rec + e[k]);
            const u64x2 r0 = r[0];
            a[k] = __longlong_as_double((long long)r0.x);
            ky[k] = r0.y;
            v[k] = (uint32_t)r[1].x;
        }
#pragma unroll
        for (int k = 0; k < kReplayAhead; ++k)
            if (i + k < i1) replay_step(w, c, started, a[k], v[k], ky[k] > gkey);
    }
}

// Thread per output row: replay the voxel's updates in the reference's order (the log is in key order, `order` is its stable
// sort by slot).
__global__ __launch_bounds__(256) void replay_rgb_kernel(int64_t n, long long gs2, const int32_t* __restrict__ perm,
                                                         const unsigned long long* __restrict__ keys_sorted,
                                                         const int32_t* __restrict__ order, const long long* __restrict__ seg_start,
                                                         const long long* __restrict__ seg_end, ReplayLog log,
                                                         float* __restrict__ weight, uint8_t* __restrict__ grid_rgb) {
    const unsigned long long gkey = n >= gs2 ? keys_sorted[gs2 - 1] : kNoKey;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x) {
        const int32_t sl = perm[r];
        double w = 0.0, c[3] = {0.0, 0.0, 0.0};
        bool started = false;
        replay_walk(w, c, started, seg_start[sl], seg_end[sl], order, log, gkey);
        if (started) {
            if (weight) weight[r] = (float)w;
            if (grid_rgb)
                for (int k = 0; k < 3; ++k) grid_rgb[r * 3 + k] = (uint8_t)fmin(fmax(c[k], 0.0), 255.0);
        }
    }
}

// Multi-GPU: the sequential replay is a CHAIN over ranks (frames are sharded contiguously, so every update of rank r comes
// before every update of rank r + 1): a rank receives the per-voxel state left by its predecessors, continues it with its own
// log and passes it on -- 24 bytes per voxel per hop instead of shipping the logs (avlmaps_amd/parallel.py).
struct ReplayState {
    double w;
    float c[3];
    uint32_t started;
};
static_assert(sizeof(ReplayState) == 24, "ReplayState is exchanged between ranks as 3 x int64");

__global__ __launch_bounds__(256) void replay_chain_kernel(int64_t n, unsigned long long gkey, const int64_t* __restrict__ row_of_slot,
                                                           const int32_t* __restrict__ order, const long long* __restrict__ seg_start,
                                                           const long long* __restrict__ seg_end, ReplayLog log,
                                                           ReplayState* __restrict__ state) {
    for (int64_t sl = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; sl < n; sl += (int64_t)gridDim.x * blockDim.x) {
        const int64_t si = row_of_slot[sl];
        if (si < 0 || seg_start[sl] >= seg_end[sl]) continue;          // negative index: slot not part of this call
        ReplayState& st = state[si];
        double w = st.w, c[3] = {(double)st.c[0], (double)st.c[1], (double)st.c[2]};
        bool started = st.started != 0;
        replay_walk(w, c, started, seg_start[sl], seg_end[sl], order, log, gkey);
        st.w = w;
        for (int k = 0; k < 3; ++k) st.c[k] = (float)c[k];
        st.started = started ? 1u : 0u;
    }
}

__global__ __launch_bounds__(256) void replay_apply_kernel(int64_t n, const ReplayState* __restrict__ state, float* __restrict__ weight,
                                                           uint8_t* __restrict__ grid_rgb) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x) {
        const ReplayState st = state[r];
        if (!st.started) continue;
        if (weight) weight[r] = (float)st.w;
        if (grid_rgb)
            for (int k = 0; k < 3; ++k) grid_rgb[r * 3 + k] = (uint8_t)fminf(fmaxf(st.c[k], 0.f), 255.f);
    }
}

// row_dirty[r] = the voxel in output row r was fused since the flags were last cleared
__global__ void row_dirty_kernel(int64_t n, const int32_t* __restrict__ perm, uint8_t* __restrict__ dirty, uint8_t* __restrict__ row_dirty,
                                 int clear) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x) {
        const int32_t sl = perm[r];
        row_dirty[r] = dirty[sl];
        if (clear) dirty[sl] = 0;
    }
}

__global__ void iota_kernel(int32_t* __restrict__ v, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) v[i] = (int32_t)i;
}

// wave per imported voxel row: rebuild accumulators from a finalised map (resume, vlmap_builder.py:212-222)
__global__ __launch_bounds__(256) void import_map_kernel(int64_t n, int D, int n0, int gs, int vh, const float* __restrict__ grid_feat,
                                                         const int32_t* __restrict__ grid_pos, const float* __restrict__ weight,
                                                         const uint8_t* __restrict__ grid_rgb, int32_t* __restrict__ cell_slot,
                                                         int32_t* __restrict__ slot_cell, unsigned long long* __restrict__ slot_key,
                                                         double* __restrict__ sum_feat, double* __restrict__ sum_w4,
                                                         float* __restrict__ first_feat, double* __restrict__ first_alpha,
                                                         int* __restrict__ err_flags) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r = wave0; r < n; r += nwaves) {
        const double w = (double)weight[r];
        for (int c = lane; c < D; c += 64) {
            sum_feat[r * D + c] = (double)grid_feat[r * D + c] * w;   // first-touch weighting is already baked in
            first_feat[r * D + c] = 0.f;
        }
        if (lane == 0) {
            const int row = grid_pos[r * 3], col = grid_pos[r * 3 + 1], h = grid_pos[r * 3 + 2];
            if (row < 0 || row >= n0 || col < 0 || col >= gs || h < 0 || h >= vh) {
                atomicOr(err_flags, 4);
            } else {
                const int32_t cell = (row * gs + col) * vh + h;
                cell_slot[cell] = (int32_t)r;
                slot_cell[r] = cell;
            }
            slot_key[r] = (unsigned long long)r;                        // imported voxels order before any new one
            first_alpha[r] = 0.0;                                       // no first-touch term: it is baked into the imported row
            sum_w4[r * 4] = w;
            for (int c = 0; c < 3; ++c) sum_w4[r * 4 + 1 + c] = (grid_rgb ? (double)grid_rgb[r * 3 + c] : 0.0) * w;
        }
    }
}

// pass 1 of the global (multi-floor) builder: bounding box of the transformed sampled points
// (vlmap_builder_multi_floor.py:97-118).  minmax = [min xyz, max xyz] as order-preserving uint64 keys of the doubles.
__device__ __forceinline__ unsigned long long f64_key(double v) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
}

__global__ __launch_bounds__(256) void bbox_kernel(FrameParams fp, const float* __restrict__ depth,
                                                   const int32_t* __restrict__ sample_idx, unsigned long long* __restrict__ minmax) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= fp.P) return;
    const int pix = sample_idx[s];
    if (pix < 0 || pix >= fp.H * fp.W) return;
    const double x = (double)(pix % fp.W) + 0.5, y = (double)(pix / fp.W) + 0.5;
    const double z = fp.depth_u16 ? (double)reinterpret_cast<const uint16_t*>(depth)[pix] / fp.depth_div : (double)depth[pix];
    const double pl0 = fma(fp.kinv[2], 1.0, fma(fp.kinv[1], y, fp.kinv[0] * x)) * z;
    const double pl1 = fma(fp.kinv[5], 1.0, fma(fp.kinv[4], y, fp.kinv[3] * x)) * z;
    const double pl2 = fma(fp.kinv[8], 1.0, fma(fp.kinv[7], y, fp.kinv[6] * x)) * z;
    if (!((pl2 > fp.min_depth) && (pl2 < fp.max_depth))) return;
    const double g[3] = {fma(fp.t[3], 1.0, fma(fp.t[2], pl2, fma(fp.t[1], pl1, fp.t[0] * pl0))),
                         fma(fp.t[7], 1.0, fma(fp.t[6], pl2, fma(fp.t[5], pl1, fp.t[4] * pl0))),
                         fma(fp.t[11], 1.0, fma(fp.t[10], pl2, fma(fp.t[9], pl1, fp.t[8] * pl0)))};
    for (int c = 0; c < 3; ++c) {
        const unsigned long long k = f64_key(g[c]);
        atomicMin(&minmax[c], k);
        atomicMax(&minmax[3 + c], k);
    }
}

}  // namespace avl

using namespace avl;

struct LogSegments;
struct avl_builder {
    int n0, gs, vh, D;   // grid n0 x gs x vh (n0 == gs for the square mobile-base map)
    double cs;
    int64_t capacity;
    size_t ncell;
    int32_t* cell_slot = nullptr;
    int32_t* slot_cell = nullptr;
    unsigned long long* slot_key = nullptr;
    double* sum_feat = nullptr;
    double* sum_w4 = nullptr;
    float* first_feat = nullptr;
    double* first_alpha = nullptr;
    int32_t* head = nullptr;                 // lists of the launch being linked
    int32_t* head_alt = nullptr;             // deferred fuse: lists of the frame whose K3 is still pending (the two swap per frame)
    uint8_t* dirty = nullptr;                // slot fused since the last clearing finalize (incremental checkpoints)
    unsigned long long* counters = nullptr;  // [0] slots handed out, [1] samples fused, [2] per-frame voxel groups fused
    int* err_flags = nullptr;
    char* recs_mem = nullptr;
    OwnerList owners{};                      // batched launches only (see struct OwnerList)
    Recs recs{}, recs_alt{};                 // recs_alt: records of the pending frame (deferred fuse), swapped like head
    int recs_cap = 0;
    // deferred fuse (avl_builder_set_deferred_fuse): K3 of a frame runs inside the NEXT frame's launch
    int deferred = 0;
    struct Pending {
        int P = 0;                           // 0: nothing pending
        unsigned long long frame_key = 0;
        const float* feat = nullptr;
    } pend;
    unsigned long long key_bias = 0;  // set after import_map so that imported voxels order before new ones
    ReplayLog log{};
    long long log_cap = 0, log_used = 0;
    char* rs_mem = nullptr;     // scratch of the log's slot-sorted form (LogSegments), allocated WITH the log: the first finalisation /
    size_t rs_bytes = 0;        // merge of a build does not grow a pool by a GB inside its timed path (22 ms at 78 M samples)
    size_t rs_tmp_bytes = 0;
    BatchEntry* d_table = nullptr;
    int table_cap = 0;
    int64_t vox_bound = 0;       // host-side upper bound on the voxel counter (every fused sample may create one voxel)
    int64_t max_capacity = 0;    // 0: the capacity is fixed; else the accumulators double up to this many voxels
    // the slot-sorted replay log of the last avl_builder_replay_chain call: the round-4 merge calls it twice per merge (voxels that
    // depend on no other rank, then the ones whose predecessor's state had to arrive first) and sorts the log once
    LogSegments* ls_cache = nullptr;
    long long ls_log_used = -1;
    int64_t ls_n = -1;

    // next frame's stateless half of K1 in the C frame loop (PreGather): two buffers of recs_cap records, the one K1 reads and the one
    // being written
    PreRec* pre_buf[2] = {nullptr, nullptr};
    struct PreHeld {           // what pre_buf[buf] holds: the frame with exactly these inputs and parameters
        FrameParams fp{};
        const int32_t* samples = nullptr;
        const void* depth = nullptr;
        const uint8_t* rgb = nullptr;
        int buf = 0;
        bool valid = false;
    } pre_held;
    struct PreNext {           // set by avl_builder_integrate_frames for the launch being issued: the frame after it
        const int32_t* samples = nullptr;
        const float* depth = nullptr;
        const uint8_t* rgb = nullptr;
        const double* h_pc_transform = nullptr;
    } pre_next;
};

// the parameters the stateless half of K1 reads (everything of FrameParams up to the grid shape, the mode; not pre / capacity / batch)
static bool same_geometry(const FrameParams& a, const FrameParams& b) {
    return std::memcmp(&a, &b, offsetof(FrameParams, batch)) == 0 && a.P_frame == b.P_frame && a.mode == b.mode && a.depth_u16 == b.depth_u16;
}

static int builder_check_flags(avl_builder* b, hipStream_t st) {
    int flags = 0;
    AVL_HIP_CHECK(hipMemcpyAsync(&flags, b->err_flags, sizeof(int), hipMemcpyDeviceToHost, st));
    AVL_HIP_CHECK(hipStreamSynchronize(st));
    if (flags & 1) {
        set_error("voxel capacity %lld exhausted: create the builder with a larger capacity", (long long)b->capacity);
        return AVL_ERR_CAPACITY;
    }
    if (flags & 2) {
        set_error("a sampled point projected outside the RGB image (the reference raises IndexError here)");
        return AVL_ERR_INVALID;
    }
    if (flags & 16) {
        set_error("internal error: a voxel under creation was never published to the samples waiting for it (deferred fuse)");
        return AVL_ERR_STATE;
    }
    return AVL_OK;   // bit 8 (global mode: samples outside the pass-1 bounding box were dropped) is informational
}

static int read_counter(avl_builder* b, int which, int64_t* h_n, hipStream_t st) {
    unsigned long long v = 0;
    AVL_HIP_CHECK(hipMemcpyAsync(&v, b->counters + which, sizeof(v), hipMemcpyDeviceToHost, st));
    AVL_HIP_CHECK(hipStreamSynchronize(st));
    *h_n = (int64_t)v;
    return AVL_OK;
}

static int flush_pending(avl_builder* b, hipStream_t st);

static int ensure_recs(avl_builder* b, int P, hipStream_t st) {
    if (P <= b->recs_cap) return AVL_OK;
    int rc = flush_pending(b, st);   // the pending frame's records live in the buffers about to be freed
    if (rc != AVL_OK) return rc;
    AVL_HIP_CHECK(hipStreamSynchronize(st));
    if (b->recs_mem) AVL_HIP_CHECK(hipFree(b->recs_mem));
    b->recs_mem = nullptr;
    const size_t cap = ((size_t)P + P / 4 + 1024 + 255) / 256 * 256;
    const size_t one = (cap * (8 + 4 * 4 + 1) + 255) / 256 * 256;
    const size_t own = (cap * (8 + 4 * 4) + (cap / 256 + 1) * 4 + 255) / 256 * 256;
    const size_t pre = (cap * sizeof(PreRec) + 255) / 256 * 256;
    AVL_HIP_CHECK(hipMalloc((void**)&b->recs_mem, 2 * one + own + 2 * pre));
    auto carve = [&](Recs& r, char* p) {
        r.alpha = reinterpret_cast<double*>(p); p += cap * 8;
        r.slot = reinterpret_cast<int32_t*>(p); p += cap * 4;
        r.fpix = reinterpret_cast<int32_t*>(p); p += cap * 4;
        r.rgb = reinterpret_cast<uint32_t*>(p); p += cap * 4;
        r.next = reinterpret_cast<int32_t*>(p); p += cap * 4;
        r.owner = reinterpret_cast<uint8_t*>(p);
    };
    carve(b->recs, b->recs_mem);
    carve(b->recs_alt, b->recs_mem + one);
    {
        char* p = b->recs_mem + 2 * one;
        OwnerList& o = b->owners;
        o.o_alpha = reinterpret_cast<double*>(p); p += cap * 8;
        o.o_s = reinterpret_cast<int32_t*>(p); p += cap * 4;
        o.o_slot = reinterpret_cast<int32_t*>(p); p += cap * 4;
        o.o_fpix = reinterpret_cast<int32_t*>(p); p += cap * 4;
        o.o_rgb = reinterpret_cast<uint32_t*>(p); p += cap * 4;
        o.ocnt = reinterpret_cast<int32_t*>(p);
    }
    b->pre_buf[0] = reinterpret_cast<PreRec*>(b->recs_mem + 2 * one + own);
    b->pre_buf[1] = reinterpret_cast<PreRec*>(b->recs_mem + 2 * one + own + pre);
    b->pre_held.valid = false;
    b->recs_cap = (int)cap;
    return AVL_OK;
}

// The reference doubles its arrays when max_id reaches their length (_reserve_map_space, vlmap_builder.py:286-311).  Here the
// per-slot arrays are reallocated at (at least) twice the size and copied device-to-device; cell_slot is indexed by cell and
// does not change.  Called between launches only (stream drained first).
static int grow_builder(avl_builder* b, int64_t want, hipStream_t st) {
    // `want` is a worst-case bound (every sample of the next launch creates a voxel): grow as far as allowed; if the map
    // really outgrows max_capacity the kernel's own overflow flag reports it (AVL_ERR_CAPACITY at the next counter read)
    int64_t cap = b->capacity;
    while (cap < want) cap *= 2;
    if (cap > b->max_capacity) cap = b->max_capacity;
    if (cap <= b->capacity) return AVL_OK;
    int rcf = flush_pending(b, st);
    if (rcf != AVL_OK) return rcf;
    AVL_HIP_CHECK(hipStreamSynchronize(st));
    const size_t oc = (size_t)b->capacity, nc = (size_t)cap, D = (size_t)b->D;
    auto regrow = [&](void** p, size_t elem, int fill) -> hipError_t {
        void* q = nullptr;
        hipError_t e = hipMalloc(&q, nc * elem);
        if (e != hipSuccess) return e;
        e = hipMemcpyAsync(q, *p, oc * elem, hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess && fill >= 0) e = hipMemsetAsync(static_cast<char*>(q) + oc * elem, fill, (nc - oc) * elem, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) { (void)hipFree(q); return e; }
        (void)hipFree(*p);
        *p = q;
        return hipSuccess;
    };
    hipError_t e = regrow((void**)&b->slot_cell, sizeof(int32_t), 0xFF);
    if (e == hipSuccess) e = regrow((void**)&b->slot_key, sizeof(unsigned long long), 0xFF);
    if (e == hipSuccess) e = regrow((void**)&b->head, sizeof(int32_t), 0xFF);
    if (e == hipSuccess) e = regrow((void**)&b->head_alt, sizeof(int32_t), 0xFF);
    if (e == hipSuccess) e = regrow((void**)&b->dirty, 1, 0);
    if (e == hipSuccess) e = regrow((void**)&b->sum_feat, D * sizeof(double), -1);
    if (e == hipSuccess) e = regrow((void**)&b->sum_w4, 4 * sizeof(double), -1);
    if (e == hipSuccess) e = regrow((void**)&b->first_feat, D * sizeof(float), -1);
    if (e == hipSuccess) e = regrow((void**)&b->first_alpha, sizeof(double), -1);
    if (e != hipSuccess) {
        set_error("growing the voxel accumulators from %lld to %lld voxels failed: %s", (long long)b->capacity, (long long)cap,
                  hipGetErrorString(e));
        return AVL_ERR_HIP;   // arrays already regrown keep their new size; the handle stays usable at the old capacity
    }
    b->capacity = cap;
    return AVL_OK;
}

// Compaction of the replay log to the entries that updated a voxel (slot != 0xFFFFFFFF), order kept: kLogParts contiguous parts,
// one workgroup each -- count, one-workgroup scan of the counts, then every part writes its survivors' log position and slot behind
// its offset (ballot ranks inside a wave, LDS across the four waves).  rocprim::select with a predicate over a counting iterator
// took 0.92 ms for 78 M entries (0.43 GB of traffic); these three kernels read the slots twice and write 2 x 4 B per survivor.
constexpr int kLogParts = 2048;

__global__ __launch_bounds__(256) void log_count_kernel(const uint32_t* __restrict__ slot, long long L, long long chunk, int* __restrict__ counts) {
    __shared__ int wsum[4];
    const long long lo = (long long)blockIdx.x * chunk, hi = lo + chunk < L ? lo + chunk : L;
    int c = 0;
    for (long long i = lo + threadIdx.x; i < hi; i += 256) c += slot[i] != 0xFFFFFFFFu;
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ __launch_bounds__(1024) void log_offsets_kernel(const int* __restrict__ counts, long long* __restrict__ offsets, long long* __restrict__ total) {
    __shared__ long long part[1024];
    static_assert(kLogParts == 2048, "two parts per thread");
    const int t = threadIdx.x;
    const long long a = counts[2 * t], b = counts[2 * t + 1];
    part[t] = a + b;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const long long v = t >= off ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    const long long before = part[t] - (a + b);
    offsets[2 * t] = before;
    offsets[2 * t + 1] = before + a;
    if (t == 1023) *total = part[t];
}

__global__ __launch_bounds__(256) void log_compact_kernel(const uint32_t* __restrict__ slot, long long L, long long chunk,
                                                          const long long* __restrict__ offsets, int32_t* __restrict__ active,
                                                          uint32_t* __restrict__ active_slot) {
    __shared__ int wsum[4];
    const long long lo = (long long)blockIdx.x * chunk, hi = lo + chunk < L ? lo + chunk : L;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long long base = offsets[blockIdx.x];
    for (long long i0 = lo; i0 < hi; i0 += 256) {                 // (chunk is a multiple of 256: a uniform trip count per workgroup)
        const long long i = i0 + threadIdx.x;
        const uint32_t sl = i < hi ? slot[i] : 0xFFFFFFFFu;
        const bool keep = sl != 0xFFFFFFFFu;
        const unsigned long long m = __ballot(keep);
        const int rank = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __popcll(m);
        __syncthreads();
        int before = 0, all = 0;
        for (int w = 0; w < 4; ++w) {
            before += w < wave ? wsum[w] : 0;
            all += wsum[w];
        }
        if (keep) {
            active[base + before + rank] = (int32_t)i;
            active_slot[base + before + rank] = sl;
        }
        base += all;
        __syncthreads();
    }
}

// The replay log sorted by voxel: `order` = the positions of the entries that updated a voxel, stably sorted by slot (the log is
// in key order, so every voxel's run is in the reference's update order), [seg_start[s], seg_end[s]) = the run of slot s.
// Only ~35 % of the sampled pixels update a voxel (depth mask, feature-image bounds): the log is first compacted to those
// entries, and the radix sort only looks at the bits a slot index can have -- a third of the entries and three of the four
// passes (sorting all 78 M entries of a 10 000-frame log instead: 0.58 ms per pass against 0.22).
struct LogSegments {
    uint32_t *active_slot = nullptr, *sorted_slot = nullptr;
    int32_t *active = nullptr, *order = nullptr;
    long long *seg_start = nullptr, *seg_end = nullptr, *d_count = nullptr;
    void* tmp = nullptr;
    char *block1 = nullptr, *block2 = nullptr;   // TWO pool allocations hold all of the above (a hipMallocAsync / hipFreeAsync pair costs
                                                 // ~90 us of host time: nine of them were most of a small rank's replay)
    static size_t al(size_t bytes) { return (bytes + 255) / 256 * 256; }
    int build(avl_builder* b, int64_t n, hipStream_t st) {
        const long long L = b->log_used;
        const long long chunk = ((L + kLogParts - 1) / kLogParts + 255) / 256 * 256;
        size_t tmp_bytes = 0;
        const size_t b_active = al((size_t)L * sizeof(int32_t)), b_seg = al((size_t)n * sizeof(long long));
        const size_t b_counts = al(kLogParts * sizeof(int)), b_off = al(kLogParts * sizeof(long long));
        // the two L-sized arrays (and, below, the two La-sized ones + the sort's storage) come from the scratch allocated with the log
        // when it is there and large enough; the small per-voxel arrays always from the pool
        const bool own = b->rs_mem && b->rs_bytes >= 4 * b_active + b->rs_tmp_bytes + 256;
        AVL_HIP_CHECK(hipMallocAsync((void**)&block1, (own ? 0 : 2 * b_active) + 256 + 2 * b_seg + b_counts + b_off, st));
        char* p1 = block1;
        if (own) {
            active = reinterpret_cast<int32_t*>(b->rs_mem);
            active_slot = reinterpret_cast<uint32_t*>(b->rs_mem + b_active);
        } else {
            active = reinterpret_cast<int32_t*>(p1);
            active_slot = reinterpret_cast<uint32_t*>(p1 + b_active);
            p1 += 2 * b_active;
        }
        d_count = reinterpret_cast<long long*>(p1);
        seg_start = reinterpret_cast<long long*>(p1 + 256);
        seg_end = reinterpret_cast<long long*>(p1 + 256 + b_seg);
        int* counts = reinterpret_cast<int*>(p1 + 256 + 2 * b_seg);
        long long* offsets = reinterpret_cast<long long*>(p1 + 256 + 2 * b_seg + b_counts);
        AVL_HIP_CHECK(hipMemsetAsync(seg_start, 0, 2 * b_seg, st));
        hipLaunchKernelGGL(log_count_kernel, dim3(kLogParts), dim3(256), 0, st, b->log.slot, L, chunk, counts);
        hipLaunchKernelGGL(log_offsets_kernel, dim3(1), dim3(1024), 0, st, counts, offsets, d_count);
        hipLaunchKernelGGL(log_compact_kernel, dim3(kLogParts), dim3(256), 0, st, b->log.slot, L, chunk, offsets, active, active_slot);
        long long La = 0;
        AVL_HIP_CHECK(hipMemcpyAsync(&La, d_count, sizeof(La), hipMemcpyDeviceToHost, st));
        AVL_HIP_CHECK(hipStreamSynchronize(st));
        const size_t Ls = (size_t)(La > 0 ? La : 1);
        int bits = 1;
        while (bits < 32 && (1ll << bits) <= (long long)n) ++bits;      // slots are < n
        if (La > 0)
            AVL_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, (uint32_t*)nullptr, (uint32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr,
                                                    (size_t)La, 0, bits, st));
        const size_t b_ls = al(Ls * sizeof(uint32_t));
        if (own && tmp_bytes <= b->rs_tmp_bytes) {
            char* p2 = b->rs_mem + 2 * b_active;
            sorted_slot = reinterpret_cast<uint32_t*>(p2);
            order = reinterpret_cast<int32_t*>(p2 + b_ls);
            tmp = p2 + 2 * b_active;                   // (behind the four L-sized words: La <= L)
        } else {
            AVL_HIP_CHECK(hipMallocAsync((void**)&block2, 2 * b_ls + al(tmp_bytes ? tmp_bytes : 16), st));
            sorted_slot = reinterpret_cast<uint32_t*>(block2);
            order = reinterpret_cast<int32_t*>(block2 + b_ls);
            tmp = block2 + 2 * b_ls;
        }
        if (La > 0) {
            AVL_HIP_CHECK(rocprim::radix_sort_pairs(tmp, tmp_bytes, active_slot, sorted_slot, active, order, (size_t)La, 0, bits, st));
            hipLaunchKernelGGL(log_segments_kernel, dim3((unsigned)std::min<long long>((La + 255) / 256, 8192)), dim3(256), 0, st, sorted_slot,
                               La, (long long)n, seg_start, seg_end);
        }
        AVL_HIP_CHECK(hipGetLastError());
        return AVL_OK;
    }
    void release(hipStream_t st) {
        if (block2) (void)hipFreeAsync(block2, st);
        if (block1) (void)hipFreeAsync(block1, st);
        block1 = block2 = nullptr;
    }
};

static void drop_log_segments(avl_builder* b, hipStream_t st) {
    if (!b->ls_cache) return;
    b->ls_cache->release(st);
    delete b->ls_cache;
    b->ls_cache = nullptr;
    b->ls_log_used = -1;
    b->ls_n = -1;
}

// K3 over one launch's records (CH = 256-float register chunks of a feature row)
// `owners_compacted`: the K2 that produced `recs` was voxelize_link_kernel with b->owners (it compacts iff P >= kAggregateSamples);
// pipe_kernel's and voxelize_link_next_kernel's K2 never write the OwnerList, so their K3 is always wave-per-sample
static int launch_fuse(avl_builder* b, int P, unsigned long long frame_key, const BatchEntry* batch, int P_frame, const Recs& recs,
                       int32_t* head, const float* d_feat, hipStream_t st, bool owners_compacted) {
    // batched launches run over the owners K2 compacted; single frames and the generic kernel: wave per sample
    const bool compact = owners_compacted && b->D <= 1536 && P >= kAggregateSamples;
    constexpr int kWaves = AVL_K3_THREADS / 64;
    const unsigned wb = compact ? (unsigned)((P + 255) / 256) * (kFuseWaves / 4) : (unsigned)((P + kWaves - 1) / kWaves);
#define AVL_FUSE_LAUNCH(CH)                                                                                                              \
    do {                                                                                                                                 \
        if (compact)                                                                                                                     \
            hipLaunchKernelGGL((fuse_kernel<CH, true>), dim3(wb), dim3(256), 0, st, P, b->D, frame_key, batch, P_frame, recs, head, d_feat,  \
                               b->sum_feat, b->sum_w4, b->first_feat, b->first_alpha, b->slot_key, b->dirty, b->owners);                 \
        else                                                                                                                             \
            hipLaunchKernelGGL((fuse_kernel<CH, false>), dim3(wb), dim3(AVL_K3_THREADS), 0, st, P, b->D, frame_key, batch, P_frame, recs, head, d_feat, \
                               b->sum_feat, b->sum_w4, b->first_feat, b->first_alpha, b->slot_key, b->dirty, OwnerList{});               \
    } while (0)
    if (b->D <= 256) AVL_FUSE_LAUNCH(1);
    else if (b->D <= 512) AVL_FUSE_LAUNCH(2);
    else if (b->D == 768) AVL_FUSE_LAUNCH(3);      // CLIP ViT-L: three full chunks (the full-width path: unconditional row accesses)
    else if (b->D <= 1024) AVL_FUSE_LAUNCH(4);
    else if (b->D <= 1536) AVL_FUSE_LAUNCH(6);     // a fused visual | audio map (512 + 1024 columns, BASELINE config 5)
    else
        hipLaunchKernelGGL(fuse_generic_kernel, dim3(wb), dim3(256), 0, st, P, b->D, frame_key, batch, P_frame, recs, head, d_feat,
                           b->sum_feat, b->sum_w4, b->first_feat, b->first_alpha, b->slot_key, b->dirty);
#undef AVL_FUSE_LAUNCH
    AVL_HIP_CHECK(hipGetLastError());
    return AVL_OK;
}

// deferred fuse: run the K3 that is still owed (on `st`, which must be the stream the frame was integrated on)
static int flush_pending(avl_builder* b, hipStream_t st) {
    if (b->pend.P == 0) return AVL_OK;
    const int P = b->pend.P;
    b->pend.P = 0;
    return launch_fuse(b, P, b->pend.frame_key, nullptr, P, b->recs_alt, b->head_alt, b->pend.feat, st, /*owners_compacted=*/false);
}

template <int CH>
static void launch_pipe(avl_builder* b, const FrameParams& fp, unsigned pb, const void* d_depth, const int32_t* d_sample_idx,
                        const uint8_t* d_rgb, unsigned long long frame_key, const FusePrev& prev, hipStream_t st, const FrameParams& fpn,
                        const PreGather& next) {
    constexpr int kWaves = AVL_K3_THREADS / 64;
    const unsigned wb = prev.P ? (unsigned)((prev.P + kWaves - 1) / kWaves) : 0u;
    const unsigned gb = next.out ? (unsigned)((fpn.P + AVL_K3_THREADS - 1) / AVL_K3_THREADS) : 0u;
    hipLaunchKernelGGL(pipe_kernel<CH>, dim3(pb + wb + gb), dim3(AVL_K3_THREADS), 0, st, fp, (int)pb, reinterpret_cast<const float*>(d_depth), d_sample_idx,
                       d_rgb, b->cell_slot, b->slot_cell, b->recs, b->head, b->counters, b->err_flags, b->log, b->log_used, frame_key, prev,
                       b->D, b->sum_feat, b->sum_w4, b->first_feat, b->first_alpha, b->slot_key, b->dirty, fpn, next, (int)gb);
}

extern "C" {

#ifdef AVL_PROBE_CHAIN
__attribute__((visibility("default"))) int avl_debug_set_probe(void* d_buf) {
    AVL_HIP_CHECK(hipDeviceSynchronize());
    unsigned long long* p = reinterpret_cast<unsigned long long*>(d_buf);
    AVL_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(avl::g_probe), &p, sizeof(p)));
    return AVL_OK;
}
#endif

int avl_builder_reset(avl_builder* b, void* stream) {
    AVL_REQUIRE(b, "avl_builder_reset: null handle");
    hipStream_t st = as_stream(stream);
    // sum_feat / sum_w4 / first_* need no clearing: a voxel's first fuse is store-only
    AVL_HIP_CHECK(hipMemsetAsync(b->cell_slot, 0xFF, b->ncell * sizeof(int32_t), st));
    AVL_HIP_CHECK(hipMemsetAsync(b->head, 0xFF, (size_t)b->capacity * sizeof(int32_t), st));
    AVL_HIP_CHECK(hipMemsetAsync(b->head_alt, 0xFF, (size_t)b->capacity * sizeof(int32_t), st));
    b->pend = avl_builder::Pending{};   // a pending frame is dropped with the rest of the map
    b->pre_held.valid = false;
    AVL_HIP_CHECK(hipMemsetAsync(b->dirty, 0, (size_t)b->capacity, st));
    AVL_HIP_CHECK(hipMemsetAsync(b->slot_cell, 0xFF, (size_t)b->capacity * sizeof(int32_t), st));
    AVL_HIP_CHECK(hipMemsetAsync(b->slot_key, 0xFF, (size_t)b->capacity * sizeof(unsigned long long), st));
    AVL_HIP_CHECK(hipMemsetAsync(b->counters, 0, 4 * sizeof(unsigned long long), st));
    AVL_HIP_CHECK(hipMemsetAsync(b->err_flags, 0, sizeof(int), st));
    b->key_bias = 0;
    b->log_used = 0;
    b->vox_bound = 0;
    drop_log_segments(b, st);
    return AVL_OK;
}

int avl_builder_destroy(avl_builder* b) {
    if (!b) return AVL_OK;
    drop_log_segments(b, nullptr);
    (void)hipFree(b->cell_slot); (void)hipFree(b->slot_cell); (void)hipFree(b->slot_key); (void)hipFree(b->sum_feat);
    (void)hipFree(b->sum_w4); (void)hipFree(b->first_feat); (void)hipFree(b->first_alpha); (void)hipFree(b->head);
    (void)hipFree(b->head_alt);
    (void)hipFree(b->dirty);
    (void)hipFree(b->counters); (void)hipFree(b->err_flags); (void)hipFree(b->recs_mem);
    (void)hipFree(b->log.slot); (void)hipFree(b->log.rec);
    (void)hipFree(b->rs_mem);
    (void)hipFree(b->d_table);
    delete b;
    return AVL_OK;
}

int avl_builder_create_grid(avl_builder** h_out, int n0, int gs, int vh, double cs, int D, int64_t capacity) {
    AVL_REQUIRE(h_out, "avl_builder_create: null output");
    *h_out = nullptr;
    avl::keep_mempool_once();
    avl_merge2_load();      // (the merge's kernels live in another code object: loaded with the builder, not inside the first merge)
    AVL_REQUIRE(n0 > 0 && gs > 0 && vh > 0 && D > 0 && cs > 0 && capacity > 0, "avl_builder_create: bad parameters");
    const double ncell_d = (double)n0 * gs * vh;
    AVL_REQUIRE(ncell_d < 2.0e9, "avl_builder_create: gs*gs*vh = %.0f cells exceeds the int32 cell index", ncell_d);
    AVL_REQUIRE(capacity < (1ll << 31), "avl_builder_create: capacity must fit int32 voxel ids");
    avl_builder* b = new avl_builder();
    b->n0 = n0; b->gs = gs; b->vh = vh; b->D = D; b->cs = cs; b->capacity = capacity;
    b->ncell = (size_t)n0 * gs * vh;
    hipError_t e = hipSuccess;
    auto alloc = [&](void** p, size_t bytes) { if (e == hipSuccess) e = hipMalloc(p, bytes); };
    alloc((void**)&b->cell_slot, b->ncell * sizeof(int32_t));
    alloc((void**)&b->slot_cell, (size_t)capacity * sizeof(int32_t));
    alloc((void**)&b->slot_key, (size_t)capacity * sizeof(unsigned long long));
    alloc((void**)&b->sum_feat, (size_t)capacity * D * sizeof(double));
    alloc((void**)&b->sum_w4, (size_t)capacity * 4 * sizeof(double));
    alloc((void**)&b->first_feat, (size_t)capacity * D * sizeof(float));
    alloc((void**)&b->first_alpha, (size_t)capacity * sizeof(double));
    alloc((void**)&b->head, (size_t)capacity * sizeof(int32_t));
    alloc((void**)&b->head_alt, (size_t)capacity * sizeof(int32_t));
    alloc((void**)&b->dirty, (size_t)capacity);
    alloc((void**)&b->counters, 4 * sizeof(unsigned long long));
    alloc((void**)&b->err_flags, sizeof(int));
    if (e != hipSuccess) {
        set_error("avl_builder_create: hipMalloc failed: %s (capacity %lld x D %d)", hipGetErrorString(e), (long long)capacity, D);
        avl_builder_destroy(b);
        return AVL_ERR_HIP;
    }
    int rc = avl_builder_reset(b, nullptr);
    if (rc == AVL_OK && hipDeviceSynchronize() != hipSuccess) rc = AVL_ERR_HIP;
    if (rc != AVL_OK) {
        avl_builder_destroy(b);
        return rc;
    }
    *h_out = b;
    return AVL_OK;
}

int avl_builder_enable_replay_log(avl_builder* b, int64_t max_samples) {
    AVL_REQUIRE(b && max_samples > 0, "avl_builder_enable_replay_log: bad arguments");
    AVL_REQUIRE(max_samples < (1ll << 31), "avl_builder_enable_replay_log: at most 2^31 - 1 samples");
    AVL_HIP_CHECK(hipDeviceSynchronize());
    if (b->log_used != 0) {
        set_error("avl_builder_enable_replay_log: frames were already fused; enable the log on a fresh or reset builder");
        return AVL_ERR_STATE;
    }
    (void)hipFree(b->log.slot); (void)hipFree(b->log.rec);
    b->log = ReplayLog{};
    b->log_cap = 0;
    hipError_t e = hipMalloc((void**)&b->log.slot, (size_t)max_samples * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void**)&b->log.rec, (size_t)max_samples * sizeof(LogRec));
    if (e != hipSuccess) {
        (void)hipFree(b->log.slot); (void)hipFree(b->log.rec);
        b->log = ReplayLog{};
        set_error("avl_builder_enable_replay_log: hipMalloc failed: %s", hipGetErrorString(e));
        return AVL_ERR_HIP;
    }
    b->log_cap = max_samples;
    // the log's slot-sorted form (LogSegments: two words per sample + two per ACTIVE sample + the radix sort's storage) lives next to
    // the log itself: allocated here, not inside the first finalisation / merge
    (void)hipFree(b->rs_mem);
    b->rs_mem = nullptr;
    b->rs_bytes = b->rs_tmp_bytes = 0;
    {
        size_t tb = 0;
        if (rocprim::radix_sort_pairs(nullptr, tb, (uint32_t*)nullptr, (uint32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr, (size_t)max_samples, 0, 32,
                                      nullptr) == hipSuccess) {
            const size_t w = ((size_t)max_samples * 4 + 255) / 256 * 256;
            const size_t total = 4 * w + (tb + 255) / 256 * 256 + 256;
            if (hipMalloc((void**)&b->rs_mem, total) == hipSuccess) {
                b->rs_bytes = total;
                b->rs_tmp_bytes = (tb + 255) / 256 * 256;
            } else {
                b->rs_mem = nullptr;      // (not fatal: LogSegments falls back to the stream-ordered pool)
            }
        }
        (void)hipGetLastError();
    }
    return AVL_OK;
}

int avl_builder_set_max_capacity(avl_builder* b, int64_t max_capacity) {
    AVL_REQUIRE(b, "avl_builder_set_max_capacity: null handle");
    AVL_REQUIRE(max_capacity == 0 || (max_capacity >= b->capacity && max_capacity < (1ll << 31)),
                "avl_builder_set_max_capacity: must be 0 (fixed) or in [capacity, 2^31)");
    b->max_capacity = max_capacity;
    return AVL_OK;
}

int avl_builder_capacity(avl_builder* b, int64_t* h_capacity) {
    AVL_REQUIRE(b && h_capacity, "avl_builder_capacity: null argument");
    *h_capacity = b->capacity;
    return AVL_OK;
}

int avl_builder_create(avl_builder** h_out, int gs, double cs, int vh, int D, int64_t capacity) {
    return avl_builder_create_grid(h_out, gs, gs, vh, cs, D, capacity);
}

// B == 0: one frame, pointers passed directly.  B > 0: h_*_ptrs[B] give the per-frame device buffers, h_pc_transform holds B 4x4s,
// frames frame_idx .. frame_idx + B - 1 are fused by ONE K1/K2/K3 triple (samples of a voxel from different frames share one list).
static int integrate_impl(avl_builder* b, const void* d_depth, int depth_u16, double depth_div, int H, int W, const double* h_calib,
                          const double* h_calib_inv, const double* h_pc_transform, const int32_t* d_sample_idx, int P,
                          const float* d_feat, int Hf, int Wf, const uint8_t* d_rgb, int64_t frame_idx, double min_depth,
                          double max_depth, double sigma_sq, const double* h_pcd_min, void* stream, int B = 0,
                          const void* const* h_depth_ptrs = nullptr, const int32_t* const* h_sample_ptrs = nullptr,
                          const float* const* h_feat_ptrs = nullptr, const uint8_t* const* h_rgb_ptrs = nullptr) {
    AVL_REQUIRE(b, "avl_builder_integrate_frame: null handle");
    // the records a previous launch prepared are good for THIS call only: whatever path leaves this function (errors included),
    // a later call must not find them (the ring slot they were read from may hold another frame by then)
    const auto held = b->pre_held;
    b->pre_held.valid = false;
    AVL_REQUIRE(H > 0 && W > 0 && Hf > 0 && Wf > 0 && P >= 0, "avl_builder_integrate_frame: bad shape");
    AVL_REQUIRE(P < (1 << 30), "avl_builder_integrate_frame: at most 2^30 samples per frame");
    AVL_REQUIRE(B >= 0 && (int64_t)(B > 0 ? B : 1) * P < (1ll << 30), "avl_builder_integrate_batch: at most 2^30 samples per launch");
    AVL_REQUIRE(frame_idx >= 0 && frame_idx + B < (1ll << 30), "avl_builder_integrate_frame: bad frame_idx");
    AVL_REQUIRE(sigma_sq > 0, "avl_builder_integrate_frame: sigma_sq must be positive");
    if (P == 0) return AVL_OK;
    AVL_REQUIRE(h_calib && h_calib_inv && h_pc_transform, "avl_builder_integrate_frame: null pointer");
    if (B == 0) AVL_REQUIRE(d_depth && d_sample_idx && d_feat && d_rgb, "avl_builder_integrate_frame: null pointer");
    else AVL_REQUIRE(h_depth_ptrs && h_sample_ptrs && h_feat_ptrs && h_rgb_ptrs, "avl_builder_integrate_batch: null pointer table");
    hipStream_t st = as_stream(stream);
    const int P_frame = P;
    if (B > 0) P = B * P_frame;   // total samples of the launch
    int rc = ensure_recs(b, P, st);
    if (rc != AVL_OK) return rc;
    // worst case every sample of this launch creates a voxel: when the host-side bound reaches the capacity, read the real
    // counter (one sync, rare: the bound advances ~20x faster than the map) and double the accumulators if it is really close
    if (b->max_capacity > b->capacity && b->vox_bound + P > b->capacity) {
        int64_t have = 0;
        rc = read_counter(b, 0, &have, st);
        if (rc != AVL_OK) return rc;
        b->vox_bound = have;
        int64_t need = have + P;                       // a grid cannot hold more voxels than it has cells
        if (need > (int64_t)b->ncell) need = (int64_t)b->ncell;
        if (need > b->capacity) {
            rc = grow_builder(b, need, st);
            if (rc != AVL_OK) return rc;
        }
    }
    b->vox_bound += P;
    const unsigned long long key_bias = b->key_bias;
    if (B > 0) {
        if (B > b->table_cap) {
            AVL_HIP_CHECK(hipStreamSynchronize(st));
            if (b->d_table) AVL_HIP_CHECK(hipFree(b->d_table));
            b->d_table = nullptr;
            AVL_HIP_CHECK(hipMalloc((void**)&b->d_table, (size_t)(B + 16) * sizeof(BatchEntry)));
            b->table_cap = B + 16;
        }
        std::vector<BatchEntry> tab((size_t)B);
        for (int i = 0; i < B; ++i) {
            AVL_REQUIRE(h_depth_ptrs[i] && h_sample_ptrs[i] && h_feat_ptrs[i] && h_rgb_ptrs[i], "avl_builder_integrate_batch: null frame %d", i);
            memcpy(tab[i].t, h_pc_transform + 16 * i, 16 * sizeof(double));
            tab[i].depth = reinterpret_cast<const float*>(h_depth_ptrs[i]);
            tab[i].samples = h_sample_ptrs[i];
            tab[i].rgb = h_rgb_ptrs[i];
            tab[i].feat = h_feat_ptrs[i];
            tab[i].frame_key = key_bias | ((unsigned long long)(frame_idx + i) << 32);
        }
        // pageable source: the runtime stages the bytes before returning, so `tab` may go out of scope
        AVL_HIP_CHECK(hipMemcpyAsync(b->d_table, tab.data(), (size_t)B * sizeof(BatchEntry), hipMemcpyHostToDevice, st));
    }
    FrameParams fp;
    for (int i = 0; i < 9; ++i) { fp.kinv[i] = h_calib_inv[i]; fp.k[i] = h_calib[i]; fp.kf[i] = 0.0; }
    // get_sim_cam_mat(h, w): eye(3); [0,0] = [1,1] = w/2; [0,2] = w/2; [1,2] = h/2
    fp.kf[0] = fp.kf[4] = (double)Wf / 2.0;
    fp.kf[2] = (double)Wf / 2.0;
    fp.kf[5] = (double)Hf / 2.0;
    fp.kf[8] = 1.0;
    for (int i = 0; i < 16; ++i) fp.t[i] = h_pc_transform[i];
    fp.min_depth = min_depth; fp.max_depth = max_depth;
    fp.two_sigma_sq = 2 * sigma_sq;
    fp.cs = b->cs; fp.half_gs = (double)b->gs / 2.0;
    fp.H = H; fp.W = W; fp.Hf = Hf; fp.Wf = Wf; fp.n0 = b->n0; fp.n1 = b->gs; fp.n2 = b->vh; fp.P = P;
    fp.batch = B > 0 ? b->d_table : nullptr;
    fp.P_frame = P_frame;
    fp.mode = h_pcd_min ? 1 : 0;
    for (int i = 0; i < 3; ++i) fp.pcd_min[i] = h_pcd_min ? h_pcd_min[i] : 0.0;
    fp.depth_u16 = depth_u16;
    fp.depth_div = depth_div;
    fp.capacity = b->capacity;
    // PreGather: start from what the previous launch prepared for exactly this frame; let this launch prepare the frame the C loop
    // issues next (single float32 frames below the batched-launch size, the register-resident K3 kernels)
    fp.pre = nullptr;
    FrameParams fpn{};
    PreGather next{};
    const bool pre_ok = B == 0 && !depth_u16 && b->D <= 1536 && P < kAggregateSamples;
    if (pre_ok) {
        const auto& h = held;
        if (h.valid && h.samples == d_sample_idx && h.depth == d_depth && h.rgb == d_rgb && same_geometry(h.fp, fp)) fp.pre = b->pre_buf[h.buf];
        if (b->pre_next.samples) {
            fpn = fp;
            fpn.pre = nullptr;
            for (int i = 0; i < 16; ++i) fpn.t[i] = b->pre_next.h_pc_transform[i];
            next = PreGather{b->pre_buf[fp.pre ? 1 - h.buf : 0], b->pre_next.depth, b->pre_next.samples, b->pre_next.rgb};
        }
    }
    const int next_buf = next.out == b->pre_buf[1] ? 1 : 0;
    auto hold_next = [&]() {
        if (!next.out) return;
        b->pre_held.fp = fpn;
        b->pre_held.samples = next.samples;
        b->pre_held.depth = next.depth;
        b->pre_held.rgb = next.rgb;
        b->pre_held.buf = next_buf;
        b->pre_held.valid = true;
    };
    const unsigned long long frame_key = b->key_bias | ((unsigned long long)frame_idx << 32);

    if (b->ls_cache) drop_log_segments(b, st);   // the sorted log of the last merge is stale from here on
    if (b->log.slot && b->log_used + P > b->log_cap) {
        set_error("replay log full (%lld samples): enable it with a larger max_samples", b->log_cap);
        return AVL_ERR_CAPACITY;
    }
    const unsigned pb = (unsigned)((P + 255) / 256);
    if (b->deferred && B == 0 && b->D <= 1536) {
        // ONE launch: K1 + K2 of this frame next to K3 of the previous one; this frame's K3 rides in the next launch (or a flush)
        const FusePrev prev{b->pend.P, b->pend.frame_key, b->recs_alt, b->head_alt, b->pend.feat};
        const unsigned pb = (unsigned)((P + AVL_K3_THREADS - 1) / AVL_K3_THREADS);      // (its K1 + K2 workgroups have the kernel's size)
        if (b->D <= 256) launch_pipe<1>(b, fp, pb, d_depth, d_sample_idx, d_rgb, frame_key, prev, st, fpn, next);
        else if (b->D <= 512) launch_pipe<2>(b, fp, pb, d_depth, d_sample_idx, d_rgb, frame_key, prev, st, fpn, next);
        else if (b->D == 768) launch_pipe<3>(b, fp, pb, d_depth, d_sample_idx, d_rgb, frame_key, prev, st, fpn, next);
        else if (b->D <= 1024) launch_pipe<4>(b, fp, pb, d_depth, d_sample_idx, d_rgb, frame_key, prev, st, fpn, next);
        else launch_pipe<6>(b, fp, pb, d_depth, d_sample_idx, d_rgb, frame_key, prev, st, fpn, next);
        if (b->log.slot) b->log_used += P;
        b->pend.P = P;
        b->pend.frame_key = frame_key;
        b->pend.feat = d_feat;
        std::swap(b->recs, b->recs_alt);
        std::swap(b->head, b->head_alt);
        AVL_HIP_CHECK(hipGetLastError());
        hold_next();
        return AVL_OK;
    }
    rc = flush_pending(b, st);
    if (rc != AVL_OK) return rc;
    const bool k2_compacts = !(fp.pre || next.out);
    if (!k2_compacts) {     // a frame of the C loop: K1 from the prepared records and / or the next frame's stateless half in front
        const unsigned gb = next.out ? pb : 0u;
        hipLaunchKernelGGL(voxelize_link_next_kernel, dim3(gb + pb), dim3(256), 0, st, fp, reinterpret_cast<const float*>(d_depth), d_sample_idx,
                           d_rgb, b->cell_slot, b->slot_cell, b->recs, b->head, b->counters, b->err_flags, b->log, b->log_used, frame_key, fpn,
                           next, (int)gb);
    } else {
        hipLaunchKernelGGL(voxelize_link_kernel, dim3(pb), dim3(256), 0, st, fp, reinterpret_cast<const float*>(d_depth), d_sample_idx, d_rgb,
                           b->cell_slot, b->slot_cell, b->recs, b->head, b->counters, b->err_flags, b->log, b->log_used, frame_key, b->owners);
    }
    if (b->log.slot) b->log_used += P;
    rc = launch_fuse(b, P, frame_key, fp.batch, P_frame, b->recs, b->head, d_feat, st, k2_compacts);
    if (rc == AVL_OK) hold_next();
    return rc;
}

int avl_builder_set_deferred_fuse(avl_builder* b, int on, void* stream) {
    AVL_REQUIRE(b, "avl_builder_set_deferred_fuse: null handle");
    if (!on) {
        int rc = flush_pending(b, as_stream(stream));
        if (rc != AVL_OK) return rc;
    }
    b->deferred = on ? 1 : 0;
    return AVL_OK;
}

int avl_builder_flush(avl_builder* b, void* stream) {
    AVL_REQUIRE(b, "avl_builder_flush: null handle");
    return flush_pending(b, as_stream(stream));
}

int avl_builder_release_scratch(avl_builder* b, int64_t keep_bytes, void* stream) {
    AVL_REQUIRE(b, "avl_builder_release_scratch: null handle");
    AVL_REQUIRE(keep_bytes >= 0, "avl_builder_release_scratch: keep_bytes < 0");
    hipStream_t st = as_stream(stream);
    drop_log_segments(b, st);                       // the slot-sorted replay log cached between the two replay calls of a merge
    AVL_HIP_CHECK(hipStreamSynchronize(st));        // the pool only gives back what no stream still uses
    avl::trim_mempool((uint64_t)keep_bytes);
    return AVL_OK;
}

int avl_builder_integrate_frame(avl_builder* b, const float* d_depth, int H, int W, const double* h_calib,
                                const double* h_calib_inv, const double* h_pc_transform, const int32_t* d_sample_idx, int P,
                                const float* d_feat, int Hf, int Wf, const uint8_t* d_rgb, int64_t frame_idx,
                                double min_depth, double max_depth, double sigma_sq, void* stream) {
    return integrate_impl(b, d_depth, 0, 1.0, H, W, h_calib, h_calib_inv, h_pc_transform, d_sample_idx, P, d_feat, Hf, Wf, d_rgb,
                          frame_idx, min_depth, max_depth, sigma_sq, nullptr, stream);
}

int avl_builder_integrate_frame_global(avl_builder* b, const void* d_depth, int depth_is_u16, double depth_div, int H, int W,
                                       const double* h_calib, const double* h_calib_inv, const double* h_transform,
                                       const int32_t* d_sample_idx, int P, const float* d_feat, int Hf, int Wf,
                                       const uint8_t* d_rgb, int64_t frame_idx, double min_depth, double max_depth,
                                       double sigma_sq, const double* h_pcd_min, void* stream) {
    AVL_REQUIRE(h_pcd_min, "avl_builder_integrate_frame_global: pcd_min is required");
    AVL_REQUIRE(!depth_is_u16 || depth_div > 0, "avl_builder_integrate_frame_global: depth_div must be positive");
    return integrate_impl(b, d_depth, depth_is_u16 ? 1 : 0, depth_div, H, W, h_calib, h_calib_inv, h_transform, d_sample_idx, P, d_feat,
                          Hf, Wf, d_rgb, frame_idx, min_depth, max_depth, sigma_sq, h_pcd_min, stream);
}

int avl_builder_integrate_batch(avl_builder* b, int B, const float* const* h_depth_ptrs, int H, int W, const double* h_calib,
                                const double* h_calib_inv, const double* h_pc_transforms, const int32_t* const* h_sample_ptrs, int P,
                                const float* const* h_feat_ptrs, int Hf, int Wf, const uint8_t* const* h_rgb_ptrs, int64_t frame_idx0,
                                double min_depth, double max_depth, double sigma_sq, void* stream) {
    AVL_REQUIRE(B > 0, "avl_builder_integrate_batch: B must be positive");
    return integrate_impl(b, nullptr, 0, 1.0, H, W, h_calib, h_calib_inv, h_pc_transforms, nullptr, P, nullptr, Hf, Wf, nullptr, frame_idx0,
                          min_depth, max_depth, sigma_sq, nullptr, stream, B, reinterpret_cast<const void* const*>(h_depth_ptrs),
                          h_sample_ptrs, h_feat_ptrs, h_rgb_ptrs);
}

int avl_builder_integrate_frames(avl_builder* b, int n_frames, const float* const* h_depth_ptrs, int H, int W, const double* h_calib,
                                 const double* h_calib_inv, const double* h_pc_transforms, const int32_t* const* h_sample_ptrs, int P,
                                 const float* const* h_feat_ptrs, int Hf, int Wf, const uint8_t* const* h_rgb_ptrs, int64_t frame_idx0,
                                 double min_depth, double max_depth, double sigma_sq, void* stream) {
    AVL_REQUIRE(n_frames > 0, "avl_builder_integrate_frames: n_frames must be positive");
    AVL_REQUIRE(h_depth_ptrs && h_sample_ptrs && h_feat_ptrs && h_rgb_ptrs && h_pc_transforms, "avl_builder_integrate_frames: null pointer table");
    AVL_REQUIRE(b, "avl_builder_integrate_frames: null handle");
    for (int i = 0; i < n_frames; ++i) {
        // the frame after this one, for the gather workgroups of this frame's launch (PreGather); the last frame of a call has none:
        // what a later call brings is not known to be resident yet
        b->pre_next = {};
        if (i + 1 < n_frames && h_sample_ptrs[i + 1] && h_depth_ptrs[i + 1] && h_rgb_ptrs[i + 1])
            b->pre_next = {h_sample_ptrs[i + 1], h_depth_ptrs[i + 1], h_rgb_ptrs[i + 1], h_pc_transforms + 16 * (i + 1)};
        const int rc = integrate_impl(b, h_depth_ptrs[i], 0, 1.0, H, W, h_calib, h_calib_inv, h_pc_transforms + 16 * i, h_sample_ptrs[i], P,
                                      h_feat_ptrs[i], Hf, Wf, h_rgb_ptrs[i], frame_idx0 + i, min_depth, max_depth, sigma_sq, nullptr, stream);
        b->pre_next = {};
        if (rc != AVL_OK) return rc;
    }
    return AVL_OK;
}

int avl_builder_num_voxels(avl_builder* b, int64_t* h_n, void* stream) {
    AVL_REQUIRE(b && h_n, "avl_builder_num_voxels: null argument");
    int rc = flush_pending(b, as_stream(stream));
    if (rc != AVL_OK) return rc;
    rc = builder_check_flags(b, as_stream(stream));
    if (rc != AVL_OK) return rc;
    rc = read_counter(b, 0, h_n, as_stream(stream));
    if (rc == AVL_OK && *h_n > b->capacity) *h_n = b->capacity;
    return rc;
}

int avl_builder_num_points(avl_builder* b, int64_t* h_n, void* stream) {
    AVL_REQUIRE(b && h_n, "avl_builder_num_points: null argument");
    if (int rc = flush_pending(b, as_stream(stream)); rc != AVL_OK) return rc;
    return read_counter(b, 1, h_n, as_stream(stream));
}

int avl_builder_num_groups(avl_builder* b, int64_t* h_n, void* stream) {
    AVL_REQUIRE(b && h_n, "avl_builder_num_groups: null argument");
    if (int rc = flush_pending(b, as_stream(stream)); rc != AVL_OK) return rc;
    return read_counter(b, 2, h_n, as_stream(stream));
}

static int launch_finalize(int64_t n, int D, int gs, int vh, int64_t row0, const int32_t* perm, const int32_t* d_cell,
                           const double* d_sum_feat, int64_t ld_sf, const double* d_sum_w4, int64_t ld_w4, const float* d_first_feat,
                           const double* d_first_alpha, float* d_grid_feat, int32_t* d_grid_pos, float* d_weight,
                           uint8_t* d_grid_rgb, int32_t* d_occupied_ids, hipStream_t st) {
    int64_t blocks = (n + 3) / 4;
    const int64_t maxb = (int64_t)num_cus() * 16;
    if (blocks > maxb) blocks = maxb;
    hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)blocks), dim3(256), 0, st, n, D, gs, vh, row0, perm, d_cell, d_sum_feat, ld_sf,
                       d_sum_w4, ld_w4, d_first_feat, d_first_alpha, d_grid_feat, d_grid_pos, d_weight, d_grid_rgb, d_occupied_ids);
    AVL_HIP_CHECK(hipGetLastError());
    return AVL_OK;
}

int avl_finalize_raw(int64_t n, int D, int gs, int vh, const int32_t* d_cell, const double* d_sum_feat,
                     const double* d_sum_w4, const float* d_first_feat, const double* d_first_alpha, float* d_grid_feat,
                     int32_t* d_grid_pos, float* d_weight, uint8_t* d_grid_rgb, int32_t* d_occupied_ids, void* stream) {
    AVL_REQUIRE(n >= 0 && D > 0 && gs > 0 && vh > 0, "avl_finalize_raw: bad shape");
    if (n == 0) return AVL_OK;
    AVL_REQUIRE(d_cell && d_sum_w4 && d_first_alpha, "avl_finalize_raw: null input");
    AVL_REQUIRE(!d_grid_feat || (d_sum_feat && d_first_feat), "avl_finalize_raw: grid_feat needs sum_feat and first_feat");
    return launch_finalize(n, D, gs, vh, 0, nullptr, d_cell, d_sum_feat, D, d_sum_w4, 4, d_first_feat, d_first_alpha, d_grid_feat,
                           d_grid_pos, d_weight, d_grid_rgb, d_occupied_ids, as_stream(stream));
}

int avl_finalize_merged(int64_t n, int64_t row0, int D, int gs, int vh, const int32_t* d_cell, const double* d_acc,
                        int64_t ld_acc, float* d_grid_feat, int32_t* d_grid_pos, float* d_weight, uint8_t* d_grid_rgb,
                        int32_t* d_occupied_ids, void* stream) {
    AVL_REQUIRE(n >= 0 && row0 >= 0 && D > 0 && gs > 0 && vh > 0 && ld_acc >= D + 4, "avl_finalize_merged: bad shape");
    AVL_REQUIRE(row0 + n < (1ll << 31), "avl_finalize_merged: voxel ids must fit int32");
    if (n == 0) return AVL_OK;
    AVL_REQUIRE(d_cell && d_acc, "avl_finalize_merged: null input");
    return launch_finalize(n, D, gs, vh, row0, nullptr, d_cell, d_acc, ld_acc, d_acc + D, ld_acc, nullptr, nullptr, d_grid_feat,
                           d_grid_pos, d_weight, d_grid_rgb, d_occupied_ids, as_stream(stream));
}

int avl_builder_scatter_merge(avl_builder* b, int64_t n, const int64_t* d_row_of_slot, const uint64_t* d_global_key,
                              double* d_acc, int64_t ld_acc, void* stream) {
    AVL_REQUIRE(b, "avl_builder_scatter_merge: null handle");
    hipStream_t st = as_stream(stream);
    int64_t have = 0;
    int rc = avl_builder_num_voxels(b, &have, stream);
    if (rc != AVL_OK) return rc;
    AVL_REQUIRE(n == have, "avl_builder_scatter_merge: n=%lld but the map holds %lld voxels", (long long)n, (long long)have);
    AVL_REQUIRE(ld_acc >= b->D + 4, "avl_builder_scatter_merge: ld_acc must be >= D + 4");
    if (n == 0) return AVL_OK;
    AVL_REQUIRE(d_row_of_slot && d_global_key && d_acc, "avl_builder_scatter_merge: null pointer");
    int64_t blocks = (n + 3) / 4;
    const int64_t maxb = (int64_t)num_cus() * 16;
    if (blocks > maxb) blocks = maxb;
    hipLaunchKernelGGL(scatter_merge_kernel, dim3((unsigned)blocks), dim3(256), 0, st, n, b->D, d_row_of_slot,
                       reinterpret_cast<const unsigned long long*>(d_global_key), b->slot_key, b->sum_feat, b->sum_w4, b->first_feat,
                       b->first_alpha, d_acc, ld_acc);
    AVL_HIP_CHECK(hipGetLastError());
    return AVL_OK;
}

static int check_slot_list(avl_builder* b, int64_t k, const void* slots, const void* out, int64_t ld, const char* who, void* stream) {
    AVL_REQUIRE(b, "%s: null handle", who);
    AVL_REQUIRE(k >= 0 && ld >= b->D, "%s: bad shape", who);
    int64_t have = 0;
    int rc = avl_builder_num_voxels(b, &have, stream);
    if (rc != AVL_OK) return rc;
    AVL_REQUIRE(k <= have, "%s: %lld slots listed but the map holds %lld voxels", who, (long long)k, (long long)have);
    AVL_REQUIRE(k == 0 || (slots && out), "%s: null pointer", who);
    return AVL_OK;
}

int avl_builder_export_rows_f32(avl_builder* b, int64_t k, const int32_t* d_slots, float* d_out, int64_t ld, void* stream) {
    int rc = check_slot_list(b, k, d_slots, d_out, ld, "avl_builder_export_rows_f32", stream);
    if (rc != AVL_OK || k == 0) return rc;
    const int64_t blocks = std::min<int64_t>((k + 3) / 4, (int64_t)num_cus() * 16);
    hipLaunchKernelGGL(export_rows_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), k, b->D, d_slots, b->sum_feat,
                       b->sum_w4, b->first_feat, b->first_alpha, d_out, ld);
    AVL_HIP_CHECK(hipGetLastError());
    return AVL_OK;
}

int avl_builder_export_rows_f64(avl_builder* b, int64_t k, const int32_t* d_slots, const uint8_t* d_own, double* d_out, int64_t ld,
                                void* stream) {
    int rc = check_slot_list(b, k, d_slots, d_out, ld, "avl_builder_export_rows_f64", stream);
    if (rc != AVL_OK || k == 0) return rc;
    AVL_REQUIRE(d_own, "avl_builder_export_rows_f64: null ownership flags");
    const int64_t blocks = std::min<int64_t>((k + 3) / 4, (int64_t)num_cus() * 16);
    hipLaunchKernelGGL(export_rows_f64_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), k, b->D, d_slots, d_own, b->sum_feat,
                       b->first_feat, b->first_alpha, d_out, ld);
    AVL_HIP_CHECK(hipGetLastError());
    return AVL_OK;
}

int avl_builder_m2_pack(avl_builder* b, int64_t n, int ws, int rank, int64_t own_r0, const int64_t* h_cum, const int64_t* h_lo,
                        const int64_t* h_dlo, const int64_t* h_row0, const int64_t* h_side_off, const int64_t* h_done_off,
                        const int64_t* h_part_off, const int32_t* d_order, const int32_t* d_row, const int32_t* d_prev, const int32_t* d_next,
                        const int32_t* d_sidx, int64_t* d_send, float* d_own_feat, void* stream) {
    AVL_REQUIRE(b, "avl_builder_m2_pack: null handle");
    AVL_REQUIRE(n >= 0 && ws >= 1 && ws <= 64 && rank >= 0 && rank < ws && own_r0 >= 0, "avl_builder_m2_pack: bad arguments");
    hipStream_t st = as_stream(stream);
    int rc = flush_pending(b, st);
    if (rc != AVL_OK) return rc;
    if (n == 0) return AVL_OK;
    AVL_REQUIRE(n <= b->capacity, "avl_builder_m2_pack: n=%lld exceeds the capacity %lld", (long long)n, (long long)b->capacity);
    AVL_REQUIRE(h_cum && h_lo && h_dlo && h_row0 && h_side_off && h_done_off && h_part_off && d_order && d_row && d_prev && d_next && d_sidx && d_send,
                "avl_builder_m2_pack: null pointer");
    M2PackSeg sg{};
    for (int q = 0; q <= 64; ++q) sg.cum[q] = h_cum[q < ws ? q : ws];
    for (int q = 0; q < ws; ++q) {
        sg.lo[q] = h_lo[q];
        sg.dlo[q] = h_dlo[q];
        sg.row0[q] = h_row0[q];
        sg.side_off[q] = h_side_off[q];
        sg.done_off[q] = h_done_off[q];
        sg.part_off[q] = h_part_off[q];
    }
    AVL_REQUIRE(sg.cum[0] == 0 && sg.cum[ws] == n, "avl_builder_m2_pack: the destination ranges cover %lld voxels, n = %lld", sg.cum[ws], (long long)n);
    int64_t blocks = (n + 3) / 4;
    const int64_t maxb = (int64_t)num_cus() * 16;
    if (blocks > maxb) blocks = maxb;
    hipLaunchKernelGGL(m2_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (long long)n, ws, rank, b->D, (long long)own_r0, sg, d_order, d_row,
                       d_prev, d_next, d_sidx, b->sum_feat, b->sum_w4, b->first_feat, b->first_alpha, reinterpret_cast<long long*>(d_send),
                       d_own_feat);
    AVL_HIP_CHECK(hipGetLastError());
    return AVL_OK;
}

int avl_finalize_side(int64_t n, int64_t row0, int gs, int vh, const int32_t* d_cell, const double* d_w4, int32_t* d_grid_pos,
                      float* d_weight, uint8_t* d_grid_rgb, int32_t* d_occupied_ids, void* stream) {
    AVL_REQUIRE(n >= 0 && row0 >= 0 && gs > 0 && vh > 0, "avl_finalize_side: bad shape");
    AVL_REQUIRE(row0 + n < (1ll << 31), "avl_finalize_side: voxel ids must fit int32");
    if (n == 0) return AVL_OK;
    AVL_REQUIRE(d_cell && d_w4, "avl_finalize_side: null input");
    hipLaunchKernelGGL(finalize_side_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 8192)), dim3(256), 0, as_stream(stream), n, gs,
                       vh, row0, d_cell, d_w4, d_grid_pos, d_weight, d_grid_rgb, d_occupied_ids);
    AVL_HIP_CHECK(hipGetLastError());
    return AVL_OK;
}

int avl_builder_finalize(avl_builder* b, int64_t n, float* d_grid_feat, int32_t* d_grid_pos, float* d_weight,
                         uint8_t* d_grid_rgb, int32_t* d_occupied_ids, void* stream) {
    return avl_builder_finalize_ex(b, n, d_grid_feat, d_grid_pos, d_weight, d_grid_rgb, d_occupied_ids, nullptr, 0, stream);
}

static int ensure_log_segments(avl_builder* b, int64_t n, hipStream_t st);

int avl_builder_finalize_ex(avl_builder* b, int64_t n, float* d_grid_feat, int32_t* d_grid_pos, float* d_weight,
                            uint8_t* d_grid_rgb, int32_t* d_occupied_ids, uint8_t* d_row_dirty, int clear_dirty, void* stream) {
    AVL_REQUIRE(b, "avl_builder_finalize: null handle");
    hipStream_t st = as_stream(stream);
    int64_t have = 0;
    int rc = avl_builder_num_voxels(b, &have, stream);
    if (rc != AVL_OK) return rc;
    AVL_REQUIRE(n == have, "avl_builder_finalize: n=%lld but the map holds %lld voxels", (long long)n, (long long)have);
    if (d_occupied_ids) AVL_HIP_CHECK(hipMemsetAsync(d_occupied_ids, 0xFF, b->ncell * sizeof(int32_t), st));
    if (n == 0) {
        AVL_HIP_CHECK(hipStreamSynchronize(st));
        return AVL_OK;
    }
    // rows in the reference's voxel-id order = slots sorted by first-touch key
    unsigned long long* keys_out = nullptr;
    int32_t *iota = nullptr, *perm = nullptr;
    void* tmp = nullptr;
    size_t tmp_bytes = 0;
    // one pool allocation for the four temporaries (each hipMallocAsync / hipFreeAsync pair is ~90 us of host time)
    AVL_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, b->slot_key, keys_out, iota, perm, (size_t)n, 0, 64, st));
    const size_t b_keys = ((size_t)n * sizeof(unsigned long long) + 255) / 256 * 256, b_idx = ((size_t)n * sizeof(int32_t) + 255) / 256 * 256;
    char* block = nullptr;
    AVL_HIP_CHECK(hipMallocAsync((void**)&block, b_keys + 2 * b_idx + (tmp_bytes ? tmp_bytes : 16), st));
    keys_out = reinterpret_cast<unsigned long long*>(block);
    iota = reinterpret_cast<int32_t*>(block + b_keys);
    perm = reinterpret_cast<int32_t*>(block + b_keys + b_idx);
    tmp = block + b_keys + 2 * b_idx;
    hipLaunchKernelGGL(iota_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 4096)), dim3(256), 0, st, iota, n);
    AVL_HIP_CHECK(rocprim::radix_sort_pairs(tmp, tmp_bytes, b->slot_key, keys_out, iota, perm, (size_t)n, 0, 64, st));
    rc = launch_finalize(n, b->D, b->gs, b->vh, 0, perm, b->slot_cell, b->sum_feat, b->D, b->sum_w4, 4, b->first_feat, b->first_alpha,
                         d_grid_feat, d_grid_pos, d_weight, d_grid_rgb, d_occupied_ids, st);
    if (rc == AVL_OK && b->log.slot && b->key_bias == 0 && b->log_used > 0 && (d_weight || d_grid_rgb)) {
        // exact sequential weight / grid_rgb: stable sort of the key-ordered log by slot, then replay per voxel
        // (the builder's ONE voxel-sorted form of the log, shared with the merge's replay: it lives in the scratch allocated with the log,
        // so a private second build here would overwrite a cached one; it stays valid until the next frame is fused)
        rc = ensure_log_segments(b, n, st);
        if (rc == AVL_OK) {
            const LogSegments& ls = *b->ls_cache;
            hipLaunchKernelGGL(replay_rgb_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 8192)), dim3(256), 0, st, n,
                               (long long)b->n0 * b->gs, perm, keys_out, ls.order, ls.seg_start, ls.seg_end, b->log, d_weight, d_grid_rgb);
            if (hipGetLastError() != hipSuccess) rc = AVL_ERR_HIP;
        }
    }
    if (rc == AVL_OK && d_row_dirty) {
        hipLaunchKernelGGL(row_dirty_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 8192)), dim3(256), 0, st, n, perm, b->dirty,
                           d_row_dirty, clear_dirty);
        if (hipGetLastError() != hipSuccess) rc = AVL_ERR_HIP;
    }
    (void)hipFreeAsync(block, st);
    if (rc != AVL_OK) return rc;
    AVL_HIP_CHECK(hipStreamSynchronize(st));
    return AVL_OK;
}

int avl_builder_drop_replay_cache(avl_builder* b, void* stream) {
    AVL_REQUIRE(b, "avl_builder_drop_replay_cache: null handle");
    drop_log_segments(b, as_stream(stream));
    return AVL_OK;
}

// the log sorted by voxel (LogSegments), built if it is not there: a merge asks for it FIRST, so that the compaction, the sort and the one
// host synchronisation in between run under the plan's host work and collectives instead of in front of the replay
static int ensure_log_segments(avl_builder* b, int64_t n, hipStream_t st) {
    if (b->ls_cache && b->ls_log_used == b->log_used && b->ls_n == n) return AVL_OK;
    drop_log_segments(b, st);
    b->ls_cache = new LogSegments();
    int rc = b->ls_cache->build(b, n, st);
    if (rc != AVL_OK) {
        drop_log_segments(b, st);
        return rc;
    }
    b->ls_log_used = b->log_used;
    b->ls_n = n;
    return AVL_OK;
}

int avl_builder_replay_prepare(avl_builder* b, int64_t n, void* stream) {
    AVL_REQUIRE(b, "avl_builder_replay_prepare: null handle");
    int64_t have = 0;
    int rc = avl_builder_num_voxels(b, &have, stream);
    if (rc != AVL_OK) return rc;
    AVL_REQUIRE(n == have, "avl_builder_replay_prepare: n=%lld but the map holds %lld voxels", (long long)n, (long long)have);
    if (!b->log.slot || b->key_bias != 0 || n == 0 || b->log_used == 0) return AVL_OK;      // (nothing to prepare: replay_chain reports a missing log)
    // (a build on a stream of the builder's own, overlapping the merge plan's host work, was measured in round 6: nothing in the 8-rank
    // rehearsal, +10 ms on the first merge of a process for the stream and its events -- the form is built on the caller's stream)
    return ensure_log_segments(b, n, as_stream(stream));
}

int avl_builder_replay_chain(avl_builder* b, int64_t n, const int64_t* d_row_of_slot, uint64_t grow_key, void* d_state,
                             void* stream) {
    AVL_REQUIRE(b, "avl_builder_replay_chain: null handle");
    hipStream_t st = as_stream(stream);
    int64_t have = 0;
    int rc = avl_builder_num_voxels(b, &have, stream);
    if (rc != AVL_OK) return rc;
    AVL_REQUIRE(n == have, "avl_builder_replay_chain: n=%lld but the map holds %lld voxels", (long long)n, (long long)have);
    if (!b->log.slot || b->key_bias != 0) {
        set_error("avl_builder_replay_chain: the builder has no replay log (avl_builder_enable_replay_log on a fresh builder)");
        return AVL_ERR_STATE;
    }
    if (n == 0 || b->log_used == 0) return AVL_OK;
    AVL_REQUIRE(d_row_of_slot && d_state, "avl_builder_replay_chain: null pointer");
    rc = ensure_log_segments(b, n, st);
    if (rc != AVL_OK) return rc;
    const LogSegments& ls = *b->ls_cache;
    hipLaunchKernelGGL(replay_chain_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 8192)), dim3(256), 0, st, n,
                       (unsigned long long)grow_key, d_row_of_slot, ls.order, ls.seg_start, ls.seg_end, b->log,
                       reinterpret_cast<ReplayState*>(d_state));
    if (hipGetLastError() != hipSuccess) rc = AVL_ERR_HIP;
    return rc;
}

// dst[d_rows[i] - row0, 0:cols] += src[i, 0:cols]  (float64): folds the contributions one rank received from ONE peer into its
// block of final rows.  A peer holds a voxel at most once, so the rows of a call are distinct: plain read-modify-write, and the
// caller's peer-by-peer order of the calls fixes the summation order (reproducible merges).  Wave per row.
__global__ __launch_bounds__(256) void rows_add_f64_kernel(int64_t n, int cols, const int64_t* __restrict__ rows, int64_t row0, int64_t nrows,
                                                           const double* __restrict__ src, int64_t ld_src, double* __restrict__ dst,
                                                           int64_t ld_dst, int* __restrict__ err_flag) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t i = wave0; i < n; i += nwaves) {
        const int64_t r = rows[i] - row0;
        if (r < 0 || r >= nrows) {
            if (lane == 0 && err_flag) atomicOr(err_flag, 1);
            continue;
        }
        const double* a = src + i * ld_src;
        double* o = dst + r * ld_dst;
        for (int c = lane; c < cols; c += 64) o[c] += a[c];
    }
}

// the same for rows of a few columns (the four [alpha, alpha rgb] sums of a side record): a lane per element, not a wave per row
__global__ __launch_bounds__(256) void rows_add_f64_narrow_kernel(int64_t n, int cols, const int64_t* __restrict__ rows, int64_t row0, int64_t nrows,
                                                                  const double* __restrict__ src, int64_t ld_src, double* __restrict__ dst,
                                                                  int64_t ld_dst, int* __restrict__ err_flag) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n * cols; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = t / cols;
        const int c = (int)(t - i * cols);
        const int64_t r = rows[i] - row0;
        if (r < 0 || r >= nrows) {
            if (c == 0 && err_flag) atomicOr(err_flag, 1);
            continue;
        }
        dst[r * ld_dst + c] += src[i * ld_src + c];
    }
}

static void launch_rows_add(int64_t n, int cols, const int64_t* d_rows, int64_t row0, int64_t nrows, const double* d_src, int64_t ld_src,
                            double* d_dst, int64_t ld_dst, int* flag, hipStream_t st) {
    const int64_t maxb = (int64_t)num_cus() * 16;
    if (cols <= 16) {
        const int64_t blocks = std::min<int64_t>((n * cols + 255) / 256, maxb);
        hipLaunchKernelGGL(rows_add_f64_narrow_kernel, dim3((unsigned)blocks), dim3(256), 0, st, n, cols, d_rows, row0, nrows, d_src, ld_src, d_dst,
                           ld_dst, flag);
    } else {
        const int64_t blocks = std::min<int64_t>((n + 3) / 4, maxb);
        hipLaunchKernelGGL(rows_add_f64_kernel, dim3((unsigned)blocks), dim3(256), 0, st, n, cols, d_rows, row0, nrows, d_src, ld_src, d_dst, ld_dst,
                           flag);
    }
}

int avl_rows_add_f64(int64_t n, int cols, const int64_t* d_rows, int64_t row0, int64_t nrows, const double* d_src, int64_t ld_src,
                     double* d_dst, int64_t ld_dst, void* stream) {
    AVL_REQUIRE(n >= 0 && cols > 0 && nrows >= 0 && ld_src >= cols && ld_dst >= cols, "avl_rows_add_f64: bad shape");
    if (n == 0) return AVL_OK;
    AVL_REQUIRE(d_rows && d_src && d_dst, "avl_rows_add_f64: null pointer");
    hipStream_t st = as_stream(stream);
    int* flag = static_cast<int*>(avl::scratch(64));
    if (!flag) return AVL_ERR_HIP;
    AVL_HIP_CHECK(hipMemsetAsync(flag, 0, sizeof(int), st));
    launch_rows_add(n, cols, d_rows, row0, nrows, d_src, ld_src, d_dst, ld_dst, flag, st);
    int h = 0;
    AVL_HIP_CHECK(hipMemcpyAsync(&h, flag, sizeof(int), hipMemcpyDeviceToHost, st));
    AVL_HIP_CHECK(hipStreamSynchronize(st));
    AVL_REQUIRE(h == 0, "avl_rows_add_f64: a row index lies outside [row0, row0 + nrows)");
    return AVL_OK;
}

int avl_rows_add_f64_async(int64_t n, int cols, const int64_t* d_rows, int64_t row0, int64_t nrows, const double* d_src, int64_t ld_src,
                           double* d_dst, int64_t ld_dst, int32_t* d_err_flag, void* stream) {
    AVL_REQUIRE(n >= 0 && cols > 0 && nrows >= 0 && ld_src >= cols && ld_dst >= cols, "avl_rows_add_f64_async: bad shape");
    if (n == 0) return AVL_OK;
    AVL_REQUIRE(d_rows && d_src && d_dst && d_err_flag, "avl_rows_add_f64_async: null pointer");
    launch_rows_add(n, cols, d_rows, row0, nrows, d_src, ld_src, d_dst, ld_dst, reinterpret_cast<int*>(d_err_flag), as_stream(stream));
    AVL_HIP_CHECK(hipGetLastError());
    return AVL_OK;
}

// out[rows[i], :] = (float)(acc[i, :] / w[rows[i]]): the shared rows of a rank's block from their float64 sums -- finalize_kernel's
// division -- without the (k, D) float64 and float32 temporaries of the tensor expression.  Wave per row.
__global__ __launch_bounds__(256) void rows_div_f32_kernel(int64_t k, int D, const double* __restrict__ acc, const int64_t* __restrict__ rows,
                                                           const double* __restrict__ w4, float* __restrict__ out, int64_t n_out,
                                                           int* __restrict__ err_flag) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t i = wave0; i < k; i += nwaves) {
        const int64_t r = rows[i];
        if (r < 0 || r >= n_out) {
            if (lane == 0 && err_flag) atomicOr(err_flag, 1);
            continue;
        }
        const double w = w4[r * 4];
        const double* a = acc + i * D;
        float* o = out + r * D;
        for (int c = lane; c < D; c += 64) o[c] = (float)(a[c] / w);
    }
}

int avl_rows_div_f32(int64_t k, int D, const double* d_acc, const int64_t* d_rows, const double* d_w4, float* d_out, int64_t n_out,
                     int32_t* d_err_flag, void* stream) {
    AVL_REQUIRE(k >= 0 && D > 0 && n_out >= 0, "avl_rows_div_f32: bad shape");
    if (k == 0) return AVL_OK;
    AVL_REQUIRE(d_acc && d_rows && d_w4 && d_out, "avl_rows_div_f32: null pointer");
    int64_t blocks = (k + 3) / 4;
    const int64_t maxb = (int64_t)num_cus() * 16;
    if (blocks > maxb) blocks = maxb;
    hipLaunchKernelGGL(rows_div_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), k, D, d_acc, d_rows, d_w4, d_out, n_out,
                       reinterpret_cast<int*>(d_err_flag));
    AVL_HIP_CHECK(hipGetLastError());
    return AVL_OK;
}

int avl_replay_state_apply(int64_t n, const void* d_state, float* d_weight, uint8_t* d_grid_rgb, void* stream) {
    AVL_REQUIRE(n >= 0, "avl_replay_state_apply: bad n");
    if (n == 0) return AVL_OK;
    AVL_REQUIRE(d_state, "avl_replay_state_apply: null state");
    hipLaunchKernelGGL(replay_apply_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 8192)), dim3(256), 0, as_stream(stream), n,
                       reinterpret_cast<const ReplayState*>(d_state), d_weight, d_grid_rgb);
    AVL_HIP_CHECK(hipGetLastError());
    return AVL_OK;
}

int avl_builder_import_map(avl_builder* b, int64_t n, const float* d_grid_feat, const int32_t* d_grid_pos,
                           const float* d_weight, const uint8_t* d_grid_rgb, void* stream) {
    AVL_REQUIRE(b, "avl_builder_import_map: null handle");
    AVL_REQUIRE(n >= 0, "avl_builder_import_map: bad n");
    hipStream_t st = as_stream(stream);
    // a map that grew past the initial capacity (the reference doubles its arrays, _reserve_map_space vlmap_builder.py:286-311,
    // and resumes such a map): grow like a frame launch would, if the handle is allowed to
    if (n > b->capacity && b->max_capacity > b->capacity) {
        const int rcg = grow_builder(b, n, st);
        if (rcg != AVL_OK) return rcg;
    }
    AVL_REQUIRE(n <= b->capacity, "avl_builder_import_map: %lld voxels exceed the capacity %lld (avl_builder_set_max_capacity lets it grow)",
                (long long)n, (long long)b->capacity);
    int64_t have = 0;
    int rc = avl_builder_num_voxels(b, &have, stream);
    if (rc != AVL_OK) return rc;
    if (have != 0) {
        set_error("avl_builder_import_map: the map already holds %lld voxels (import into an empty builder)", (long long)have);
        return AVL_ERR_STATE;
    }
    if (n == 0) {
        b->key_bias = 1ull << 62;   // continuing a map: the voxels of new frames order after every imported one (none here: the
        return AVL_OK;              // other ranks of a resumed multi-GPU build import nothing but must use the same key space)
    }
    AVL_REQUIRE(d_grid_feat && d_grid_pos && d_weight, "avl_builder_import_map: null input");
    int64_t blocks = (n + 3) / 4;
    const int64_t maxb = (int64_t)num_cus() * 16;
    if (blocks > maxb) blocks = maxb;
    hipLaunchKernelGGL(import_map_kernel, dim3((unsigned)blocks), dim3(256), 0, st, n, b->D, b->n0, b->gs, b->vh, d_grid_feat, d_grid_pos,
                       d_weight, d_grid_rgb, b->cell_slot, b->slot_cell, b->slot_key, b->sum_feat, b->sum_w4, b->first_feat,
                       b->first_alpha, b->err_flags);
    const unsigned long long nn = (unsigned long long)n;
    AVL_HIP_CHECK(hipMemcpyAsync(b->counters, &nn, sizeof(nn), hipMemcpyHostToDevice, st));
    int flags = 0;
    AVL_HIP_CHECK(hipMemcpyAsync(&flags, b->err_flags, sizeof(int), hipMemcpyDeviceToHost, st));
    AVL_HIP_CHECK(hipStreamSynchronize(st));
    if (flags & 4) {
        set_error("avl_builder_import_map: a grid_pos row lies outside the (gs, gs, vh) grid");
        return AVL_ERR_INVALID;
    }
    b->key_bias = 1ull << 62;
    b->vox_bound = n;
    return AVL_OK;
}

int avl_builder_export_raw(avl_builder* b, int64_t n, int32_t* d_cell, uint64_t* d_first_key, double* d_sum_feat,
                           double* d_sum_w4, float* d_first_feat, double* d_first_alpha, void* stream) {
    AVL_REQUIRE(b, "avl_builder_export_raw: null handle");
    hipStream_t st = as_stream(stream);
    int64_t have = 0;
    int rc = avl_builder_num_voxels(b, &have, stream);
    if (rc != AVL_OK) return rc;
    AVL_REQUIRE(n >= 0 && n <= have, "avl_builder_export_raw: n=%lld but the map holds %lld voxels", (long long)n, (long long)have);
    if (n == 0) return AVL_OK;
    const size_t D = (size_t)b->D;
    auto cp = [&](void* dst, const void* src, size_t bytes) {
        return dst ? hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st) : hipSuccess;
    };
    AVL_HIP_CHECK(cp(d_cell, b->slot_cell, (size_t)n * sizeof(int32_t)));
    AVL_HIP_CHECK(cp(d_first_key, b->slot_key, (size_t)n * sizeof(uint64_t)));
    AVL_HIP_CHECK(cp(d_sum_feat, b->sum_feat, (size_t)n * D * sizeof(double)));
    AVL_HIP_CHECK(cp(d_sum_w4, b->sum_w4, (size_t)n * 4 * sizeof(double)));
    AVL_HIP_CHECK(cp(d_first_feat, b->first_feat, (size_t)n * D * sizeof(float)));
    AVL_HIP_CHECK(cp(d_first_alpha, b->first_alpha, (size_t)n * sizeof(double)));
    return AVL_OK;
}

// values of the sort below: 0, 1, 2, ...
__global__ void iota64_kernel(int64_t* __restrict__ v, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) v[i] = i;
}

}  // extern "C"

template <typename K>
static hipError_t argsort_bits_impl(void* tmp, size_t& tmp_bytes, const void* keys, void* keys_out, const int64_t* iota, int64_t* perm,
                                    int64_t n, int bits, hipStream_t st) {
    // LSD radix sort over the low `bits` bits only (the keys are non-negative and smaller than 2^bits): stable
    return rocprim::radix_sort_pairs(tmp, tmp_bytes, reinterpret_cast<const K*>(keys), reinterpret_cast<K*>(keys_out), iota, perm,
                                     (size_t)n, 0, bits, st);
}

static size_t argsort_align(size_t b) { return (b + 255) / 256 * 256; }

extern "C" {

int avl_argsort_bits_work_bytes(int64_t n, int key_bytes, int bits, size_t* h_bytes) {
    AVL_REQUIRE(h_bytes && n >= 0 && n < (1ll << 31) && (key_bytes == 4 || key_bytes == 8) && bits >= 1 && bits <= 8 * key_bytes - 1,
                "avl_argsort_bits_work_bytes: bad arguments");
    size_t tmp_bytes = 0;
    const hipError_t e = key_bytes == 8 ? argsort_bits_impl<uint64_t>(nullptr, tmp_bytes, nullptr, nullptr, nullptr, nullptr, n ? n : 1, bits, nullptr)
                                        : argsort_bits_impl<uint32_t>(nullptr, tmp_bytes, nullptr, nullptr, nullptr, nullptr, n ? n : 1, bits, nullptr);
    AVL_HIP_CHECK(e);
    // [values 0..n-1 | sorted keys | rocPRIM's own storage (its size depends on the bit range: rocPRIM picks the passes by it)]
    *h_bytes = argsort_align((size_t)n * 8) + argsort_align((size_t)n * key_bytes) + argsort_align(tmp_bytes) + 256;
    return AVL_OK;
}

int avl_argsort_bits(int64_t n, const void* d_keys, int key_bytes, int bits, int64_t* d_perm, void* d_work, size_t work_bytes,
                     void* stream) {
    AVL_REQUIRE(n >= 0 && n < (1ll << 31), "avl_argsort_bits: bad n");
    AVL_REQUIRE(key_bytes == 4 || key_bytes == 8, "avl_argsort_bits: keys are int32 or int64");
    AVL_REQUIRE(bits >= 1 && bits <= 8 * key_bytes - 1, "avl_argsort_bits: bits must be in [1, %d]", 8 * key_bytes - 1);
    if (n == 0) return AVL_OK;
    AVL_REQUIRE(d_keys && d_perm && d_work, "avl_argsort_bits: null pointer");
    size_t need = 0;
    int rc = avl_argsort_bits_work_bytes(n, key_bytes, bits, &need);
    if (rc != AVL_OK) return rc;
    AVL_REQUIRE(work_bytes >= need, "avl_argsort_bits: work buffer of %zu bytes, %zu needed (avl_argsort_bits_work_bytes)", work_bytes, need);
    hipStream_t st = as_stream(stream);
    char* w = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(d_work) + 255) / 256 * 256);
    int64_t* iota = reinterpret_cast<int64_t*>(w);
    void* keys_out = w + argsort_align((size_t)n * 8);
    void* tmp = reinterpret_cast<char*>(keys_out) + argsort_align((size_t)n * key_bytes);
    size_t tmp_bytes = need - 256 - argsort_align((size_t)n * 8) - argsort_align((size_t)n * key_bytes);
    hipLaunchKernelGGL(iota64_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 4096)), dim3(256), 0, st, iota, n);
    const hipError_t e = key_bytes == 8 ? argsort_bits_impl<uint64_t>(tmp, tmp_bytes, d_keys, keys_out, iota, d_perm, n, bits, st)
                                        : argsort_bits_impl<uint32_t>(tmp, tmp_bytes, d_keys, keys_out, iota, d_perm, n, bits, st);
    if (e != hipSuccess) {
        set_error("avl_argsort_bits: %s", hipGetErrorString(e));
        return AVL_ERR_HIP;
    }
    return AVL_OK;
}

int avl_points_bbox(const void* d_depth, int depth_is_u16, double depth_div, int H, int W, const double* h_calib_inv,
                    const double* h_transform, const int32_t* d_sample_idx, int P, double min_depth, double max_depth,
                    double* h_minmax, void* stream) {
    AVL_REQUIRE(H > 0 && W > 0 && P >= 0 && h_calib_inv && h_transform && h_minmax, "avl_points_bbox: bad arguments");
    if (P == 0) return AVL_OK;
    AVL_REQUIRE(d_depth && d_sample_idx, "avl_points_bbox: null pointer");
    hipStream_t st = as_stream(stream);
    auto key = [](double v) {
        unsigned long long u;
        memcpy(&u, &v, 8);
        return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
    };
    auto unkey = [](unsigned long long k) {
        unsigned long long u = (k & 0x8000000000000000ull) ? (k & 0x7fffffffffffffffull) : ~k;
        double v;
        memcpy(&v, &u, 8);
        return v;
    };
    unsigned long long h_keys[6];
    for (int i = 0; i < 6; ++i) h_keys[i] = key(h_minmax[i]);
    unsigned long long* d_keys = nullptr;
    AVL_HIP_CHECK(hipMallocAsync((void**)&d_keys, sizeof(h_keys), st));
    AVL_HIP_CHECK(hipMemcpyAsync(d_keys, h_keys, sizeof(h_keys), hipMemcpyHostToDevice, st));
    FrameParams fp{};
    for (int i = 0; i < 9; ++i) fp.kinv[i] = h_calib_inv[i];
    for (int i = 0; i < 16; ++i) fp.t[i] = h_transform[i];
    fp.min_depth = min_depth; fp.max_depth = max_depth;
    fp.H = H; fp.W = W; fp.P = P;
    fp.depth_u16 = depth_is_u16 ? 1 : 0;
    fp.depth_div = depth_div;
    hipLaunchKernelGGL(bbox_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st, fp, reinterpret_cast<const float*>(d_depth),
                       d_sample_idx, d_keys);
    AVL_HIP_CHECK(hipMemcpyAsync(h_keys, d_keys, sizeof(h_keys), hipMemcpyDeviceToHost, st));
    AVL_HIP_CHECK(hipStreamSynchronize(st));
    (void)hipFreeAsync(d_keys, st);
    for (int i = 0; i < 6; ++i) h_minmax[i] = unkey(h_keys[i]);
    return AVL_OK;
}

}  // extern "C"
