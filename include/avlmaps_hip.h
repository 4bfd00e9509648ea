/*
 * avlmaps_hip.h -- C ABI of libavlmaps_hip.so: the MI355X (gfx950) implementation of the AVLMaps
 * map-creation / landmark-indexing hot path.
 *
 * The upstream reference is pure Python and has no FFI layer; each entry point below replaces a
 * span of reference Python (cited as path:line relative to the upstream repo root) and is what a
 * ctypes binding inside the reference would call.
 *
 * Conventions
 *   - every function returns an int status (AVL_OK == 0); avl_last_error() returns a thread-local
 *     human-readable message for the last failure on the calling thread.
 *   - pointers named d_* are DEVICE pointers (HIP), pointers named h_* are HOST pointers.
 *     Small fixed-size parameter blocks (3x3 / 4x4 float64 matrices) are always host pointers.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls are
 *     asynchronous with respect to the host unless stated otherwise.
 *   - caller-owned buffers in, library-owned state only behind opaque handles, nothing allocated
 *     by the library crosses the boundary.
 *   - one handle is used by one host thread at a time; one process (or thread) per GPU.
 */
#ifndef AVLMAPS_HIP_H
#define AVLMAPS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AVL_API __attribute__((visibility("default")))

enum {
    AVL_OK = 0,
    AVL_ERR_INVALID = 1,   /* bad argument                                   */
    AVL_ERR_HIP = 2,       /* a HIP runtime call failed                      */
    AVL_ERR_CAPACITY = 3,  /* voxel capacity exhausted (builder)             */
    AVL_ERR_NO_DEVICE = 4, /* no usable gfx950 device                        */
    AVL_ERR_STATE = 5      /* call not valid in the handle's current state   */
};

/* ------------------------------------------------------------------------------------------------
 * library / device plumbing
 * ------------------------------------------------------------------------------------------------ */
AVL_API const char* avl_last_error(void);
AVL_API int avl_version(void);                       /* major*10000 + minor*100 + patch */
AVL_API int avl_device_count(int* h_count);
AVL_API int avl_set_device(int device);
AVL_API int avl_get_device(int* h_device);           /* the calling THREAD's current device (HIP keeps it per thread) */
AVL_API int avl_device_name(int device, char* h_buf, size_t buf_len);
AVL_API int avl_device_sync(void);
AVL_API int avl_stream_create(void** h_stream_out);
AVL_API int avl_stream_destroy(void* stream);
AVL_API int avl_stream_sync(void* stream);
/* device memory helpers so that hosts without a GPU array library can drive the ABI */
AVL_API int avl_malloc(void** h_ptr_out, size_t bytes);
AVL_API int avl_free(void* d_ptr);
/* page-locked host memory (hipHostMalloc): device-to-host copies into it run at PCIe rate (~50 GB/s) instead of the ~6 GB/s of a
 * pageable destination; used as the staging buffer of the checkpoint rows */
/* Host only (no GPU needed): advance a NumPy legacy Mersenne-twister state (np.random.get_state(): key[624], pos) as n_shuffles
 * calls of np.random.shuffle on an array of n_items elements would (vlmap_builder.py:275-277 shuffles arange(H*W) once per
 * frame) -- the draws only, nothing is permuted.  A rank of a sharded build fast-forwards past the frames of the ranks before
 * it with this, so that a seeded N-rank run samples the pixels of the seeded single-process run. */
AVL_API int avl_mt19937_skip_shuffles(uint32_t* h_key624, int* h_pos, int64_t n_items, int64_t n_shuffles);
/* Host only: h_out[k] = perm[k * rate] of the permutation np.random.shuffle(np.arange(n_items)) produces from this state
 * (vlmap_builder.py:275-277: shuffle_mask[::depth_sample_rate]); the state advances exactly as the shuffle advances it.
 * h_scratch: n_items int32 of work space, h_out: ceil(n_items / rate) int32.  Same result as NumPy, about twice as fast (int32
 * indices, branch-free rejection): the serial part of a pixel-faithful build. */
AVL_API int avl_mt19937_shuffle_sample(uint32_t* h_key624, int* h_pos, int64_t n_items, int64_t rate, int32_t* h_scratch, int32_t* h_out);
AVL_API int avl_host_alloc(void** h_ptr_out, size_t bytes);
AVL_API int avl_host_free(void* h_ptr);
AVL_API int avl_memset(void* d_ptr, int value, size_t bytes, void* stream);
AVL_API int avl_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes, void* stream);
AVL_API int avl_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes, void* stream);
AVL_API int avl_memcpy_d2d(void* d_dst, const void* d_src, size_t bytes, void* stream);
/* Stable argsort of n NON-NEGATIVE integer keys smaller than 2^bits: d_perm[i] = position of the i-th smallest key, ties in
 * position order (an LSD radix sort over the low `bits` bits only).  d_keys int32 or int64 (key_bytes 4 / 8), d_perm int64[n].
 * The multi-GPU merge plan sorts 3-bit destination ranks and 22-bit row numbers held in int64 tensors (avlmaps_amd/parallel.py:
 * the bookkeeping around vlmap_builder.py:163-170's voxel ids); a generic 64-bit sort makes eight passes where one or three do. */
AVL_API int avl_argsort_bits_work_bytes(int64_t n, int key_bytes, int bits, size_t* h_bytes);
AVL_API int avl_argsort_bits(int64_t n, const void* d_keys, int key_bytes, int bits, int64_t* d_perm, void* d_work, size_t work_bytes,
                             void* stream);
/* The reverse: row d_rows[i] of d_dst (n_dst rows) = row i of d_src; rows distinct, 16-byte multiples at 16-byte aligned bases.
 * A row index outside [0, n_dst) is skipped and sets bit 0 of *d_err_flag (nullable).  The owner side of the multi-GPU merge places
 * the finished float32 rows it received at their final positions with it (avlmaps_amd/parallel.py). */
AVL_API int avl_scatter_rows(const void* d_src, int64_t row_bytes, const int64_t* d_rows, int64_t n, void* d_dst, int64_t n_dst,
                             int32_t* d_err_flag, void* stream);
/* dst row i = src row d_rows[i] (rows of row_bytes bytes, int64 indices): packs the rows an incremental checkpoint has to
 * write (the changed and the new voxels, avl_builder_finalize_ex's d_row_dirty) so that only they cross PCIe -- the reference
 * rewrites the whole map file every 100 frames (vlmap_builder.py:180-183). */
AVL_API int avl_gather_rows(const void* d_src, int64_t row_bytes, const int64_t* d_rows, int64_t n, void* d_dst, void* stream);
/* Read-only streaming probe over a caller buffer of `rows` x `row_floats` float32.
 * pattern bit 0: 0 = plain coalesced 16-byte grid-stride reads, 1 = the similarity kernels' row-line walk;
 * pattern bit 1: 0 = best GB/s of `iters` individually synchronised passes (burst rate),
 *                2 = mean GB/s of `iters` back-to-back passes (sustained rate at the package's power operating point);
 * pattern bits 2-3 (row-line walk only): workgroups per CU, 0 = two (default), 4 = one, 8 = three.
 * Used by bench.py to report the box's practical HBM read ceiling next to the 8 TB/s spec peak.  Synchronous. */
AVL_API int avl_hbm_read_probe(const void* d_buf, int64_t rows, int row_floats, int pattern, int iters, float* h_best_gbs,
                               void* stream);
/* HIP-event timing on `stream` (bench.py measures kernels on the stream they are launched on) */
AVL_API int avl_event_create(void** h_event_out);
AVL_API int avl_event_destroy(void* event);
AVL_API int avl_event_record(void* event, void* stream);
AVL_API int avl_event_sync(void* event);
/* work submitted to `stream` after this call waits (on the device, the host does not block) until `event` has completed: the
 * hand-over between the copy stream of the builder's pinned frame staging (avlmaps_amd/device.py FrameStager: depth / rgb / sample
 * list of frame i + k cross PCIe while frame i is fused) and the stream the frame kernels run on */
AVL_API int avl_stream_wait_event(void* stream, void* event);
AVL_API int avl_event_elapsed_ms(void* start, void* stop, float* h_ms);

/* ------------------------------------------------------------------------------------------------
 * (1) voxel x query similarity + row argmax
 *     replaces  avlmaps/utils/clip_utils.py:227-229   scores_list = map_feats @ text_feats.T
 *               avlmaps/map/vlmap.py:123-124          max_ids = np.argmax(scores_mat, axis=1)
 *               avlmaps/utils/index_utils.py:153-161  (same pair, obstacle classes)
 *     The score is the reference's RAW dot product (no normalisation).  Ties in the argmax resolve
 *     to the lowest query index, like np.argmax.
 * ------------------------------------------------------------------------------------------------ */
enum {
    AVL_SIM_AUTO = 0,      /* SPLIT_F16 when the shape allows it (D % 64 == 0, 16-byte aligned rows), else EXACT        */
    AVL_SIM_EXACT = 1,     /* float32 products and accumulation (an fmaf chain): v_mfma_f32_32x32x2_f32 on the matrix
                              cores when the shape allows it, otherwise the vector-ALU kernel (any N, D, Q, strides)     */
    AVL_SIM_SPLIT_F16 = 2, /* fp16 hi/lo split, 3 MFMA per product, fp32 accumulate: |err| <~ 1e-6*|a||q|, HBM-bound.
                              Range-guarded: a row whose largest |element| is outside [2^-7, 2^15) -- where the unscaled
                              fp16 pair would lose bits or saturate -- or non-finite is recomputed in float32 by a
                              follow-up kernel (np.argmax semantics for NaN rows), so every row is float32-class.       */
    AVL_SIM_EXACT_VALU = 3,/* force the vector-ALU float32 kernel                                                        */
    AVL_SIM_PREPARED = 4,  /* d_feat was converted by avl_sim_prepare_map: SPLIT_F16 without the on-the-fly split        */
    AVL_SIM_PREPARED24 = 5 /* d_feat is the compact 3-byte form of avl_sim_prepare_map24 (avl_sim_scores_prepared24, or
                              avl_sim_scores_blocks with d_feat = d_map24, ld_feat = D and its d_row_scale)               */
};

/* One-off, IN-PLACE conversion of a device-resident float32 map (N, D; D % 64 == 0, 16-byte aligned rows) into the split
 * layout the matrix-core kernel consumes directly: every group of 8 floats (32 bytes) becomes fp16 hi[8] | fp16 lo[8]
 * (same 4 bytes per element, same row stride).  Meant for a map that is indexed many times (VLMap keeps its private device
 * copy in this form).  The float32 values are not recoverable exactly.
 *   d_row_scale != NULL (N floats, out): every row is first scaled by the power of two that brings its largest |element| into
 *     [2^14, 2^15) and 2^-s is stored here; pass it to avl_sim_scores_prepared.  Rows of any magnitude (a voxel observed once
 *     from 5 m away stores feat * exp(-r^2/1.2) ~ 1e-9, vlmap_builder.py:166-168) keep the full ~22 bits: this is the form
 *     VLMap uses.  Rows containing NaN / inf are left unscaled and score NaN (argmax 0).
 *   d_row_scale == NULL: no scaling; avl_sim_scores(..., AVL_SIM_PREPARED) then gives scores bit-identical to the on-the-fly
 *     split of the raw map (without its range guard: meant for maps known to be LSeg-scale). */
AVL_API int avl_sim_prepare_map(float* d_feat, int64_t N, int D, int64_t ld_feat, float* d_row_scale, void* stream);

/*
 * d_feat     (N, D) float32 row-major with row stride ld_feat (elements)  -- VLMap.grid_feat
 * d_queries  (Q, D) float32 row-major with row stride ld_q                -- text_feats
 * d_scores   (N, Q) float32 row-major, or NULL to skip materialising scores_mat
 * d_argmax   (N,) int32, or NULL
 * d_best     (N,) float32 score of the argmax column, or NULL
 * precision  one of AVL_SIM_*
 */
AVL_API int avl_sim_scores(const float* d_feat, int64_t N, int D, int64_t ld_feat, const float* d_queries, int Q,
                           int64_t ld_q, float* d_scores, int32_t* d_argmax, float* d_best, int precision,
                           void* stream);

/* Scratch of the matrix-core paths: the prepared query image (avl_sim_workspace_bytes: depends on D, Q only) and, for
 * the raw split path, one range-guard word per 32 voxel rows (avl_sim_workspace_bytes_n = image + guard words).  Pass a
 * buffer of that size (it also covers avl_sim_scores_blocks' gathered queries) to avl_sim_scores_ws to keep every allocation off the hot path (benchmark / graph capture); with a
 * smaller or NULL workspace the library allocates from the stream-ordered pool. */
AVL_API int avl_sim_workspace_bytes(int D, int Q, size_t* h_bytes);
AVL_API int avl_sim_workspace_bytes_n(int64_t N, int D, int Q, size_t* h_bytes);
AVL_API int avl_sim_scores_ws(const float* d_feat, int64_t N, int D, int64_t ld_feat, const float* d_queries, int Q,
                              int64_t ld_q, float* d_scores, int32_t* d_argmax, float* d_best, int precision,
                              void* d_workspace, size_t workspace_bytes, void* stream);
/* Block-structured query sets: h_col_begin / h_col_end (Q host ints) give every query's non-zero column window [begin, end)
 * (a superset is fine).  Queries with the same window (rounded out to 128 columns) are scored against just those columns of
 * the map, one launch group per window: on a fused visual | audio map (BASELINE config 5: 512 + 1024 columns, every query
 * living in one modality block) the map is still read once overall but the matrix cores do half the work.  Results equal the
 * dense call (same products; zero products dropped); argmax ties still resolve to the lowest query index.  d_row_scale: see
 * avl_sim_scores_prepared (NULL unless precision == AVL_SIM_PREPARED on a scaled map, or AVL_SIM_PREPARED24: then d_feat is the
 * compact map of avl_sim_prepare_map24 and ld_feat == D).  Falls back to the dense path when the windows do not split the
 * queries or the shape does not allow it. */
AVL_API int avl_sim_scores_blocks(const float* d_feat, const float* d_row_scale, int64_t N, int D, int64_t ld_feat,
                                  const float* d_queries, int Q, int64_t ld_q, const int32_t* h_col_begin,
                                  const int32_t* h_col_end, float* d_scores, int32_t* d_argmax, float* d_best, int precision,
                                  void* d_workspace, size_t workspace_bytes, void* stream);
/* Scores of a map prepared WITH row scaling (d_row_scale from avl_sim_prepare_map; NULL = prepared without). */
AVL_API int avl_sim_scores_prepared(const float* d_feat, const float* d_row_scale, int64_t N, int D, int64_t ld_feat,
                                    const float* d_queries, int Q, int64_t ld_q, float* d_scores, int32_t* d_argmax,
                                    float* d_best, void* d_workspace, size_t workspace_bytes, void* stream);

/* COMPACT resident copy of a map that is indexed many times (VLMap.compact_map, its default when D % 128 == 0): 3 bytes per element
 * instead of 4 -- fp16 hi (round to nearest) and one byte per element holding the residual x - hi in units of ulp(hi) / 256, every row
 * scaled by a power of two (d_row_scale out, N floats).  Row layout (round 4; opaque to callers, it only has to come from this
 * function): a plane of hi values [0, 2 D) -- column c at byte 2 c -- then a plane of residual bytes [2 D, 3 D), column c at
 * 128 (c >> 7) + 64 ((c >> 5) & 1) + 32 ((c >> 6) & 1) + (c & 31): every load of the kernels' walk is a whole cache line shared by the
 * two halves of a wave (the first layout interleaved hi[32] | residuals[32] per 96 bytes and requested every third line twice).
 * d_map24: N * D * 3 bytes, out of place (the float32 map is left alone).  A query pass then reads a quarter less HBM; the
 * residuals are rebuilt as fp16 in registers (five vector-ALU instructions per two elements) and the arithmetic is the same three
 * fp16 MFMAs.  Accuracy: 19 significant bits per element (hi's 11 + 8: the residual's exponent is implied by hi) instead of 22 --
 * max |score - float64| 2.3e-6 on LSeg-scale rows against 1.4e-6 for the 4-byte forms (tests/test_sim_gpu.py): still float32-class.
 * Elements below 2^-11 of their row's maximum keep hi only.  D % 128 == 0.  Every matrix-core kernel reads this form: the
 * resident-query kernel (D <= 512, up to ~78 queries per pass: 0.70 -> 0.57 ms at 2 M x 512 x 64), the K-swap kernel (D <= 1024),
 * the streamed kernels (up to 128 queries per pass) and the column-block launches of avl_sim_scores_blocks (precision
 * AVL_SIM_PREPARED24; BASELINE config 5's fused 1536-column map is 9.2 GB instead of 12.3 GB per pass). */
AVL_API int avl_sim_prepare_map24(const float* d_feat, int64_t N, int D, int64_t ld_feat, void* d_map24, float* d_row_scale,
                                  void* stream);
AVL_API int avl_sim_scores_prepared24(const void* d_map24, const float* d_row_scale, int64_t N, int D, const float* d_queries,
                                      int Q, int64_t ld_q, float* d_scores, int32_t* d_argmax, float* d_best,
                                      void* d_workspace, size_t workspace_bytes, void* stream);

/* Host-buffer convenience wrapper (synchronous): copies in, runs, copies out. */
AVL_API int avl_sim_scores_host(const float* h_feat, int64_t N, int D, const float* h_queries, int Q,
                                float* h_scores, int32_t* h_argmax, float* h_best, int precision);

/* mask[i] = (argmax[i] == cat_id) as uint8 -- avlmaps/map/vlmap.py:124 */
AVL_API int avl_mask_from_argmax(const int32_t* d_argmax, int64_t N, int32_t cat_id, uint8_t* d_mask, void* stream);
/* the same mask bit-packed: d_bits (ceil(N / 64),) uint64, word w bit i = (argmax[64 w + i] == cat_id), i.e. the byte / bit order of
 * np.unpackbits(bitorder="little"); bits beyond N are 0.  1 bit instead of 4 bytes per voxel has to reach the host. */
AVL_API int avl_mask_bits_from_argmax(const int32_t* d_argmax, int64_t N, int32_t cat_id, uint64_t* d_bits, void* stream);

/* index and value of the maximum of a float32 vector, first maximum wins -- the navigator's
 * heatmap argmax, avlmaps/robot/habitat_lang_robot.py:427-430.  Synchronous (returns host scalars). */
AVL_API int avl_argmax_f32(const float* d_vals, int64_t N, int64_t* h_index, float* h_value, void* stream);

/* the k largest values with their indices, descending, ties in ascending index order (np.argsort(-v, kind="stable")[:k]);
 * h_index (k,) int64 and h_value (k,) float32 are host buffers.  Synchronous. */
AVL_API int avl_topk_f32(const float* d_vals, int64_t N, int k, int64_t* h_index, float* h_value, void* stream);

/* ------------------------------------------------------------------------------------------------
 * (2) map builder: depth back-projection + voxelisation + weighted feature fusion
 *     replaces  avlmaps/map/vlmap_builder.py:129-178 (per-frame body of create_mobile_base_map)
 *               avlmaps/utils/mapping_utils.py:226-251 depth2pc, :305-315 transform_pc,
 *               :345-349 base_pos2grid_id_3d, :599-605 project_point
 *     State lives on the device behind the handle; avl_builder_finalize reproduces the reference's
 *     arrays (grid_feat, grid_pos, weight, grid_rgb, occupied_ids) in the reference's voxel-id order.
 * ------------------------------------------------------------------------------------------------ */
typedef struct avl_builder avl_builder;

/* gs, cs, vh: grid size, cell size (m), cells in height (= int(camera_height / cs), vlmap_builder.py:201)
 * D: feature dimension; capacity: maximum number of occupied voxels the handle can hold. */
AVL_API int avl_builder_create(avl_builder** h_out, int gs, double cs, int vh, int D, int64_t capacity);
/* Rectangular grid n0 rows x n1 cols x n2 heights (the global multi-floor map: grid_size[[0, 2, 1]],
 * vlmap_builder_multi_floor.py:222-224). */
AVL_API int avl_builder_create_grid(avl_builder** h_out, int n0, int n1, int n2, double cs, int D, int64_t capacity);
AVL_API int avl_builder_destroy(avl_builder* b);
AVL_API int avl_builder_reset(avl_builder* b, void* stream);
/* The reference doubles its arrays whenever max_id reaches their length (_reserve_map_space, vlmap_builder.py:286-311).
 * max_capacity > capacity: the per-voxel accumulators double (realloc + device copy, between launches) up to max_capacity
 * voxels instead of failing with AVL_ERR_CAPACITY; 0 (the default after create) keeps the capacity fixed. */
AVL_API int avl_builder_set_max_capacity(avl_builder* b, int64_t max_capacity);
AVL_API int avl_builder_capacity(avl_builder* b, int64_t* h_capacity);
/* Deferred fuse: frame-by-frame integration in ONE launch per frame instead of two dependent ones.  With on != 0,
 * avl_builder_integrate_frame / _frame_global run the geometry + list linking of the frame they are given next to the feature
 * fusion of the PREVIOUS frame (disjoint state, same kernel); the frame's own fusion rides in the next call's launch or in
 * avl_builder_flush.  The map that results is the same, bit for bit (the samples of a voxel are summed in ascending sample
 * order in either mode; only a voxel that receives more than 64 samples in ONE launch may differ in the last feature bit, as
 * it may between two runs of any mode).  What changes for the caller: the d_feat buffer of a
 * frame is read by the launch of the NEXT integrate / flush / finalize / num_* / export call, so it must stay valid and
 * unmodified until that call has been enqueued (all on one stream).  d_depth, d_sample_idx and d_rgb are consumed by the
 * call they are passed to, as before.  Every entry point that reads the map flushes first; batched calls flush and then run
 * the three-launch path.  The reference's loop (vlmap_builder.py:102-183) has the same shape: features of frame i are
 * produced, then fused, one frame at a time. */
AVL_API int avl_builder_set_deferred_fuse(avl_builder* b, int on, void* stream);
AVL_API int avl_builder_flush(avl_builder* b, void* stream);
/* Give back what the builder's finalisation / replay / merge temporaries left cached: the slot-sorted replay log kept between the
 * two replay calls of a merge, and everything above keep_bytes in the device's stream-ordered pool (hipMemPoolTrimTo).  The
 * library keeps AVLMAPS_MEMPOOL_KEEP_MB (default 1024) MB of that pool between calls; call this after the last merge / final
 * save of a build that shares the GPU with a feature extractor.  Synchronises the stream. */
AVL_API int avl_builder_release_scratch(avl_builder* b, int64_t keep_bytes, void* stream);

/* Optional: keep a 24-byte log entry per sampled pixel (up to max_samples in total) so that avl_builder_finalize can
 * REPLAY the reference's sequential weight / grid_rgb updates exactly -- float32 weight accumulation and the truncating
 * uint8 colour store of vlmap_builder.py:166-178, including the dtype switch at _reserve_map_space (:286-311).
 * Without the log, finalize returns float32(sum alpha) and the once-truncated weighted mean colour.
 * Call on a fresh (or reset) builder.  Not used after avl_builder_import_map. */
AVL_API int avl_builder_enable_replay_log(avl_builder* b, int64_t max_samples);

/*
 * Fuse one RGB-D frame.
 *   d_depth        (H, W) float32 metres
 *   h_calib        3x3 float64 row-major camera matrix  (map_config.cam_calib_mat)
 *   h_calib_inv    3x3 float64 = numpy.linalg.inv(calib) as the reference computes it (mapping_utils.py:237)
 *   h_pc_transform 4x4 float64 row-major camera->map transform (vlmap_builder.py:133)
 *   d_sample_idx   (P,) int32 flattened pixel indices in the reference's sampling order
 *                  (shuffle_mask[::depth_sample_rate], vlmap_builder.py:275-277), BEFORE depth masking
 *   d_feat         (Hf, Wf, D) float32 CHANNELS-LAST pixel features (the reference holds (1, D, Hf, Wf))
 *   d_rgb          (H, W, 3) uint8
 *   frame_idx      position of the frame in the sequence (defines first-touch order across frames)
 *   min_depth/max_depth  strict bounds on camera-frame z (0.1, 6: vlmap_builder.py:129)
 *   sigma_sq       0.6 (vlmap_builder.py:157)
 */
AVL_API int avl_builder_integrate_frame(avl_builder* b, const float* d_depth, int H, int W, const double* h_calib,
                                        const double* h_calib_inv, const double* h_pc_transform,
                                        const int32_t* d_sample_idx, int P, const float* d_feat, int Hf, int Wf,
                                        const uint8_t* d_rgb, int64_t frame_idx, double min_depth,
                                        double max_depth, double sigma_sq, void* stream);

/*
 * Fuse B consecutive frames (frame_idx0 .. frame_idx0 + B - 1) with ONE launch pair.  Same semantics and results as B calls
 * of avl_builder_integrate_frame; the samples of a voxel coming from different frames share one list, so the voxel row is
 * read-modified-written once per batch instead of once per frame, and the per-launch latencies are amortised.
 * All frames share H, W, P, Hf, Wf and the camera matrix.  h_*_ptrs are HOST arrays of B DEVICE pointers
 * (depth (H,W) f32, samples (P,) i32, feat (Hf,Wf,D) f32 channels-last, rgb (H,W,3) u8); h_pc_transforms is (B, 16) float64.
 * The per-frame device buffers must stay valid until the launches have run (stream order).
 */
AVL_API int avl_builder_integrate_batch(avl_builder* b, int B, const float* const* h_depth_ptrs, int H, int W,
                                        const double* h_calib, const double* h_calib_inv, const double* h_pc_transforms,
                                        const int32_t* const* h_sample_ptrs, int P, const float* const* h_feat_ptrs, int Hf,
                                        int Wf, const uint8_t* const* h_rgb_ptrs, int64_t frame_idx0, double min_depth,
                                        double max_depth, double sigma_sq, void* stream);

/*
 * The frame-by-frame loop itself (vlmap_builder.py:102-183's `for frame_i in ...`), for callers whose frames are already resident:
 * exactly n_frames calls of avl_builder_integrate_frame -- one launch pair per frame, or one launch with deferred fuse; NOT a
 * batch: no two frames share a launch or a list -- issued from C.  A call through a language binding costs more host time than the
 * frame's kernels take (12.4 us per ctypes call against 11.9 us of pipe_kernel, tools/probe_frame_loop.py).  Arguments as for
 * avl_builder_integrate_batch; with deferred fuse the LAST frame's features are still to be fused when the call returns.
 * Every frame of the call must be resident when it is made (as for a batch): while frame i is launched, a few workgroups of the
 * same launch already read frame i + 1's sample indices, depth and colour image and run the half of its back-projection that
 * does not depend on the map (geometry, projections, colour gather, weight), so that frame i + 1's dependent chain starts at the
 * voxel-hash lookup.  Same arithmetic, same maps; the frames of one launch still share nothing.
 */
AVL_API int avl_builder_integrate_frames(avl_builder* b, int n_frames, const float* const* h_depth_ptrs, int H, int W,
                                        const double* h_calib, const double* h_calib_inv, const double* h_pc_transforms,
                                        const int32_t* const* h_sample_ptrs, int P, const float* const* h_feat_ptrs, int Hf,
                                        int Wf, const uint8_t* const* h_rgb_ptrs, int64_t frame_idx0, double min_depth,
                                        double max_depth, double sigma_sq, void* stream);

/*
 * Global (multi-floor) variant of the frame fusion: replaces vlmap_builder_multi_floor.py:137-199.
 *   voxel index = np.round((p_global - pcd_min) / cs) per axis, (row, height, col) = (x, y, z) (:146);
 *   d_depth is float32 metres, or uint16 with metres = value / depth_div (depth PNGs / 1000.0, :105);
 *   h_transform = camera_pose_tf @ habitat2cam_rot_tf (:141); h_pcd_min = lower bounding-box corner from pass 1.
 * Samples that fall outside the grid are dropped (the reference wraps negative indices or raises).
 */
AVL_API int avl_builder_integrate_frame_global(avl_builder* b, const void* d_depth, int depth_is_u16, double depth_div, int H,
                                               int W, const double* h_calib, const double* h_calib_inv,
                                               const double* h_transform, const int32_t* d_sample_idx, int P,
                                               const float* d_feat, int Hf, int Wf, const uint8_t* d_rgb, int64_t frame_idx,
                                               double min_depth, double max_depth, double sigma_sq, const double* h_pcd_min,
                                               void* stream);

/* Pass 1 of the global builder (vlmap_builder_multi_floor.py:97-118): fold the transformed, depth-masked sampled points of
 * one frame into h_minmax = [min x, min y, min z, max x, max y, max z] (in/out; start from +inf / -inf).  Synchronous. */
AVL_API int avl_points_bbox(const void* d_depth, int depth_is_u16, double depth_div, int H, int W, const double* h_calib_inv,
                            const double* h_transform, const int32_t* d_sample_idx, int P, double min_depth, double max_depth,
                            double* h_minmax, void* stream);

/* Seed an EMPTY builder from a finished map so that more frames can be fused on top (the reference's resume path,
 * vlmap_builder.py:212-222): d_grid_feat (n,D) f32, d_grid_pos (n,3) i32, d_weight (n,) f32, d_grid_rgb (n,3) u8 or NULL.
 * Voxel ids 0..n-1 are kept; new voxels are appended after them.  The accumulators grow to n first when the handle may
 * (avl_builder_set_max_capacity): a map that outgrew gs*gs voxels resumes like upstream's.  n == 0 only marks the builder as
 * continuing a map (the other ranks of a resumed multi-GPU build: same first-touch key space as the rank that imported).
 * Synchronous. */
AVL_API int avl_builder_import_map(avl_builder* b, int64_t n, const float* d_grid_feat, const int32_t* d_grid_pos,
                                   const float* d_weight, const uint8_t* d_grid_rgb, void* stream);

/* number of occupied voxels so far (synchronises the stream) */
AVL_API int avl_builder_num_voxels(avl_builder* b, int64_t* h_n, void* stream);
/* number of sampled points that updated a voxel so far (synchronises the stream) */
AVL_API int avl_builder_num_points(avl_builder* b, int64_t* h_n, void* stream);
/* number of (frame, voxel) groups fused so far = voxel rows read-modified-written (synchronises the stream) */
AVL_API int avl_builder_num_groups(avl_builder* b, int64_t* h_n, void* stream);

/*
 * Produce the reference's arrays.  Slot ids are assigned in first-touch order (a deterministic prefix
 * scan over each frame's sample list), so row r of every output IS the reference's voxel id r:
 *   d_grid_feat (n, D) f32, d_grid_pos (n, 3) i32, d_weight (n,) f32, d_grid_rgb (n, 3) u8,
 *   d_occupied_ids (gs, gs, vh) i32 (-1 = empty) -- any of them may be NULL.
 * n must equal avl_builder_num_voxels().  Synchronous.
 */
AVL_API int avl_builder_finalize(avl_builder* b, int64_t n, float* d_grid_feat, int32_t* d_grid_pos, float* d_weight,
                                 uint8_t* d_grid_rgb, int32_t* d_occupied_ids, void* stream);

/* avl_builder_finalize plus, for incremental checkpoints: d_row_dirty (n,) uint8 (nullable) receives 1 for every output row whose
 * voxel was fused since the flags were last cleared, and clear_dirty != 0 clears them.  Voxel ids never change once assigned
 * (new voxels are appended), so a checkpoint only has to rewrite the dirty rows and append rows >= the previous n -- upstream
 * rewrites the whole file every 100 frames (vlmap_builder.py:180-183). */
AVL_API int avl_builder_finalize_ex(avl_builder* b, int64_t n, float* d_grid_feat, int32_t* d_grid_pos, float* d_weight,
                                    uint8_t* d_grid_rgb, int32_t* d_occupied_ids, uint8_t* d_row_dirty, int clear_dirty,
                                    void* stream);

/*
 * Multi-GPU merge support (frames sharded over ranks; the exchange itself runs in the host layer over
 * RCCL, see avlmaps_amd/parallel.py).  Export the raw per-voxel accumulators of the first n slots:
 *   d_cell        (n,)   int32   linear cell index (row*gs + col)*vh + h
 *   d_first_key   (n,)   uint64  first-touch key (frame_idx << 32 | position in the frame's sample list)
 *   d_sum_feat    (n, D) float64 sum_i alpha_i * f_i
 *   d_sum_w4      (n, 4) float64 [sum alpha, sum alpha*r, sum alpha*g, sum alpha*b]
 *   d_first_feat  (n, D) float32 feature of the first-touch point
 *   d_first_alpha (n,)   float64 alpha of the first-touch point
 * Any output may be NULL.
 */
AVL_API int avl_builder_export_raw(avl_builder* b, int64_t n, int32_t* d_cell, uint64_t* d_first_key,
                                   double* d_sum_feat, double* d_sum_w4, float* d_first_feat,
                                   double* d_first_alpha, void* stream);

/*
 * Stateless finalisation of raw accumulators (the arrays of avl_builder_export_raw, possibly merged
 * across ranks and re-ordered by first-touch key): applies the reference's first-touch weighting
 *   grid_feat = (sum_feat - a1*(1-a1)*first_feat) / sum_alpha      (vlmap_builder.py:166-174 closed form)
 * and writes grid_feat (n,D) f32, grid_pos (n,3) i32, weight (n,) f32, grid_rgb (n,3) u8 in the given row
 * order, and occupied_ids[cell] = row index (d_occupied_ids (gs,gs,vh) must be pre-filled with -1).
 * Any output may be NULL.
 */
AVL_API int avl_finalize_raw(int64_t n, int D, int gs, int vh, const int32_t* d_cell, const double* d_sum_feat,
                             const double* d_sum_w4, const float* d_first_feat, const double* d_first_alpha,
                             float* d_grid_feat, int32_t* d_grid_pos, float* d_weight, uint8_t* d_grid_rgb,
                             int32_t* d_occupied_ids, void* stream);

/*
 * Multi-GPU merge on the device (avlmaps_amd/parallel.py drives the RCCL calls).  After the ranks have agreed
 * on the union of occupied cells and all-reduced (MIN) the first-touch keys, every rank scatters its accumulators into a
 * zero-initialised dense (M, ld_acc >= D + 4) float64 buffer that is then sum-reduced ONCE:
 *   row d_row_of_slot[s] <- [sum_feat[s] + (own ? a1^2 : a1) first_feat[s]  |  sum alpha, sum alpha * (r, g, b)]
 * where sum_feat[s] sums alpha f over every sample of the slot EXCEPT its local first touch (a1, first_feat) and
 * own = (this rank's first-touch key of the voxel == d_global_key[row]): the owner of the global first touch contributes it
 * with the reference's weight a1^2 (vlmap_builder.py:166-174), every other rank with a1, so no first_feat / first_alpha
 * exchange is needed.
 * n must equal avl_builder_num_voxels().  d_row_of_slot (n,) int64, d_global_key (M,) uint64.
 */
AVL_API int avl_builder_scatter_merge(avl_builder* b, int64_t n, const int64_t* d_row_of_slot, const uint64_t* d_global_key,
                                      double* d_acc, int64_t ld_acc, void* stream);
/* Finalise n rows of reduced accumulators laid out as above (d_acc points at the first of them, d_cell (n,) are their linear
 * cells): grid_feat = acc[:, :D] / sum alpha etc.; output row r is voxel id row0 + r (occupied_ids[cell] = row0 + r), so one
 * rank can finalise one block of a reduce-scattered map.  Any output may be NULL; d_occupied_ids must be pre-filled with -1. */
AVL_API int avl_finalize_merged(int64_t n, int64_t row0, int D, int gs, int vh, const int32_t* d_cell, const double* d_acc,
                                int64_t ld_acc, float* d_grid_feat, int32_t* d_grid_pos, float* d_weight, uint8_t* d_grid_rgb,
                                int32_t* d_occupied_ids, void* stream);
/* Exact sequential weight / grid_rgb across ranks (needs the replay log): the per-voxel state {float64 weight, float32 rgb[3],
 * uint32 started} = 24 bytes, d_state (M, 24 B) zero-initialised on the first rank, is continued with this rank's log for its
 * own voxels (row d_row_of_slot[s]; a NEGATIVE index leaves slot s out of this call: the round-4 merge replays the voxels that do not
 * depend on another rank first and the others once their predecessor's state has arrived) and handed to the next rank, in rank
 * order (contiguous frame shards).  grow_key = first-touch
 * key of the voxel with global id gs*gs - 1 (after it the reference's arrays have been re-allocated with other dtypes,
 * vlmap_builder.py:286-311), or ~0 if the map is smaller.  avl_replay_state_apply writes weight / grid_rgb from the final state. */
AVL_API int avl_builder_replay_chain(avl_builder* b, int64_t n, const int64_t* d_row_of_slot, uint64_t grow_key, void* d_state,
                                     void* stream);
/* The log sorted by voxel is kept between avl_builder_replay_chain calls until the next frame is fused (a merge calls it twice);
 * this drops it, so that a caller timing a merge twice on the same map pays for the sort both times (bench.py). */
AVL_API int avl_builder_drop_replay_cache(avl_builder* b, void* stream);
/* build that voxel-sorted log NOW (n = avl_builder_num_voxels) instead of inside the next avl_builder_replay_chain / finalize; a no-op without
 * a replay log */
AVL_API int avl_builder_replay_prepare(avl_builder* b, int64_t n, void* stream);
/* ------------------------------------------------------------------------------------------------
 * The local steps of the multi-GPU merge plan (avlmaps_amd/parallel.py, plan_merge_directory): which final row -- the reference's
 * voxel id, vlmap_builder.py:163-170 -- every voxel of every rank gets.  torch.distributed carries the collectives between them;
 * each entry point replaces the 20-40 small tensor operations a rank ran between two collectives by a radix sort over the bits
 * the keys have and one or two kernels (no host synchronisation inside).  At most 64 ranks.  d_work: caller-owned scratch of
 * avl_merge_work_bytes(max(n, R)) bytes.
 *   avl_merge_partition  local voxels -> directory ranks: d_ordd[n] = the slots grouped by directory rank (stable), d_cell_sorted[n] =
 *                        their cells in that order (what the first all_to_all sends), d_head[ws + 3] = [voxels per directory
 *                        rank | n | smallest first-touch key | largest]
 *   avl_merge_dir_scan   directory side, the R cells that arrived (grouped by source rank; d_rc[ws] = how many from each): d_perm[R] =
 *                        their (cell, source rank) order, d_first[R] = "first entry of its cell" in that order, and in ARRIVAL order
 *                        d_prev_r / d_next_r[R] = the neighbouring contributors of the entry's cell (-1: none), d_reply[R] = both
 *                        packed for the way back ((prev + 1) | (next + 1) << 16), d_m3r / d_m4r[R] = the masks of the two later
 *                        round trips; d_cnt[2 ws + 1] = [m3r per source | m4r per source | distinct cells]
 *   avl_merge_classify   back home, d_back[n] = the replies in sending order: d_prev / d_next[n] per slot, d_is_new[n] per slot,
 *                        d_m3 / d_m4[n] in sending order, d_cnt[1 + 4 ws] = [new voxels | m3 per directory rank | m4 per directory
 *                        rank | voxels per prev rank | voxels per next rank]
 *   avl_merge_rows_new   after the count all_gather (c = this rank's new voxels, base = its first final row): d_idx_new[c] = the new
 *                        voxels' slots in first-touch-key order (radix sort over key_bits), d_row[n] = base + position for them and
 *                        -1 for the others, d_send3[n3] = the rows of the new voxels other ranks share, in sending order (d_m3)
 *   avl_merge_dir_rows   directory side: d_recv3 = the rows that arrived for the d_m3r entries (arrival order); every entry of a cell
 *                        inherits the row of the cell's first contributor; d_send4[n4] = those rows for the d_m4r entries
 *   avl_merge_rows_other home: d_recv4[n4] = the rows of this rank's voxels whose first touch another rank holds (d_m4, sending
 *                        order) -> d_row
 *                        (scratch of these three: avl_merge_rows_work_bytes(max(n, R)))
 * ------------------------------------------------------------------------------------------------ */
AVL_API int avl_merge_work_bytes(int64_t n, size_t* h_bytes);
AVL_API int avl_merge_rows_work_bytes(int64_t n, size_t* h_bytes);
AVL_API int avl_merge_rows_new(int64_t n, int64_t c, const uint8_t* d_is_new, const int64_t* d_key, int key_bits, int64_t base,
                               const int64_t* d_ordd, const uint8_t* d_m3, int64_t n3, int64_t* d_row, int64_t* d_idx_new, int64_t* d_send3,
                               void* d_work, size_t work_bytes, void* stream);
AVL_API int avl_merge_dir_rows(int64_t R, const int64_t* d_recv3, const uint8_t* d_m3r, const uint8_t* d_first, const int64_t* d_perm,
                               const uint8_t* d_m4r, int64_t n4, int64_t* d_send4, void* d_work, size_t work_bytes, void* stream);
AVL_API int avl_merge_rows_other(int64_t n, const uint8_t* d_m4, const int64_t* d_ordd, const int64_t* d_recv4, int64_t n4, int64_t* d_row,
                                 void* d_work, size_t work_bytes, void* stream);
AVL_API int avl_merge_partition(int64_t n, const int32_t* d_cell, const int64_t* d_key, int ws, int64_t* d_ordd, int32_t* d_cell_sorted,
                                int64_t* d_head, void* d_work, size_t work_bytes, void* stream);
AVL_API int avl_merge_dir_scan(int64_t R, const int32_t* d_recv, const int64_t* d_rc, int ws, int cell_bits, int64_t* d_perm, uint8_t* d_first,
                               int64_t* d_prev_r, int64_t* d_next_r, int32_t* d_reply, uint8_t* d_m3r, uint8_t* d_m4r, int64_t* d_cnt,
                               void* d_work, size_t work_bytes, void* stream);
AVL_API int avl_merge_classify(int64_t n, const int32_t* d_back, const int64_t* d_ordd, const int32_t* d_cell_sorted, int ws, int64_t* d_prev,
                               int64_t* d_next, uint8_t* d_is_new, uint8_t* d_m3, uint8_t* d_m4, int64_t* d_cnt, void* stream);

/* The 64-byte side record every local voxel sends to the owner of its final row (avlmaps_amd/parallel.py: MixedExchange):
 * [row | cell << 32 | single << 63, sum_w4 (4 x f64), replay state (3 x i64)].
 *   avl_merge_side_pack    sender: record i (final-row order) from voxel d_order[i]: d_rows_sorted[i], d_single_sorted[i] (by position),
 *                          d_cell / d_w4 (n x 4) / d_state (n x 3, nullable = zeros; sent only where d_next[voxel] < 0,
 *                          d_next nullable = everywhere) (by voxel) -> d_side (n x 8 int64)
 *   avl_merge_side_unpack  owner: R records that arrived -> d_rows[i] = row - r0 (clamped into the block), d_own_cell[row] = cell,
 *                          d_state[row] = the state of the voxel's LAST contributor (`started` != 0); d_own_cell (n_own) and
 *                          d_state (n_own x 3) are zeroed by the caller.  A row outside [r0, r0 + n_own) is skipped and sets
 *                          bit 0 of *d_err_flag (nullable).
 * The reference has no such step (one process, one map: vlmap_builder.py:136-178); it is part of the N-rank merge of north_star. */
AVL_API int avl_merge_side_pack(int64_t n, const int64_t* d_order, const int64_t* d_rows_sorted, const uint8_t* d_single_sorted,
                                const int32_t* d_cell, const double* d_w4, const int64_t* d_state, const int64_t* d_next, int64_t* d_side,
                                void* stream);
AVL_API int avl_merge_side_unpack(int64_t R, const int64_t* d_side, int64_t r0, int64_t n_own, int64_t* d_rows, int32_t* d_own_cell,
                                  int64_t* d_state, int* d_err_flag, void* stream);

/* The N-rank merge of the map build, second form ("gather plan"; avlmaps_amd/merge2.py carries the collectives, csrc/avl_merge2.hip
 * holds the kernels).  The reference has no such step (one process, one map: vlmap_builder.py:102-183 is the loop being sharded); a
 * voxel's final row is the reference's voxel id = its position in first-touch order (vlmap_builder.py:163-170).
 *   avl_merge2_prepare  a rank's own list, sorted by first-touch key (keys below 2^key_bits): d_key_sorted, d_cell_sorted, d_perm (n x i32:
 *                     position in that order -> voxel slot), d_hdr[4] = n, smallest key, largest key, flags (the header every rank
 *                     all_gathers first).
 *   avl_merge2_plan   EVERY rank, after the all_gather of the ranks' key-sorted (first-touch key, cell) lists: d_gathered holds ws chunks
 *                     of nmax + (nmax + 1) / 2 int64 words each -- [keys (nmax x i64) | cells (nmax x i32)], h_n_all[p] entries valid in
 *                     chunk p; keys must be ordered by rank (contiguous frame shards), so that the concatenation is in key order and a
 *                     voxel's row is the number of first contributors before it; cells below 2^cell_bits; d_perm: this rank's own
 *                     avl_merge2_prepare permutation.  Work buffer: avl_merge2_work_bytes(sum n, n of this rank, ws, nchunk).  Results (byte offsets into
 *                     d_work returned in h_off[11]): 0 row (n x i32, final row of own voxel s), 1 prev, 2 next (n x i32: the
 *                     neighbouring contributors of the voxel in rank order, -1 = none), 3 order (n x i32: own voxels in final-row
 *                     order), 4 sidx (n x i32: single-rank voxels before position i of that order), 5 selA, 6 selB (n x i64:
 *                     avl_builder_replay_chain selections: s where prev < 0 / prev >= 0, else -1), 7 idx_prev, 8 idx_next
 *                     (n x i32: own voxels grouped by prev / next rank, final-row order inside a group, voxels without one last;
 *                     only with want_replay_lists), 9 rowcell (M x i32: cell of every final row), 10 the device copy of h_res.
 *                     h_res[2 + 3 ws^2] (host; the call synchronises once to deliver it): M, the first-touch key of row grow_row
 *                     (-1 if M <= grow_row), then three ws x ws tables [sender p][receiver q]: voxels of p whose row q owns, the
 *                     single-rank ones among them, voxels of q whose previous contributor is p.
 *                     With nchunk > 0 (<= avl_merge2_max_chunks; chunk_rows * nchunk must cover ceil(sum n / ws) rows) h_res continues
 *                     with nchunk x two ws x ws tables: the first two tables restricted to rows [q per + c chunk_rows, q per + (c + 1)
 *                     chunk_rows) of every owner q's block (per = ceil(M / ws)) -- the sizes of an exchange done chunk by chunk.
 *   avl_builder_m2_pack   sender, ONE exchange (the whole payload or one chunk of it): the n voxels of the call in destination order --
 *                     wave w serves rank q with h_cum[q] <= w < h_cum[q + 1] and is voxel h_lo[q] + (w - h_cum[q]) of d_order (the
 *                     rank's final-row order; h_dlo[q] single-rank voxels precede h_lo[q] there) -> the send buffer (int64 words):
 *                     64-byte side records at word h_side_off[q] ([row - h_row0[q] | list index << 32 | single << 63 | direct << 62,
 *                     sum_w4 (4 x f64), 3 zero words]), finished float32 rows of single-rank voxels at h_done_off[q] (row stride
 *                     (D + 1) / 2 words), float64 partial rows of shared voxels at h_part_off[q].  d_own_feat (nullable): this rank's
 *                     block of grid_feat (first row own_r0) -- single-rank voxels whose row it owns are written there directly (`direct`).
 *   avl_merge2_side_state  after the replay: words 5..7 of the call's side records (same h_cum / h_lo) = the replay state (d_state n x 3
 *                     i64, nullable) where this rank is the voxel's last contributor (d_next < 0), zeros elsewhere.
 *   avl_merge2_state_gather / _scatter   the 24-byte replay states of the listed voxels <-> a contiguous hop buffer.
 *   avl_merge2_fold   owner: the block [r0, r0 + n_own) from what the peers sent (h_side / h_done / h_part: per peer the device
 *                     addresses of its three lists, h_count[p] records; the rank's own lists stay in its send buffer): contributors
 *                     summed in rank order, grid_feat / grid_pos / weight / grid_rgb / cell of the block written.  d_work: 256-aligned scratch of
 *                     avl_merge2_fold_work_bytes (which record of peer p belongs to row r, sum alpha and a to-do flag per row).  Bits of *d_err_flag: 1 = a record outside the
 *                     block, 2 = a row nobody sent, 4 = a single-rank record next to other contributors. */
AVL_API int avl_merge2_load(void);   /* load the merge's code object now (avl_builder_create does): not inside the first merge */
AVL_API int avl_merge2_prepare_work_bytes(int64_t n, size_t* h_bytes);
AVL_API int avl_merge2_prepare(int64_t n, const int64_t* d_key, const int32_t* d_cell, int key_bits, int64_t flags, int64_t* d_key_sorted,
                               int32_t* d_cell_sorted, int32_t* d_perm, int64_t* d_hdr, void* d_work, size_t work_bytes, void* stream);
AVL_API int avl_merge2_max_chunks(int ws, int* h_max);    /* how many chunks of an owner's block the plan can size (tables in LDS) */
AVL_API int avl_merge2_work_bytes(int64_t n_entries, int64_t n_own, int ws, int nchunk, size_t* h_bytes);
AVL_API int avl_merge2_plan(int ws, int rank, const int64_t* h_n_all, int64_t nmax, const int64_t* d_gathered, const int32_t* d_perm,
                            int cell_bits, int64_t grow_row, int want_replay_lists, int64_t chunk_rows, int nchunk, void* d_work,
                            size_t work_bytes, int64_t* h_off, int64_t* h_res, void* stream);
AVL_API int avl_builder_m2_pack(avl_builder* b, int64_t n, int ws, int rank, int64_t own_r0, const int64_t* h_cum, const int64_t* h_lo,
                                const int64_t* h_dlo, const int64_t* h_row0, const int64_t* h_side_off, const int64_t* h_done_off,
                                const int64_t* h_part_off, const int32_t* d_order, const int32_t* d_row, const int32_t* d_prev,
                                const int32_t* d_next, const int32_t* d_sidx, int64_t* d_send, float* d_own_feat, void* stream);
AVL_API int avl_merge2_side_state(int64_t n, int ws, const int64_t* h_cum, const int64_t* h_lo, const int64_t* h_side_off,
                                  const int32_t* d_order, const int32_t* d_next, const int64_t* d_state, int64_t* d_send, void* stream);
AVL_API int avl_merge2_state_gather(int64_t k, const int32_t* d_idx, const int64_t* d_state, int64_t* d_out, void* stream);
AVL_API int avl_merge2_state_scatter(int64_t k, const int32_t* d_idx, const int64_t* d_in, int64_t* d_state, void* stream);
AVL_API int avl_merge2_fold(int64_t n_own, int64_t r0, int ws, int D, int gs, int vh, const void* const* h_side, const void* const* h_done,
                            const void* const* h_part, const int64_t* h_count, const int32_t* d_rowcell, int have_log, float* d_grid_feat,
                            int32_t* d_grid_pos, float* d_weight, uint8_t* d_grid_rgb, int32_t* d_cell, void* d_work, size_t work_bytes,
                            int32_t* d_err_flag, void* stream);
AVL_API int avl_merge2_fold_work_bytes(int64_t n_own, int ws, size_t* h_bytes);

/* Shared rows of a rank's block of the merged map: row d_rows[i] of d_out (n_out x D float32) = (float)(d_acc[i, :] / d_w4[d_rows[i], 0]),
 * i < k -- the division of finalize (vlmap_builder.py:172-174's running mean in closed form) applied to the float64 sums several
 * ranks contributed to (avlmaps_amd/parallel.py).  An index outside [0, n_out) is skipped and sets bit 0 of *d_err_flag (nullable). */
AVL_API int avl_rows_div_f32(int64_t k, int D, const double* d_acc, const int64_t* d_rows, const double* d_w4, float* d_out, int64_t n_out,
                             int32_t* d_err_flag, void* stream);
AVL_API int avl_replay_state_apply(int64_t n, const void* d_state, float* d_weight, uint8_t* d_grid_rgb, void* stream);
/* Mixed payload of the row-sharded merge (round 4).  A voxel only ONE rank touched is finished where its accumulators live:
 * d_out[i, :D] = (float)((a1^2 first_feat[s] + sum_feat[s]) / sum alpha), s = d_slots[i] -- the float64 expression of the
 * single-process finalisation (vlmap_builder.py:166-178 closed form), so the row is bit-identical to the single-process map and
 * travels as 4 B per element.  Voxels several ranks touched ship float64 partial sums: d_out[i, :D] = sum_feat[s] +
 * (d_own[i] ? a1^2 : a1) first_feat[s] (d_own[i] != 0: this rank holds the voxel's global first touch).  d_slots (k,) int32 distinct slots. */
AVL_API int avl_builder_export_rows_f32(avl_builder* b, int64_t k, const int32_t* d_slots, float* d_out, int64_t ld, void* stream);
AVL_API int avl_builder_export_rows_f64(avl_builder* b, int64_t k, const int32_t* d_slots, const uint8_t* d_own, double* d_out,
                                        int64_t ld, void* stream);
/* grid_pos / weight / grid_rgb / occupied_ids of n merged rows from their linear cells and their [sum alpha, sum alpha * (r, g, b)]
 * float64 quadruples d_w4 (n, 4) alone (avl_finalize_merged without the feature part); output row r is voxel id row0 + r. */
AVL_API int avl_finalize_side(int64_t n, int64_t row0, int gs, int vh, const int32_t* d_cell, const double* d_w4, int32_t* d_grid_pos,
                              float* d_weight, uint8_t* d_grid_rgb, int32_t* d_occupied_ids, void* stream);
/* Row-sharded merge (parallel.merge_accumulator_sharded; SURVEY.md 8e "reduce-scatter by voxel range"): rank r owns the final
 * rows [row0, row0 + nrows) and receives from every peer only that peer's contributions to them -- n rows of `cols` = D + 4
 * float64 with their final row index d_rows (n,) int64, a peer holds a voxel at most once so the indices of one call are
 * distinct.  d_dst[d_rows[i] - row0, :cols] += d_src[i, :cols].  Calls in peer order give a reproducible sum.  Synchronous
 * (reports an out-of-range row as AVL_ERR_INVALID). */
AVL_API int avl_rows_add_f64(int64_t n, int cols, const int64_t* d_rows, int64_t row0, int64_t nrows, const double* d_src,
                             int64_t ld_src, double* d_dst, int64_t ld_dst, void* stream);
/* The same fold without the host round trip: an out-of-range row index sets *d_err_flag (device int32, zeroed by the caller) to 1
 * and is skipped; the caller reads the flag once after the last peer (the round-4 merge folds up to ws lists per exchange). */
AVL_API int avl_rows_add_f64_async(int64_t n, int cols, const int64_t* d_rows, int64_t row0, int64_t nrows, const double* d_src,
                                   int64_t ld_src, double* d_dst, int64_t ld_dst, int32_t* d_err_flag, void* stream);

/* ------------------------------------------------------------------------------------------------
 * (3) nearest-target distance-decay heatmap
 *     replaces  avlmaps/utils/visualize_utils.py:29-49 get_heatmap_from_mask_3d
 *     heat[i] = 1 for mask[i] != 0, else clip(1 - min_t ||pos_t - pos_i|| / cell_size * decay, 0, 1)
 * ------------------------------------------------------------------------------------------------ */
AVL_API int avl_heatmap_from_mask(const int32_t* d_grid_pos, const uint8_t* d_mask, int64_t N, double cell_size,
                                  double decay_rate, float* d_heat, void* stream);
/* The same heat for a map that is queried again and again (AVLMap.index_object on one map, many object names): a PLAN holds what
 * depends on the voxel positions only -- the bounding box, the voxels in CELL order (one radix sort) and the zeroed bit-grid
 * buffers -- so that a call is a memset, a scatter of the targets and the window scan walked in cell order (the lanes of a wave
 * are then spatial neighbours and their column loads coalesce: the scan is 3x faster than in voxel-id order, and no host
 * round trip for the bounding box remains).  Results are the same bits as avl_heatmap_from_mask.  d_grid_pos is caller-owned
 * and must outlive the plan; one plan = one device = one host thread at a time.  AVL_ERR_INVALID when the bounding box has
 * >= 2^32 cells (use the stateless call). */
typedef struct avl_heat_plan avl_heat_plan;
AVL_API int avl_heat_plan_create(avl_heat_plan** h_out, const int32_t* d_grid_pos, int64_t N, void* stream);
AVL_API int avl_heat_plan_destroy(avl_heat_plan* plan);
AVL_API int avl_heatmap_from_mask_planned(avl_heat_plan* plan, const uint8_t* d_mask, double cell_size, double decay_rate,
                                          float* d_heat, void* stream);

/* ------------------------------------------------------------------------------------------------
 * (4) top-down 2-D products of the voxel map (the consumers that loop over all N voxels in Python upstream).
 *     All take device pointers; images are row-major uint8 with 1 = true.  Synchronous (they report index errors).
 *     Voxel positions index the image like NumPy does: negative values wrap once, anything else out of range is the
 *     reference's IndexError -> AVL_ERR_INVALID.
 * ------------------------------------------------------------------------------------------------ */
/* avlmaps/utils/visualize_utils.py:77-83 pool_3d_label_to_2d: mask2d[row, col] |= mask[i].  d_mask2d (gs, gs) is cleared first. */
AVL_API int avl_pool_label_2d(const int32_t* d_grid_pos, const uint8_t* d_mask, int64_t N, int gs, uint8_t* d_mask2d, void* stream);
/* avlmaps/map/map.py:106-113 generate_rgb_topdown_map: rgb2d[row, col] = grid_rgb[i], the LAST voxel (largest i) of a
 * column wins as in the sequential loop; untouched cells are 0.  d_rgb2d (gs, gs, 3). */
AVL_API int avl_rgb_topdown(const int32_t* d_grid_pos, const uint8_t* d_grid_rgb, int64_t N, int gs, uint8_t* d_rgb2d, void* stream);
/* avlmaps/map/map.py:79-95 generate_obstacle_map: free[r, c] = no voxel id > 0 among heights [h_begin, h_end) of
 * occupied_ids (n0, n1, vh) -- the index range of `(heights > h_min) & (heights < h_max)`, which the host evaluates in
 * float64 exactly as upstream.  Asynchronous. */
AVL_API int avl_obstacle_map(const int32_t* d_occupied_ids, int n0, int n1, int vh, int h_begin, int h_end, uint8_t* d_free,
                             void* stream);
/* avlmaps/utils/index_utils.py:163-177 (body of get_dynamic_obstacles_map_3d after the argmax): voxels whose class
 * d_class[i] (the similarity kernel's argmax, still on the device) has h_class_is_obstacle[class] != 0 mark
 * (row - rmin, col - cmin) of the (H, W) crop; out_free = !(marked & (cropped_free == 0)). */
AVL_API int avl_obstacle_scatter(const int32_t* d_grid_pos, const int32_t* d_class, int64_t N, const uint8_t* h_class_is_obstacle,
                                 int Q, int rmin, int cmin, int H, int W, const uint8_t* d_cropped_free, uint8_t* d_out_free,
                                 void* stream);

/* ------------------------------------------------------------------------------------------------
 * (6) sliding-window merge of the pixel-feature extractor's output (device-resident, channels-last)
 *     replaces  avlmaps/utils/lseg_utils.py:61-102 (window accumulation, count normalisation, crop, and the per-frame
 *               device-to-host copy of the (1, D, Hf, Wf) result)
 *     d_win     (G, D, crop, crop) row-major: the model's output for the batch of G windows, float32 (is_f16 = 0) or float16
 *     h_origin  (G, 2) host int32: (h0, w0) of every window on the padded canvas, in the reference's loop order (idh major)
 *     d_out     (height, width, D) float32 channels-last -- the layout avl_builder_integrate_frame gathers from:
 *               out[y, x, :] = (sum over the windows g covering (y, x), in window order, of win[g, :, y - h0, x - w0]) / count
 *     G <= 64.  Every pixel of (height, width) must be covered by a window (AVL_ERR_INVALID otherwise).
 * ------------------------------------------------------------------------------------------------ */
AVL_API int avl_lseg_merge_windows(const void* d_win, int is_f16, int G, int D, int crop, const int32_t* h_origin, int height,
                                   int width, float* d_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AVLMAPS_HIP_H */
