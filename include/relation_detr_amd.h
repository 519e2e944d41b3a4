/*
 * relation_detr_amd.h -- C ABI of librelation_detr_amd.so (gfx950 / MI355X).
 *
 * Drop-in boundary for the Relation-DETR hot path.  Each entry point names the reference
 * interface it replaces (paths relative to the reference repository root).
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer (hipMalloc / torch tensor.data_ptr()) unless marked host;
 *   - tensors are contiguous, row-major; the caller allocates every output;
 *   - the library never allocates, never synchronises and keeps no global state: a call only
 *     enqueues kernels on `stream` (a hipStream_t passed as void*; NULL = the default stream),
 *     so calls are re-entrant across streams and capturable in a hipGraph;
 *   - return value: RDETR_OK (0) or a negative rdetr_status; rdetr_status_string() explains it.
 *     Unlike the reference (kernel-launch errors are printf'd and ignored,
 *     models/bricks/ops/cuda/ms_deform_im2col_cuda.cuh:937-941) launch errors are returned.
 */
#ifndef RELATION_DETR_AMD_H
#define RELATION_DETR_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RDETR_ABI_VERSION 3

typedef enum rdetr_status {
    RDETR_OK = 0,
    RDETR_ERR_INVALID_ARG = -1,   /* null pointer, negative size, misaligned pointer */
    RDETR_ERR_UNSUPPORTED = -2,   /* shape outside what the kernels are built for */
    RDETR_ERR_LAUNCH = -3         /* hipGetLastError() != hipSuccess after the launch */
} rdetr_status;

int rdetr_abi_version(void);
const char *rdetr_status_string(int status);

/* ---------------------------------------------------------------------------------------------
 * Multi-scale deformable attention, forward.
 * Replaces  _C.ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc,
 *           attn_weight, im2col_step)            models/bricks/ops/cuda/ms_deform_attn_cuda.cu:12-72,148-150
 *           (kernel ms_deformable_im2col_gpu_kernel, ms_deform_im2col_cuda.cuh:226-288) and equals
 *           multi_scale_deformable_attn_pytorch  models/bricks/ms_deform_attn.py:159-212.
 *
 *   value          [B, S, H, D]        levels packed along S, row-major (y*W_l + x) inside a level
 *   spatial_shapes [L, 2] int64 (h,w)  level_start_index [L] int64          (device memory)
 *   sampling_loc   [B, Nq, H, L, P, 2] fp32, (x, y) normalised to [0,1]
 *   attn_weight    [B, Nq, H, L, P]    fp32 (already soft-maxed over L*P by the caller)
 *   out            [B, Nq, H*D]        channel = head*D + c
 *
 * Fast path (H = 8, D = 32, P = 4, L <= 8, see rdetr_msda_fast_path()): the query-run kernel of
 * csrc/msda_fwd.hip; any other (H, D, P) runs a generic one-thread-per-output kernel.  There is no im2col_step batch
 * restriction (the reference requires B % min(B, 64) == 0, ms_deform_attn_cuda.cu:42-44).
 * NaN sampling locations contribute zero (the CUDA op's guard, ms_deform_im2col_cuda.cuh:277).
 * The bf16 variant stores value and out as bfloat16 and keeps locations, weights and the
 * accumulation in fp32 (an extension: the reference op is fp32/fp64 only, ms_deform_attn_cuda.cu:56).
 */
int rdetr_msda_forward_f32(const float *value, const int64_t *spatial_shapes, const int64_t *level_start_index,
                           const float *sampling_loc, const float *attn_weight, int B, int S, int H, int D,
                           int L, int Nq, int P, float *out, void *stream);

int rdetr_msda_forward_bf16(const uint16_t *value, const int64_t *spatial_shapes, const int64_t *level_start_index,
                            const float *sampling_loc, const float *attn_weight, int B, int S, int H, int D,
                            int L, int Nq, int P, uint16_t *out, void *stream);

/* Same operator with the sampling-location producer fused in.
 * Replaces lines 322-349 + the core call of MultiScaleDeformableAttention.forward
 * (models/bricks/ms_deform_attn.py): the caller passes the RAW outputs of the two query projections,
 *   sampling_offsets [B, Nq, H, L, P, 2]   attn_logits [B, Nq, H, L*P]   (dtype of value: fp32 / bf16)
 *   reference_points [B, Nq, L, ref_dim] fp32, ref_dim 2 (x,y) or 4 (cx,cy,w,h)
 * and the kernel performs softmax over L*P and  loc = ref + off/(W_l,H_l)  (ref_dim 2)  or
 * ref_xy + off/P * ref_wh * 0.5  (ref_dim 4)  in its set-up phase -- locations and weights never
 * exist in HBM.  Fast-path shapes only (rdetr_msda_fast_path() == 1); otherwise RDETR_ERR_UNSUPPORTED
 * and the caller uses rdetr_msda_forward_* with materialised locations / weights. */
int rdetr_msda_forward_fused_f32(const float *value, const int64_t *spatial_shapes, const int64_t *level_start_index,
                                 const float *sampling_offsets, const float *attn_logits,
                                 const float *reference_points, int ref_dim, int B, int S, int H, int D, int L, int Nq,
                                 int P, float *out, void *stream);

int rdetr_msda_forward_fused_bf16(const uint16_t *value, const int64_t *spatial_shapes,
                                  const int64_t *level_start_index, const uint16_t *sampling_offsets,
                                  const uint16_t *attn_logits, const float *reference_points, int ref_dim, int B, int S,
                                  int H, int D, int L, int Nq, int P, uint16_t *out, void *stream);

/* General fused-producer form.  `key_padding_mask` (u8 [B, S], non-zero = padded position, may be NULL) is applied inside the
 * gather: a padded pixel's row counts as zero, which is what zero-filling the projected value does
 * (models/bricks/ms_deform_attn.py:316-319) -- without a pass over the [B, S, H*D] tensor.  ROW STRIDES (in elements;
 * 0 = contiguous) of the two projection outputs, so that sampling_offsets and attention_weights can be the column slices [0, 2*H*L*P) and
 * [2*H*L*P, 3*H*L*P) of ONE [rows, 3*H*L*P] GEMM output (one projection GEMM instead of two).  ld_offsets must be even
 * (offsets are read as (x, y) pairs). */
int rdetr_msda_forward_fused_ex_f32(const float *value, const int64_t *spatial_shapes, const int64_t *level_start_index,
                                    const float *sampling_offsets, int ld_offsets, const float *attn_logits, int ld_logits,
                                    const float *reference_points, int ref_dim, const uint8_t *key_padding_mask, int B,
                                    int S, int H, int D, int L, int Nq, int P, float *out, void *stream);
int rdetr_msda_forward_fused_ex_bf16(const uint16_t *value, const int64_t *spatial_shapes,
                                     const int64_t *level_start_index, const uint16_t *sampling_offsets, int ld_offsets,
                                     const uint16_t *attn_logits, int ld_logits, const float *reference_points, int ref_dim,
                                     const uint8_t *key_padding_mask, int B, int S, int H, int D, int L, int Nq, int P,
                                     uint16_t *out, void *stream);

/* bf16 operator (both producer forms) with the VALUE LAYOUT and the KERNEL CHOICE as explicit arguments -- the library
 * reads no environment variable and keeps no state.
 *   value_layout  RDETR_VALUE_BSHD  value [B, S, H, D]: the reference operator's layout (ms_deform_attn_cuda.cu:12-19);
 *                 RDETR_VALUE_BHSD  value [B, H, S, D]: head-major -- one (image, head) plane is contiguous, so the window
 *                                   kernel fills its LDS windows at the contiguous-row rate.  Written by
 *                                   rdetr_value_to_head_major_bf16 (below); fast-path shapes only.
 *   algo          RDETR_MSDA_AUTO    what the plain entry points do: the direct kernel.  Correct for ANY level table the
 *                                    reference operator accepts (gaps, overlapping or unordered levels, sum(h*w) < S);
 *                 RDETR_MSDA_DIRECT  csrc/msda_fwd.hip -- the query-run kernel (range-checked global gathers), any Nq, L <= 8;
 *                 RDETR_MSDA_AUTO_PACKED  the faster kernel per layout: for RDETR_VALUE_BSHD the window kernel where it
 *                                    applies, else (and for RDETR_VALUE_BHSD) the direct kernel.  Same PRECONDITION as
 *                                    RDETR_MSDA_WINDOW;
 *                 RDETR_MSDA_WINDOW  csrc/msda_win.hip -- LDS-window MFMA kernel for the ENCODER shape: queries are the
 *                                    pyramid's own pixels in level_start order (Nq == S), L == 4, S >= 4096, no padding
 *                                    mask.  PRECONDITION (the caller's promise -- the level table lives in device memory
 *                                    and the library never synchronises, so it cannot check): the levels TILE [0, S)
 *                                    exactly, level_start[l] = sum of h*w of the earlier levels and sum(h*w) == S, and no
 *                                    level is larger than level 0 (the kernel enumerates its queries as the pixels of the
 *                                    levels: with gaps or overlaps output rows would be left unwritten or written twice).
 *                                    rdetr_msda_levels_window_ok() checks a HOST copy of the table.  Per 16 x 12 query tile and level a 32-pixel-wide window of the value plane is
 *                                    copied L2 -> LDS by range-checked LDS-DMA (pixels outside the level arrive as zeros) and
 *                                    gathered from there on the matrix cores; samples outside their window are fetched from
 *                                    global memory, so results never depend on the windows.  RDETR_ERR_UNSUPPORTED for any
 *                                    other shape.
 * Results of the two kernels agree to the rounding of the bf16 output (different summation order). */
#define RDETR_VALUE_BSHD 0
#define RDETR_VALUE_BHSD 1
#define RDETR_MSDA_AUTO 0
#define RDETR_MSDA_DIRECT 1
#define RDETR_MSDA_WINDOW 2
#define RDETR_MSDA_AUTO_PACKED 3
/* 1 if a level table (HOST pointers) meets the precondition of RDETR_MSDA_WINDOW / RDETR_MSDA_AUTO_PACKED for a value tensor
 * with S positions, else 0.  Pure host arithmetic. */
int rdetr_msda_levels_window_ok(const int64_t *host_spatial_shapes, const int64_t *host_level_start_index, int L, long long S);
int rdetr_msda_forward_opt_bf16(const uint16_t *value, int value_layout, const int64_t *spatial_shapes,
                                const int64_t *level_start_index, const float *sampling_loc, const float *attn_weight, int B,
                                int S, int H, int D, int L, int Nq, int P, int algo, uint16_t *out, void *stream);
/* The bf16 operator on the SWEEP kernel (csrc/msda_sweep.hip, round 4) -- the LDS-sourced gather for the ENCODER shape: queries
 * are the pyramid's own pixels in level_start order (Nq == S), L == 4, H == 8, D == 32, P == 4.  Replaces the kernel of
 * ms_deformable_im2col_cuda (ms_deform_im2col_cuda.cuh:226-288, 912-943) for that shape; same tensors as
 * ms_deform_attn_cuda_forward (ms_deform_attn_cuda.cu:12-72) except that the LEVEL TABLE is passed as HOST pointers: the grid
 * and the band height are sized from it and the kernel takes it as arguments (the reference reads spatial_shapes on the host
 * too, ms_deform_attn.py:313).  PRECONDITION as RDETR_MSDA_WINDOW (the levels tile [0, S), checked here on the host copy;
 * the caller promises that the device tensors it built sampling_loc from describe the same levels).  One persistent
 * 1024-thread workgroup per CU slides ring-buffer windows of all four levels along a band of level-0 rows, 8 columns per
 * step: only the new columns are fetched (range-checked LDS-DMA, a step ahead: outside the level = zeros), the corner rows
 * are gathered from LDS by ds_read_b64_tr_b16 into v_mfma_f32_16x16x32_bf16, samples outside their window are fetched from
 * global memory and added in fp32, so results never depend on the windows.  RDETR_ERR_UNSUPPORTED for any other shape
 * (pyramids whose coarser levels are not about half the finer ones included; callers use the direct kernel);
 * sampling_loc must be 16-byte and attn_weight 8-byte aligned.  OPT-IN: measured slower than the direct kernel (DESIGN.md
 * section 4.2 has the numbers and the measured ceiling of the data path). */
int rdetr_msda_forward_sweep_bf16(const uint16_t *value, int value_layout, const int64_t *host_spatial_shapes,
                                  const int64_t *host_level_start_index, const float *sampling_loc, const float *attn_weight,
                                  int B, int S, int H, int D, int L, int Nq, int P, uint16_t *out, void *stream);
/* The bf16 operator on the RESIDENT-LEVELS kernel (csrc/msda_res.hip, round 4), head-major value [B,H,S,D] only.  Same operator
 * (ms_deform_attn_cuda_forward, ms_deform_attn_cuda.cu:12-72; kernel ms_deform_im2col_cuda.cuh:226-288), same per-query
 * arithmetic as the query-run kernel; the points of a query are accumulated in another order (coarse and fine levels
 * alternate), so results agree with rdetr_msda_forward_opt_bf16(..., RDETR_MSDA_DIRECT) to fp32 re-association: the last bit
 * of a bf16 output differs now and then (4e-5 of the outputs at the R50 shape).  One persistent 1024-thread workgroup per CU serves one (image, head) plane and keeps the
 * plane's COARSE levels -- as many trailing levels as fit beside the staging area in the CU's 160 KB of LDS (levels 2 and 3
 * at the R50 800 x 1333 shape: half of all samples) -- resident in LDS: samples on those levels read their corner rows with
 * ds_read_b128 instead of through the texture path, whose instruction rate bounds the query-run kernel.
 * The LEVEL TABLE is passed as HOST pointers (the resident set is sized on the host; the reference reads spatial_shapes on
 * the host too, ms_deform_attn.py:313); the levels must tile [0, S).  RDETR_ERR_UNSUPPORTED -- callers then use the
 * query-run kernel -- for: L other than 4 or 5, a level table that does not tile [0, S), not even the coarsest level fitting,
 * inputs that miss the vector-load alignment (sampling_loc / attn_weight 16 bytes at L = 4). */
int rdetr_msda_forward_resident_bf16(const uint16_t *value_bhsd, const int64_t *host_spatial_shapes,
                                     const int64_t *host_level_start_index, const float *sampling_loc, const float *attn_weight,
                                     int B, int S, int H, int D, int L, int Nq, int P, uint16_t *out, void *stream);
/* ... and its fused-producer form (arguments as rdetr_msda_forward_fused_opt_bf16; no key_padding_mask: the head-major
 * value's padded rows are zero already; 2-d reference points only -- ref_dim == 4 is RDETR_ERR_UNSUPPORTED here and runs on the
 * query-run kernel). */
int rdetr_msda_forward_fused_resident_bf16(const uint16_t *value_bhsd, const int64_t *host_spatial_shapes,
                                           const int64_t *host_level_start_index, const uint16_t *sampling_offsets,
                                           int ld_offsets, const uint16_t *attn_logits, int ld_logits,
                                           const float *reference_points, int ref_dim, int B, int S, int H, int D, int L, int Nq,
                                           int P, uint16_t *out, void *stream);
int rdetr_msda_forward_fused_opt_bf16(const uint16_t *value, int value_layout, const int64_t *spatial_shapes,
                                      const int64_t *level_start_index, const uint16_t *sampling_offsets, int ld_offsets,
                                      const uint16_t *attn_logits, int ld_logits, const float *reference_points, int ref_dim,
                                      const uint8_t *key_padding_mask, int B, int S, int H, int D, int L, int Nq, int P,
                                      int algo, uint16_t *out, void *stream);

/* The fused-producer bf16 operator on a ROW-STRIDED value: pixel s of image b starts at value + (b * S + s) * value_ld elements
 * (value_ld >= H * D, a 16-byte multiple; 0 = dense), i.e. `value` may be a 256-column slice of a wider projection output.  Lets
 * the SIX cross-attention value projections of the decoder (models/bricks/relation_transformer.py:464-471 calls value_proj of
 * each layer on the same encoder memory, ms_deform_attn.py:316) run as ONE [S, 256] x [256, 6*256] GEMM whose output each
 * layer's gather reads in place.  [B,S,H,D] layout, direct kernel, fast-path shapes; other arguments as
 * rdetr_msda_forward_fused_ex_bf16. */
int rdetr_msda_forward_fused_strided_bf16(const uint16_t *value, long long value_ld, const int64_t *spatial_shapes,
                                          const int64_t *level_start_index, const uint16_t *sampling_offsets, int ld_offsets,
                                          const uint16_t *attn_logits, int ld_logits, const float *reference_points, int ref_dim,
                                          const uint8_t *key_padding_mask, int B, int S, int H, int D, int L, int Nq, int P,
                                          uint16_t *out, void *stream);

/* Projected value [B, S, H*D] bf16 (rows `ld` elements apart, ld % 8 == 0: the rows may be a column slice of a wider
 * buffer) -> head-major [B, H, S, D], with the rows of padded positions (`key_padding_mask` u8 [B, S], may be NULL)
 * written as zeros -- the zero-fill of models/bricks/ms_deform_attn.py:316-319 folded into the re-layout.  H = 8, D = 32. */
int rdetr_value_to_head_major_bf16(const uint16_t *src, long long ld, const uint8_t *key_padding_mask, int B, int S, int H,
                                   int D, uint16_t *dst, void *stream);

/* MSDA's value projection (models/bricks/ms_deform_attn.py:316-321: value_proj, then zero-fill of the padded rows) written
 * straight into the head-major layout by the hand-written MFMA projection kernel (csrc/linear.hip):
 *   out_hm [B, 8, S, 32] bf16  <-  x [B*S, 256] (rows ldx elements apart) w[256, 256]^T + bias[256] (nullable),
 * rows with row_mask != 0 (u8 [B*S], nullable) stored as zeros.  Same bits as rdetr_linear_k256_bf16 followed by
 * rdetr_value_to_head_major_bf16. */
int rdetr_linear_k256_hm_bf16(const uint16_t *x, long long ldx, const uint16_t *w, const uint16_t *bias, const uint8_t *row_mask,
                              int B, int S, uint16_t *out_hm, void *stream);

/* 1 if (H, D, L, P) is served by the query-run kernel, 0 if by the generic kernel. */
int rdetr_msda_fast_path(int H, int D, int L, int P);

/* ---------------------------------------------------------------------------------------------
 * Multi-scale deformable attention, backward.
 * Replaces  _C.ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc,
 *           attn_weight, grad_output, im2col_step) -> [grad_value, grad_sampling_loc, grad_attn_weight]
 *           models/bricks/ops/cuda/ms_deform_attn_cuda.cu:75-145, ms_deform_im2col_cuda.cuh:290-392.
 * grad_value [B,S,H,D] is accumulated with float atomics and MUST be zero-filled by the caller
 * (the reference zero-fills inside the op, ms_deform_attn_cuda.cu:113-115; the Python wrapper does
 * it here).  grad_sampling_loc / grad_attn_weight are fully overwritten.
 */
int rdetr_msda_backward_f32(const float *value, const int64_t *spatial_shapes, const int64_t *level_start_index,
                            const float *sampling_loc, const float *attn_weight, const float *grad_out, int B,
                            int S, int H, int D, int L, int Nq, int P, float *grad_value,
                            float *grad_sampling_loc, float *grad_attn_weight, void *stream);

/* The same gradients with a DETERMINISTIC grad_value ("deterministic alternative to atomics"; the reference's
 * backward adds with atomicAdd, ms_deform_im2col_cuda.cuh:290-392, so its bits depend on the run): one (pixel-row key, weight)
 * record per sample corner, a stable radix sort of the records by key, and a per-row sum in sorted = original sample order.
 * grad_sampling_loc / grad_attn_weight are fixed-order sums in both modes.  grad_value is OVERWRITTEN (every row, no
 * zero-initialisation needed).  workspace: rdetr_msda_backward_det_workspace_bytes() bytes, 16-byte aligned (records + sort
 * buffers + the sort's temporary storage; 20 bytes per sample corner: 0.9 GB at B = 4 of the R50 encoder shape); that function
 * returns 0 for empty problems / unsupported shapes and -1 when the record count exceeds 2^31.  H = 8, D = 32, P = 4, L <= 8 only
 * (RDETR_ERR_UNSUPPORTED otherwise).  Measured at the R50 encoder shape: 1.87 ms vs 1.07 ms (B = 1), 5.84 vs 4.24 ms (B = 4) for the
 * atomic kernel (tools/profile_msda_bwd.py). */
long long rdetr_msda_backward_det_workspace_bytes(int B, int S, int H, int D, int L, int Nq, int P);
int rdetr_msda_backward_det_f32(const float *value, const int64_t *spatial_shapes, const int64_t *level_start_index,
                                const float *sampling_loc, const float *attn_weight, const float *grad_out, int B, int S, int H, int D,
                                int L, int Nq, int P, void *workspace, long long workspace_bytes, float *grad_value,
                                float *grad_sampling_loc, float *grad_attn_weight, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Position-relation bias.
 * Replaces  PositionRelationEmbedding.forward  models/bricks/relation_transformer.py:520-532
 *           = box_rel_encoding (:481-490) -> get_sine_pos_embed (models/bricks/position_encoding.py:115-138)
 *             -> Conv2d(4*F -> Hh, 1x1) + ReLU -> clone.
 *   src [B, N1, 4], tgt [B, N2, 4]   cxcywh boxes, fp32
 *   proj_weight [Hh, 4*F]  (pos_proj.0.weight flattened), proj_bias [Hh] (may be NULL)
 *   out [B, Hh, N1, N2]    fp32, >= 0;  rows = src (queries), cols = tgt (keys)
 * F = num_pos_feats (16 in every relation_detr config), must be even and <= 32; Hh <= 16.
 */
int rdetr_relation_bias_f32(const float *src, const float *tgt, const float *proj_weight, const float *proj_bias,
                            int B, int N1, int N2, int Hh, int F, float scale, float temperature, float eps,
                            float *out, void *stream);

/* Same operator with a caller-provided workspace of (B*N1 + B*N2) * 2*F floats (8-byte aligned): for F = 16, Hh = 8 the
 * two size-ratio coordinates log(w_i/w_j), log(h_i/h_j) take their sine features from per-box tables (angle computed in
 * double precision once per box) by the angle-difference identities instead of evaluating sin / cos per pair -- a third
 * fewer instructions.  Results stay within 1e-4 of both the fp32 and the fp64 evaluation of the reference formula (the
 * fp32 reference itself rounds these angles to ~3e-5 rad).  Any other configuration forwards to rdetr_relation_bias_f32. */
int rdetr_relation_bias_ws_f32(const float *src, const float *tgt, const float *proj_weight, const float *proj_bias,
                               int B, int N1, int N2, int Hh, int F, float scale, float temperature, float eps,
                               float *workspace, float *out, void *stream);

/* Backward of the relation bias with respect to the 1x1 projection -- what autograd computes for
 * PositionRelationEmbedding.pos_proj in training (models/bricks/relation_transformer.py:527-532; the boxes carry no gradient,
 * :527-529):
 *     grad_weight[h][ch] = sum_{b,i,j} g[b,h,i,j] * feat[b,i,j,ch],   grad_bias[h] = sum g,   g = grad_out where active else 0
 * with the 64 sine features REGENERATED from the boxes by the forward's arithmetic (the reference keeps [N1, N2, 64] per image
 * for its backward GEMM: 207 MB at N = 900).
 *   src [B,N1,4], tgt [B,N2,4] fp32 cxcywh (tgt 16-byte aligned); grad_out [B,Hh,N1,N2] fp32; active u8 [B,Hh,N1,N2] = (forward
 *   output > 0), the ReLU's derivative; workspace: rdetr_relation_bias_backward_workspace_bytes(B, N1, N2) bytes;
 *   grad_weight [Hh, 4F] fp32, grad_bias [Hh] fp32 or NULL -- both OVERWRITTEN (no zero-initialisation needed).
 * DETERMINISTIC: per-block partial sums go to the workspace and a second kernel adds them in a fixed order (no float atomics).
 * Hh = 8, F = 16 only (RDETR_ERR_UNSUPPORTED otherwise). */
long long rdetr_relation_bias_backward_workspace_bytes(int B, int N1, int N2);
int rdetr_relation_bias_backward_f32(const float *src, const float *tgt, const float *grad_out, const uint8_t *active, int B, int N1,
                                     int N2, int Hh, int F, float scale, float temperature, float eps, float *workspace,
                                     float *grad_weight, float *grad_bias, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Bias-add + row softmax of decoder self-attention scores, in place.
 * Replaces the softmax(QK^T/sqrt(d) + attn_mask) step of nn.MultiheadAttention as called at
 * models/bricks/relation_transformer.py:452-459 (float mask [B*Hh, N, N] from :372-374).
 *   scores [BH, N1, N2] fp32, overwritten with the probabilities
 *   bias   [BH, N1, N2] fp32 or NULL (may contain -inf)
 *   mask   [N1, N2] uint8 or NULL; non-zero = excluded (the bool attn_mask convention)
 * A fully masked row yields NaN, as torch.softmax does.
 */
int rdetr_bias_softmax_f32(float *scores, const float *bias, const uint8_t *mask, int BH, int N1, int N2,
                           void *stream);

/* ---------------------------------------------------------------------------------------------
 * Decoder self-attention with the relation bias, fused:
 *   out = softmax(Q K^T * scale + bias [, bool mask]) V     per (image, head), bf16 activations, fp32 soft-max.
 * Replaces the attention core of the nn.MultiheadAttention call of the decoder layer
 *           models/bricks/relation_transformer.py:452-461 with the float bias of :369-374 as attn_mask
 * (QK^T GEMM, bias add, softmax, PV GEMM and the dtype copies between them) by one flash-style MFMA kernel.
 *   q [B, N, ..], k / v [B, M, ..]   bf16, head h at columns h*D .. h*D+D-1 of a row; ldq / ldk / ldv = row stride in
 *                                    ELEMENTS (so slices of a packed in-projection output can be passed as they are),
 *                                    image stride = rows * ld
 *   bias  fp32 [B*H, N, M] or NULL (may hold -inf);   bool_mask u8 [N, M] or NULL (non-zero = excluded)
 *   out   bf16 [B, N, ..] with row stride ldo;  D = 32 only (RDETR_ERR_UNSUPPORTED otherwise)
 * A fully masked row yields NaN, as torch.softmax does. */
int rdetr_relation_attention_bf16(const uint16_t *q, const uint16_t *k, const uint16_t *v, int ldq, int ldk, int ldv,
                                  const float *bias, const uint8_t *bool_mask, int B, int H, int D, int N, int M,
                                  float scale, uint16_t *out, int ldo, void *stream);

/* ---------------------------------------------------------------------------------------------
 * The same attention with the relation bias GENERATED INSIDE the kernel from the boxes (SURVEY.md section 8 f1 as written):
 *   out = softmax(Q K^T * attn_scale + relu(W . sine(box_rel_encoding(src_boxes, tgt_boxes)) + b) [, bool mask]) V
 * Replaces PositionRelationEmbedding.forward (models/bricks/relation_transformer.py:520-532, box_rel_encoding :481-490,
 * get_sine_pos_embed position_encoding.py:115-138) TOGETHER WITH the nn.MultiheadAttention call that consumes its result
 * (:369-374, :452-461): the [B, H, N, M] fp32 bias never exists in HBM.
 *   q, k, v, ld*, bool_mask, out, ldo   as rdetr_relation_attention_bf16
 *   src_boxes [B, N, 4], tgt_boxes [B, M, 4]   fp32 cxcywh (query i <-> src box i, key j <-> tgt box j), 16-byte aligned
 *   proj_weight [H, 4F] fp32 (pos_proj.0.weight), proj_bias [H] fp32 or NULL
 *   H = 8, D = 32, F = 16 only (RDETR_ERR_UNSUPPORTED otherwise -> materialise the bias with rdetr_relation_bias_f32)
 * bf16 inference path: the sine features are rounded to bf16 for the MFMA projection and the angles use the hardware
 * log2 / sin / cos; results are held to the same bound against the fp32 reference as rdetr_relation_attention_bf16. */
int rdetr_relation_attention_boxes_bf16(const uint16_t *q, const uint16_t *k, const uint16_t *v, int ldq, int ldk, int ldv,
                                        const float *src_boxes, const float *tgt_boxes, const float *proj_weight,
                                        const float *proj_bias, const uint8_t *bool_mask, int B, int H, int D,
                                        int N, int M, int F, float rel_scale, float temperature, float eps, float attn_scale,
                                        uint16_t *out, int ldo, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Residual add + LayerNorm over the last dimension, one pass (callers either side of the hot path).
 * Replaces the pairs  x + sublayer(x) -> nn.LayerNorm  of the encoder / decoder layers
 *           models/bricks/relation_transformer.py:262-276 (encoder layer), :452-478 (decoder layer), :360.
 *   x, residual (nullable), out  [rows, C]      gamma, beta [C]      all in one dtype (fp32 / bf16)
 *   out = (x + residual - mean) / sqrt(var + eps) * gamma + beta, statistics in fp32, biased variance
 *   (torch.nn.functional.layer_norm); the sum is not rounded to the storage type before normalising.
 * C = 256 with 16-byte-aligned pointers takes the vectorised kernel; any other C <= 8192 a strided one. */
int rdetr_add_layernorm_f32(const float *x, const float *residual, const float *gamma, const float *beta,
                            long long rows, int C, float eps, float *out, void *stream);
int rdetr_add_layernorm_bf16(const uint16_t *x, const uint16_t *residual, const uint16_t *gamma, const uint16_t *beta,
                             long long rows, int C, float eps, uint16_t *out, void *stream);
/* Same with row strides in ELEMENTS (ldx, ldr, ldo >= C): inputs / output may be column slices of wider matrices -- the
 * encoder writes each layer's output straight into its slice of the [rows, 7*C] memory-fusion input
 * (relation_transformer.py:196-214) instead of concatenating afterwards. */
int rdetr_add_layernorm_strided_f32(const float *x, const float *residual, const float *gamma, const float *beta,
                                    long long rows, int C, long long ldx, long long ldr, long long ldo, float eps,
                                    float *out, void *stream);
int rdetr_add_layernorm_strided_bf16(const uint16_t *x, const uint16_t *residual, const uint16_t *gamma,
                                     const uint16_t *beta, long long rows, int C, long long ldx, long long ldr,
                                     long long ldo, float eps, uint16_t *out, void *stream);

/* As the strided form, with a second output out2 = out + pos (the next encoder layer's `query + query_pos`,
 * models/bricks/relation_transformer.py:262): computed from the stored (rounded) `out`, i.e. the bits of a separate add.
 * pos / out2: rows ldp / ldo2 elements apart, both required. */
int rdetr_add_layernorm_pos_f32(const float *x, const float *residual, const float *gamma, const float *beta,
                                const float *pos, long long rows, int C, long long ldx, long long ldr, long long ldo,
                                long long ldp, long long ldo2, float eps, float *out, float *out2, void *stream);
int rdetr_add_layernorm_pos_bf16(const uint16_t *x, const uint16_t *residual, const uint16_t *gamma,
                                 const uint16_t *beta, const uint16_t *pos, long long rows, int C, long long ldx,
                                 long long ldr, long long ldo, long long ldp, long long ldo2, float eps, uint16_t *out,
                                 uint16_t *out2, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Fused elementwise steps of the decoder's box bookkeeping (each replaces ~8 torch launches on a [B,N,4] tensor).
 *   rdetr_box_refine_f32   out = sigmoid(delta + inverse_sigmoid(ref)),  inverse_sigmoid as util/misc.py:31-35 (eps 1e-3);
 *                          the iterative box refinement of models/bricks/relation_transformer.py:363-381.
 *                          delta: n values, fp32 (delta_is_bf16 = 0) or bf16 (1); ref, out: n fp32 values.
 *   rdetr_sine_pos_embed   get_sine_pos_embed(pos, num_pos_feats = F, temperature, scale, exchange_xy = True)
 *                          (models/bricks/position_encoding.py:101-138): pos fp32 [rows, n] -> out [rows, n * F] fp32 or bf16,
 *                          coordinate order (y, x, rest), channel = coord * F + 2k + {sin, cos}, F even, <= 128. */
int rdetr_box_refine_f32(const void *delta, int delta_is_bf16, const float *ref, long long n, float eps, float *out,
                         void *stream);
int rdetr_sine_pos_embed(const float *pos, long long rows, int n, int F, float temperature, float scale, void *out,
                         int out_is_bf16, void *stream);

/* ---------------------------------------------------------------------------------------------
 * One-pass steps of the encoder's input / output side (each replaces a torch chain that made 2-3 passes over a [B,S,C] tensor).
 *   rdetr_zero_masked_rows  x.masked_fill_(mask[..., None], 0) in place: the padding fill of the projected value
 *                           (models/bricks/ms_deform_attn.py:316-319).  x: rows of row_bytes bytes, ld_bytes apart (both
 *                           multiples of 16, x 16-byte aligned), mask: one byte per row (non-zero = padded).  Only the padded
 *                           rows are written.
 *   rdetr_row_max           out[r] = max(x[r, 0..C)) (NaN propagates), x fp32 or bf16 [rows, C] with row stride ldx, out
 *                           [rows] in x's type: the class-score maximum of the two-stage query selection
 *                           (models/bricks/relation_transformer.py:105).
 *   rdetr_nchw_to_tokens    one pyramid level src [B, C, P] (P = H*W) -> out[b, p, c] = src[b, c, p] (+ add_vec[c]), i.e.
 *                           x.flatten(2).transpose(1, 2) (+ the level embedding, relation_transformer.py:87-89) written
 *                           straight into the level's rows of the level-packed token tensor (base_transformer.py:17-23):
 *                           `out` points at the level's first row of image 0; rows are ld_out elements apart (>= C), images
 *                           out_image_stride elements.  add_vec nullable. */
int rdetr_zero_masked_rows(void *x, const unsigned char *mask, long long rows, int row_bytes, long long ld_bytes,
                           void *stream);
int rdetr_row_max(const void *x, int is_bf16, long long rows, int C, long long ldx, void *out, void *stream);

/* Row-wise top-k (largest, sorted) for the two selections on the transformer's dependency chain:
 *   torch.topk(enc_outputs_class.max(-1)[0], 900, dim=1)    models/bricks/relation_transformer.py:93 (and :105 for the hybrid branch)
 *   torch.topk(prob.view(B, -1), 300, dim=1)                models/bricks/post_process.py:30
 *   x [rows, n] fp32 or bf16, contiguous  ->  values [rows, k] fp32, indices [rows, k] int64
 * Order: value descending, equal values by index ascending, NaN above everything (torch.topk leaves the order of equal values
 * unspecified).  1 <= k <= min(n, 1024), n < 2^20.  workspace: rdetr_topk_workspace_bytes(rows, n, k) bytes of scratch, 16-byte
 * aligned. */
long long rdetr_topk_workspace_bytes(int rows, int n, int k);
int rdetr_topk(const void *x, int is_bf16, int rows, int n, int k, void *workspace, float *values, long long *indices, void *stream);

/* The decoder's box head and box refinement in one kernel (bf16 activations, embed_dim 256):
 *   out = sigmoid(W3 relu(W2 relu(W1 x + b1) + b2) + b3 + inverse_sigmoid(reference))
 * Replaces bbox_head[i] = MLP(256, 256, 4, 3) (models/bricks/basic.py:6-24, relation_transformer.py:294) followed by the
 * refinement of relation_transformer.py:363-381 (inverse_sigmoid: util/misc.py:31-35), for the layer's two calls at once:
 *   xa [M, 256] -> out_a [M, 4] (the layer's boxes, from norm(query)),  xb [M, 256] or NULL -> out_b (the next reference points)
 *   pw1, pw2   the hidden weights [256, 256] packed by rdetr_linear_pack_k256_bf16;  b1, b2 [256], w3 [4, 256], b3 [4]   bf16
 *   reference  fp32 [M, 4] (shared by both inputs), eps of inverse_sigmoid;  outputs fp32.  lda / ldb: row strides in elements.
 *   reference_is_logit != 0: the reference is in logit space already and is added as it is -- the two-stage proposal boxes
 *              sigmoid(encoder_bbox_head(output_memory) + output_proposals), relation_transformer.py:88-90
 * Hidden activations and the final delta are rounded to bf16 where the unfused path stores them. */
int rdetr_box_head_k256_bf16(const uint16_t *xa, long long lda, const uint16_t *xb, long long ldb, const uint16_t *pw1,
                             const uint16_t *b1, const uint16_t *pw2, const uint16_t *b2, const uint16_t *w3, const uint16_t *b3,
                             const float *reference, int reference_is_logit, float eps, long long M, float *out_a, float *out_b,
                             void *stream);

/* The decoder layer's query position in one launch (models/bricks/relation_transformer.py:294-296, 343-347, 452-455):
 *     out_pos = ref_point_head(emb)                      MLP(512, 256, 256, 2) on the sine embedding of the reference boxes
 *     out_pos = out_pos * query_scale(query)             MLP(256, 256, 256, 2), layers >= 1 -- when pv1 / c1 / pv2 / c2 are given
 *     out_qpp = query + out_pos                          the q = k input of the layer's self-attention
 * emb [M, 512], query [M, 256] bf16 (row strides lde / ldq elements, multiples of 8, 16-byte aligned bases); pw1a / pw1b = the
 * K halves [:, :256] / [:, 256:] of ref_point_head.layers[0].weight [256, 512], pw2 = ref_point_head.layers[1].weight, pv1 / pv2 =
 * query_scale's two [256, 256] weights -- each packed by rdetr_linear_pack_k256_bf16; b1, b2, c1, c2 bf16 [256];
 * out_pos, out_qpp [M, 256] bf16 contiguous.  Every intermediate is rounded to bf16 where the unfused sequence (four GEMMs, a
 * product, a sum) stores it. */
int rdetr_query_pos_k256_bf16(const uint16_t *emb, long long lde, const uint16_t *query, long long ldq, const uint16_t *pw1a,
                              const uint16_t *pw1b, const uint16_t *b1, const uint16_t *pw2, const uint16_t *b2, const uint16_t *pv1,
                              const uint16_t *c1, const uint16_t *pv2, const uint16_t *c2, long long M, uint16_t *out_pos,
                              uint16_t *out_qpp, void *stream);

/* The three input projections of an ENCODER layer's MultiScaleDeformableAttention in one launch
 * (models/bricks/ms_deform_attn.py:315-327; called from relation_transformer.py:262-269 with value = query, query = query + pos):
 *     out_hm [B, 8, S, 32] = head-major(value_proj(x)), rows of padded positions zero        (:315-321)
 *     out_q  [B*S, q_cols] = [sampling_offsets ; attention_weights](xq), raw outputs          (:322-327; the fused gather reads the
 *                                                                                              two column slices in place)
 * q_cols = 3 * heads * levels * points = 384 (4 feature levels) or 480 (5: the FocalNet configuration); anything else is
 * RDETR_ERR_UNSUPPORTED.
 * x, xq [B*S, 256] bf16 (row strides ldx / ldq elements, multiples of 8, 16-byte aligned bases); pwv = value_proj.weight packed by
 * rdetr_linear_pack_k256_bf16; pwq = [sampling_offsets.weight ; attention_weights.weight ; zero rows] = [512, 256] as two packed
 * [256, 256] blocks, one after the other; bv [256] / bq [q_cols] bf16 or NULL; row_mask = key_padding_mask u8 [B*S] or NULL.
 * Replaces rdetr_linear_k256_hm_bf16 + one N = q_cols GEMM: same products, fp32 accumulation, outputs rounded to bf16 once. */
int rdetr_encoder_proj_k256_bf16(const uint16_t *x, long long ldx, const uint16_t *xq, long long ldq, const uint16_t *pwv,
                                 const uint16_t *bv, const uint16_t *pwq, const uint16_t *bq, const uint8_t *row_mask, int B, int S,
                                 int q_cols, uint16_t *out_hm, uint16_t *out_q, void *stream);

/* PostProcess after its top-k (models/bricks/post_process.py:30-44) in one launch: for rank r of image b
 *   out[b][r] = (x1, y1, x2, y2, score, label) with box = boxes[b][index / C] (cxcywh in [0, 1]) converted to xyxy and scaled by the
 *   image's (w, h), label = index % C.   score fp32 [B, K], index int64 [B, K], boxes fp32 [B, N, 4], image_sizes int64 [B, 2] (h, w). */
int rdetr_detections_from_topk(const float *score, const long long *index, const float *boxes, const long long *image_sizes, int B,
                               int N, int C, int K, float *out, void *stream);

/* query_pos = a * scale and query + query_pos (models/bricks/relation_transformer.py:346-347, 452) in one pass over n contiguous
 * elements (fp32 or bf16); the product is rounded to the storage type before the add, as the two torch kernels it replaces do. */
int rdetr_scaled_pos(const void *a, const void *scale, const void *query, long long n, int is_bf16, void *pos, void *qp, void *stream);

/* Entry of a decoder layer (models/bricks/relation_transformer.py:335-343) in one launch:
 *   ref_in [B, N, L, 4] = reference [B, N, 4] * (vr[b][l].x, vr[b][l].y, vr[b][l].x, vr[b][l].y)       valid_ratios vr [B, L, 2]
 *   emb    [B, N, 4 F]  = get_sine_pos_embed(ref_in[:, :, 0, :], F, temperature, scale, exchange_xy=True)  (fp32 or bf16) */
int rdetr_decoder_reference(const float *reference, const float *valid_ratios, int B, int N, int L, int F, float temperature,
                            float scale, float *ref_in, void *emb, int emb_is_bf16, void *stream);

/* Pyramid geometry of the two-stage transformer in two launches (the torch sequences are ~40 small launches per forward):
 *   valid_ratios [B, L, 2]    unpadded fraction of each level's width / height         models/bricks/base_transformer.py:42-51
 *   reference    [B, S, L, 2] every position's centre, scaled by the valid ratios      base_transformer.py:57-70
 *   logit        [B, S, 4]    inverse sigmoid of the position's proposal box (centre, 0.05 * 2^level), +inf where the proposal
 *                             is not inside (0.01, 0.99) or the position is padded      relation_transformer.py:162-176
 *   keep         [B, S]       1 / 0 (fp32 or bf16): the factor the encoder memory is multiplied with before enc_output
 * level_masks: L HOST pointers to the levels' DEVICE bool masks [B, h_l, w_l]; level_hw: 2 L host ints (h, w); pad_mask
 * [B, S] u8 (the flattened masks) or NULL.  L <= 8. */
int rdetr_pyramid_points(const unsigned char *const *level_masks, const int *level_hw, int L, int B, const unsigned char *pad_mask,
                         int keep_is_bf16, float *valid_ratios, float *reference, float *logit, void *keep, void *stream);
int rdetr_nchw_to_tokens(const void *src, const void *add_vec, int is_bf16, int B, int C, int P,
                         long long out_image_stride, long long ld_out, void *out, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Dense projection with K = 256 (embed_dim) for tall bf16 inputs, hand-written MFMA kernel (csrc/linear.hip):
 *   out[M, N] = act(x[M, 256] w[N, 256]^T + bias[N]),  bf16 storage, fp32 accumulation, one rounding.
 * Replaces the library GEMM behind nn.Linear for MSDA's value_proj / output_proj / merged sampling_offsets + attention_weights
 * projection (models/bricks/ms_deform_attn.py:259-262) and the FFN's linear1 + ReLU (models/bricks/relation_transformer.py:226-233).
 *   x: rows ldx elements apart (>= 256), w: [N, 256] contiguous (nn.Linear.weight), bias: [N] or NULL, out: rows ldo apart;
 *   N a multiple of 32, ldx / ldo multiples of 8, bases 16-byte aligned (else RDETR_ERR_UNSUPPORTED); relu: 0 | 1. */
int rdetr_linear_k256_bf16(const uint16_t *x, long long ldx, const uint16_t *w, const uint16_t *bias, long long M, int N,
                           int relu, uint16_t *out, long long ldo, void *stream);

/* MSDA's output_proj with the encoder layer's residual + LayerNorm in its epilogue (csrc/linear.hip):
 *   out[M, 256] = LayerNorm(residual + (x[M, 256] w[256, 256]^T + bias))     (models/bricks/ms_deform_attn.py:372-376 followed by
 *   norm1(query + attn), relation_transformer.py:262-271); the projection is rounded to bf16 before the residual is added, as the
 *   unfused path stores it; fp32 two-pass statistics.  The weight is re-ordered once per weight update by
 *   rdetr_linear_pack_k256_bf16 (`packed`: 65,536 bf16 elements).  Rows ldx / ldr / ldo elements apart (multiples of 8), bases
 *   16-byte aligned; bias nullable; gamma / beta [256] bf16. */
int rdetr_linear_pack_k256_bf16(const uint16_t *w, uint16_t *packed, void *stream);
int rdetr_linear_ln_k256_bf16(const uint16_t *x, long long ldx, const uint16_t *packed, const uint16_t *bias,
                              const uint16_t *residual, long long ldr, const uint16_t *gamma, const uint16_t *beta, float eps,
                              long long M, uint16_t *out, long long ldo, void *stream);

/* Fused feed-forward block (csrc/ffn.hip): out[M, 256] = relu(x[M, 256] w1[F, 256]^T + b1[F]) w2[256, F]^T + b2[256], i.e.
 * linear2(relu(linear1(x))) of the encoder / decoder layers (models/bricks/relation_transformer.py:226-233, 272-275) without the
 * [M, F] activations ever reaching HBM.  bf16 storage, fp32 accumulation, the hidden activations rounded to bf16 where the unfused
 * path stores them.  The two weight matrices (as nn.Linear keeps them, contiguous) are first re-ordered ONCE per weight update
 * into the order the kernel streams them: rdetr_ffn_k256_pack_bf16 fills `packed` (2 * 256 * F bf16 elements, 16-byte aligned).
 * x / out rows ldx / ldo elements apart (multiples of 8, 16-byte aligned bases); F a multiple of 64, <= 4096 (else
 * RDETR_ERR_UNSUPPORTED). */
int rdetr_ffn_k256_pack_bf16(const uint16_t *w1, const uint16_t *w2, int F, uint16_t *packed, void *stream);
int rdetr_ffn_k256_bf16(const uint16_t *x, long long ldx, const uint16_t *packed, const uint16_t *b1, const uint16_t *b2,
                        long long M, int F, uint16_t *out, long long ldo, void *stream);
/* The same block with the layer's closing residual + LayerNorm in its epilogue: out = LayerNorm(x + ffn(x)) with gamma / beta [256]
 * bf16 and eps (relation_transformer.py:272-276; ffn(x) rounded to bf16 first, as the unfused path stores it; fp32 two-pass
 * statistics).  pos / out2 (both or neither, rows ldp / ldo2 apart): out2 = out + pos, the next layer's query + query_pos. */
int rdetr_ffn_ln_k256_bf16(const uint16_t *x, long long ldx, const uint16_t *packed, const uint16_t *b1, const uint16_t *b2,
                           const uint16_t *gamma, const uint16_t *beta, float eps, const uint16_t *pos, long long ldp,
                           long long M, int F, uint16_t *out, long long ldo, uint16_t *out2, long long ldo2, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* RELATION_DETR_AMD_H */
