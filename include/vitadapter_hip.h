/*
 * vitadapter_hip.h -- C ABI of libvitadapter_hip.so (MI355X / gfx950 only).
 *
 * This is the drop-in boundary of the ViT-Adapter hot path.  Every entry point takes plain
 * device pointers + sizes + a hipStream_t passed as void*; no torch types cross it.  The
 * Python extension module the reference imports (`import MultiScaleDeformableAttention as
 * MSDA`, /root/reference/detection/ops/functions/ms_deform_attn_func.py:11) is a thin ctypes
 * binding over these symbols (vit-adapter_amd/MultiScaleDeformableAttention.py); INTEGRATION.md
 * shows the binding a maintainer of the reference would add.
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless the name ends in _host
 *   - tensors are dense, row-major ("contiguous"), exactly the layouts the reference asserts
 *     (/root/reference/detection/ops/src/cuda/ms_deform_attn_cuda.cu:28-38)
 *   - work is enqueued on `stream` and the call returns immediately (no host sync, no
 *     allocation, graph-capturable)
 *   - return value: 0 = success; <0 = argument error (VAH_E_*); >0 = hipError_t from the
 *     launch.  vah_last_error() gives a thread-local human readable message.  Unlike the
 *     reference (which only printf()s launch failures, ms_deform_im2col_cuda.cuh:948-952,
 *     1321-1325) every failure is reported to the caller.
 */
#ifndef VITADAPTER_HIP_H
#define VITADAPTER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VAH_OK 0
#define VAH_E_NULL (-1)       /* a required pointer is NULL */
#define VAH_E_SHAPE (-2)      /* a dimension is negative / inconsistent / too large */
#define VAH_E_UNSUPPORTED (-3)
#define VAH_E_ALIGN (-4)      /* a pointer misses the alignment its fast path needs */

/* ABI version; bumped on any signature change. */
int vah_abi_version(void);
/* Thread-local message for the last non-zero return on this thread ("" if none). */
const char *vah_last_error(void);

/* ------------------------------------------------------------------------------------
 * Multi-scale deformable attention  (SURVEY.md section 8 rows a-1 .. a-4)
 *
 * Replaces the reference's
 *   ms_deform_attn_forward  (/root/reference/detection/ops/src/vision.cpp:14,
 *                            ms_deform_attn.h:21-40, cuda/ms_deform_attn_cuda.cu:20-80,
 *                            kernel cuda/ms_deform_im2col_cuda.cuh:237-299)
 *   ms_deform_attn_backward (/root/reference/detection/ops/src/vision.cpp:15,
 *                            ms_deform_attn.h:42-61, cuda/ms_deform_attn_cuda.cu:83-153,
 *                            kernels cuda/ms_deform_im2col_cuda.cuh:301-920)
 *
 *   value    (N, S, M, D)            S = sum_l H_l*W_l
 *   shapes   (L, 2) int64  (H_l, W_l)      -- read on the device, never copied to the host
 *   lsi      (L,)   int64  level start row -- read on the device
 *   loc      (N, Lq, M, L, P, 2)     (x, y) normalised to [0,1] over each level
 *   attn     (N, Lq, M, L, P)
 *   out      (N, Lq, M*D)            fully overwritten
 *
 * The reference's im2col_step only chunks the batch into separate launches and does not
 * change results (ms_deform_attn_cuda.cu:50-75); it therefore lives in the Python binding
 * (argument check only) and not in this ABI.
 *
 * Level guard: a level whose (H, W, start) read from device memory is not a valid window of
 * the S rows (H<1, W<1, start<0, start+H*W>S) contributes nothing instead of faulting.
 * ------------------------------------------------------------------------------------ */
int vah_msda_forward_f32(const float *value, const int64_t *shapes, const int64_t *lsi,
                         const float *loc, const float *attn,
                         int64_t N, int64_t S, int64_t M, int64_t D,
                         int64_t L, int64_t Lq, int64_t P,
                         float *out, void *stream);
int vah_msda_forward_f64(const double *value, const int64_t *shapes, const int64_t *lsi,
                         const double *loc, const double *attn,
                         int64_t N, int64_t S, int64_t M, int64_t D,
                         int64_t L, int64_t Lq, int64_t P,
                         double *out, void *stream);

/* grad_out (N, Lq, M*D).  grad_value (N,S,M,D) MUST be zero on entry (the caller allocates
 * it zeroed exactly as the reference does, ms_deform_attn_cuda.cu:121-123) and is
 * accumulated with float atomics (run-to-run last-bit nondeterminism, as in the reference).
 * grad_loc (N,Lq,M,L,P,2) and grad_attn (N,Lq,M,L,P) are fully overwritten. */
int vah_msda_backward_f32(const float *value, const int64_t *shapes, const int64_t *lsi,
                          const float *loc, const float *attn, const float *grad_out,
                          int64_t N, int64_t S, int64_t M, int64_t D,
                          int64_t L, int64_t Lq, int64_t P,
                          float *grad_value, float *grad_loc, float *grad_attn, void *stream);
int vah_msda_backward_f64(const double *value, const int64_t *shapes, const int64_t *lsi,
                          const double *loc, const double *attn, const double *grad_out,
                          int64_t N, int64_t S, int64_t M, int64_t D,
                          int64_t L, int64_t Lq, int64_t P,
                          double *grad_value, double *grad_loc, double *grad_attn, void *stream);

/* ------------------------------------------------------------------------------------
 * Fused MSDeformAttn core: softmax over the L*P attention logits +
 * sampling-location arithmetic + the gather in one kernel, and their gradients in one more.
 * Replaces lines 108-128 of /root/reference/detection/ops/modules/ms_deform_attn.py.
 *
 *   value     (N,S,M,32)        value_dtype: 0 = fp32, 1 = bf16   (out and grad_out use this dtype)
 *   offsets   (N,Lq,M,L,P,2)    raw sampling_offsets Linear output, param_dtype 0 = fp32 / 1 = bf16
 *   logits    (N,Lq,M,L*P)      raw attention_weights Linear output (softmax is done in-kernel)
 *   ref       (Lq, ref_levels, 2) fp32 reference points (x, y) in [0,1], ref_levels = 1 or L,
 *             shared by the batch (the adapter's reference grids)
 *   location  = ref + offsets / (W_l, H_l)
 * Supported: D == 32 and (L, P) in {(1,4), (3,4), (4,4)}  (vah_msda_fused_supported).
 * Backward (vah_msda_fused_backward): grad_value fp32 (N,S,M,32) zero on entry (float atomics, one per
 * sample, corner and channel as the reference); d_offsets / d_logits in param_dtype, fully written.  It is
 * the fallback of vah_msda_fused_backward_tiled below, which is what the modules call.
 * ------------------------------------------------------------------------------------ */
int vah_msda_fused_supported(int64_t D, int64_t L, int64_t P);
/* offsets_stride / logits_stride: elements between the offsets / logits of consecutive (n, q, m) rows; 0 = contiguous
 * tensors (L*P*2 and L*P).  The module keeps both in ONE fp32 matrix - the output of the paired Linear with its rows
 * ordered [head][offsets | logits] - so that a row's 3*L*P numbers share cache lines: strides 3*L*P, logits pointer =
 * offsets pointer + 2*L*P (ops/modules/ms_deform_attn.py). */
int vah_msda_fused_forward(const void *value, int value_dtype, const int64_t *shapes, const int64_t *lsi,
                           const void *offsets, const void *logits, int param_dtype,
                           int64_t offsets_stride, int64_t logits_stride,
                           const float *ref, int64_t ref_levels,
                           int64_t N, int64_t S, int64_t M, int64_t D, int64_t L, int64_t Lq, int64_t P,
                           void *out, void *stream);
int vah_msda_fused_backward(const void *value, int value_dtype, const int64_t *shapes, const int64_t *lsi,
                            const void *offsets, const void *logits, int param_dtype,
                            const float *ref, int64_t ref_levels, const void *grad_out,
                            int64_t N, int64_t S, int64_t M, int64_t D, int64_t L, int64_t Lq, int64_t P,
                            float *grad_value, void *d_offsets, void *d_logits, void *stream);

/* ------------------------------------------------------------------------------------
 * Fused forward over LDS value windows (csrc/msda_fwd_win.hip), single-level calls (L == 1, D == 32, P == 4,
 * reference points shared by the batch: the adapter's extractor).  Same result as vah_msda_fused_forward for ANY
 * offsets; the windows only decide which corner rows are served from LDS.  The queries are grouped by the 8 x 8-pixel
 * tile of the value map their reference point falls in; a group's window is the tile + halo + 1 pixels on every side.
 * The schedule (permutation of the queries, group ranges) is built ON THE DEVICE by a first small kernel from `ref`
 * and the device copies of spatial_shapes / level_start_index: no host copy of the geometry, nothing cached by tensor
 * identity, capturable in a HIP graph with fresh shape tensors.
 *   ws / ws_bytes : device workspace of at least vah_msda_win_ws_bytes(S, Lq) bytes, 16-byte aligned
 *   ws_holds_schedule : 0 = build the schedule into ws first (one single-workgroup kernel, ~20 us for 21 504 queries);
 *                   1 = ws still holds the schedule an earlier call built from the SAME ref, spatial_shapes,
 *                   level_start_index, S and Lq (the adapter makes six extractor calls per forward with one set of
 *                   deform inputs): only the forward kernel runs.  The forward kernel does not write to ws.
 * A workgroup walks (n, group, head) items: it stages the window once, then one lane per query evaluates its four
 * samples against it (replaces ms_deform_im2col_cuda.cuh:237-299 + the module's softmax / location lines).
 * ------------------------------------------------------------------------------------ */
int64_t vah_msda_win_ws_bytes(int64_t S, int64_t Lq);      /* < 0: not supported (Lq > 2^18) */
int vah_msda_fused_forward_win(const void *value, int value_dtype, const int64_t *shapes, const int64_t *lsi,
                               const void *offsets, const void *logits, int param_dtype,
                               int64_t offsets_stride, int64_t logits_stride, const float *ref,
                               int64_t N, int64_t S, int64_t M, int64_t D, int64_t Lq, int64_t P,
                               int64_t halo, void *ws, int64_t ws_bytes, int ws_holds_schedule, void *out, void *stream);

/* ------------------------------------------------------------------------------------
 * TILED BACKWARD: grad_value without atomics and without a zero-fill, for ANY sampling locations
 * (csrc/msda_tile.hip; replaces the scatter of ms_deform_im2col_cuda.cuh:87-159 as called from
 * :301-403).  No argument is a host copy of device data: like the reference kernels (cuh:274-277) these read
 * spatial_shapes / level_start_index on the device; grids and the workspace are sized from (N, S, M, L, Lq)
 * alone, so a call makes no D2H read and can be captured in a HIP graph with fresh shape tensors.
 * Kernels: (0) plan: level table, list counters zeroed; (1) binning: every (n, q, m, level) row into the lists of
 * the 8x4-pixel tiles of the value map its samples touch (fixed list capacity per level; a list that overflows is
 * replaced by a walk over all queries: slow, exact); (2) tile pass: persistent single-wave workgroups walk the
 * (n, head, tile) lists and sum each on the matrix cores (grad_out rows^T [32 ch x 64 entries] x weights
 * [64 entries x 32 px]; bf16 rows: v_mfma_f32_32x32x16_bf16 with the weights as bf16 hi + lo; fp32 rows:
 * v_mfma_f32_32x32x2_f32) and STORE the tile: every grad_value element is written exactly once, whatever it held
 * on entry.  If the levels do not tile [0, S) exactly (gaps, overlaps, a level outside the value rows) grad_value
 * is zero-filled by the plan kernel and the tile pass adds with atomics instead (the reference's zeros + atomics
 * semantics); a level outside the value rows contributes nothing.
 * grad_loc / grad_attn (d_offsets / d_logits): fp32 values - the gather kernels (scatter switched off) in front;
 * bf16 values - the tile pass itself (each sample is owned by the tile of its first in-map corner; corner dot
 * products against the tile's 10x6-pixel value window on the matrix cores) + a softmax-backward pass.
 *   ws / ws_bytes          : device workspace of at least vah_msda_tile_ws_bytes(...) bytes, 16-byte
 *                            aligned, contents arbitrary (plan, list counters, entries, partial tiles, d(out)/d(p))
 *   needs D == 32, P == 4, 1 <= L <= 4 (VAH_E_UNSUPPORTED otherwise: use the functions above)
 * vah_msda_fused_backward_tiled: grad_value_dtype 0 = fp32, 1 = bf16 (bf16 values only); grad_param_dtype: type of
 * d_offsets / d_logits, = param_dtype or bf16 for fp32 offsets / logits (bf16 values only: the module keeps the
 * sampling offsets in fp32 under autocast and hands bf16 gradients to its Linear layers).  The strides are those of
 * vah_msda_fused_forward, for the inputs and for the gradients (0 = contiguous); with strides the gradient kernels of
 * fp32 values (msda_fused.hip) are not available: VAH_E_UNSUPPORTED.
 * ------------------------------------------------------------------------------------ */
int64_t vah_msda_tile_ws_bytes(int64_t N, int64_t S, int64_t M, int64_t L, int64_t Lq, int64_t P);     /* < 0: not supported */
int vah_msda_backward_tiled_f32(const float *value, const int64_t *shapes, const int64_t *lsi,
                                const float *loc, const float *attn, const float *grad_out, int64_t N,
                                int64_t S, int64_t M, int64_t D, int64_t L, int64_t Lq, int64_t P,
                                float *grad_value, float *grad_loc, float *grad_attn, void *ws,
                                int64_t ws_bytes, void *stream);
int vah_msda_fused_backward_tiled(const void *value, int value_dtype, const int64_t *shapes,
                                  const int64_t *lsi, const void *offsets, const void *logits,
                                  int param_dtype, int64_t offsets_stride, int64_t logits_stride,
                                  const float *ref, int64_t ref_levels,
                                  const void *grad_out, int64_t N, int64_t S, int64_t M, int64_t D,
                                  int64_t L, int64_t Lq, int64_t P, void *grad_value,
                                  int grad_value_dtype, void *d_offsets, void *d_logits,
                                  int grad_param_dtype, int64_t d_offsets_stride, int64_t d_logits_stride,
                                  void *ws, int64_t ws_bytes, void *stream);

/* ------------------------------------------------------------------------------------
 * Softmax attention of the ViT blocks, bf16, head_dim 64  
 *
 * Replaces the score / softmax / value products of the reference's Attention and
 * WindowedAttention (/root/reference/detection/mmdet_custom/models/backbones/base/vit.py:83-88
 * and :154-159): out = softmax(q k^T * scale) v per (batch, head), no mask, no dropout.  The
 * N x N score matrix is never materialised.
 *
 *   q, k, v  bf16, read in place from the fused projection: element (b, n, h, d) of each lives at
 *            ptr[b*batch_stride + n*ld + h*64 + d]   (ld = 3*heads*64 for a packed qkv buffer)
 *   out      bf16 (B, N, heads, 64) with row stride ld_out elements
 *   lse      fp32 (B, heads, N): log2-domain log-sum-exp of the scaled scores (kept for backward)
 *   vt_ws    bf16 workspace of B*heads*64*vah_attn_padded_len(N) elements (V transposed)
 * ------------------------------------------------------------------------------------ */
int64_t vah_attn_padded_len(int64_t N);
int vah_attn_fwd_bf16(const void *q, const void *k, const void *v, int64_t ld, int64_t batch_stride,
                      int64_t B, int64_t H, int64_t N, float scale, void *vt_ws,
                      void *out, int64_t ld_out, float *lse, void *stream);

/* Windowed variant (WindowedAttention, base/vit.py:136-167): q, k, v are the fused projection of
 * a (B, grid_h, grid_w) token grid in its natural row-major token order; the kernels cut it into
 * win x win windows on the fly (grid padded up to a multiple of win; padded tokens read as zero
 * q = k = v rows - the reference pads AFTER the projection - take part in the softmax unmasked and
 * are never written).  No pad / unfold / fold / crop copies.  out uses the same token order.
 * Sequences Z = B * ceil(grid_h/win) * ceil(grid_w/win), N = win*win tokens each:
 * lse is (Z, heads, N), vt_ws holds Z*heads*64*vah_attn_padded_len(N) bf16.
 * win*win <= 224 (the reference's 14 x 14 windows): ONE kernel, one workgroup per (window, head) with K and V of the
 * window resident in LDS (csrc/attn_win.hip); vt_ws is not touched and may be NULL. */
int vah_attn_win_fwd_bf16(const void *q, const void *k, const void *v, int64_t ld, int64_t B,
                          int64_t grid_h, int64_t grid_w, int64_t win, int64_t H, float scale,
                          void *vt_ws, void *out, int64_t ld_out, float *lse, void *stream);

/* Backward of the same op (the reference differentiates the materialised softmax with autograd;
 * here the probabilities are recomputed from q, k and lse).  dq / dk / dv are bf16 and use the
 * addressing of q / k / v with (ld_d, batch_stride_d): pass the three slices of one packed
 * (B, N, 3, heads, 64) buffer to get the gradient of the fused qkv projection output directly.
 * ws: vah_attn_bwd_workspace_bytes(B, H, N) bytes, 16-byte aligned.  No atomics: results are
 * bitwise reproducible. */
int64_t vah_attn_bwd_workspace_bytes(int64_t B, int64_t H, int64_t N);
int vah_attn_bwd_bf16(const void *q, const void *k, const void *v, int64_t ld, int64_t batch_stride,
                      const void *out, const void *dout, int64_t ld_out, const float *lse,
                      int64_t B, int64_t H, int64_t N, float scale, void *ws,
                      void *dq, void *dk, void *dv, int64_t ld_d, int64_t batch_stride_d,
                      void *stream);
/* Attention with an additive bias per (head, query, key) - BEiT's relative position bias and class token
 * (segmentation/mmseg_custom/models/backbones/base/beit.py:120-144: softmax(q k^T * scale + bias) v).
 * bias: bf16 (heads, N, ldb), the bias TIMES log2(e), ldb a multiple of 64 >= N (columns beyond N ignored);
 * bias_t: the same matrix transposed per head ((heads, N keys, ldb queries)); ds_out (B, heads, N, ldb) bf16 receives
 * d loss / d bias of every image (sum over B = the bias gradient; columns beyond N undefined);
 * delta_ws: B * heads * N floats.  Same kernels as vah_attn_*_bf16 (csrc/attn_flash.hip). */
int vah_attn_bias_fwd_bf16(const void *q, const void *k, const void *v, int64_t ld, int64_t batch_stride, int64_t B, int64_t H,
                           int64_t N, float scale, const void *bias, int64_t ldb, void *out, int64_t ld_out, float *lse,
                           void *stream);
int vah_attn_bias_bwd_bf16(const void *q, const void *k, const void *v, int64_t ld, int64_t batch_stride, const void *out,
                           const void *dout, int64_t ld_out, const float *lse, int64_t B, int64_t H, int64_t N, float scale,
                           const void *bias, const void *bias_t, int64_t ldb, void *ds_out, float *delta_ws, void *dq, void *dk,
                           void *dv, int64_t ld_d, int64_t batch_stride_d, void *stream);
/* BEiT's relative position bias around those two calls (csrc/relpos.hip; base/beit.py:120-131):
 * vah_relpos_bias_build: bias[h][i][j] = table[index[i][j]][h] * log2(e) as bf16 (heads, N, ldb) and its per-head
 *   transpose - the `bias` / `bias_t` operands above - from the (T, heads) fp32 table and the (N, N) int64 index;
 * vah_relpos_bias_grad: dtable (T, heads) fp32 = the scatter of sum_B ds_out back through the index
 *   (ws: vah_relpos_bias_grad_ws_floats(T, heads); LDS-bin accumulation: reproducible to fp32 rounding, not bitwise). */
int vah_relpos_bias_build(const float *table, const int64_t *index, int64_t T, int64_t H, int64_t N, int64_t ldb, void *bias,
                          void *bias_t, void *stream);
int64_t vah_relpos_bias_grad_ws_floats(int64_t T, int64_t H);
int vah_relpos_bias_grad(const void *ds, const int64_t *index, int64_t B, int64_t H, int64_t N, int64_t ldb, int64_t T, float *ws,
                         float *dtable, void *stream);
/* ws: vah_attn_bwd_workspace_bytes(Z, heads, win*win) bytes.  win*win <= 224: ONE kernel (delta, dQ, dK, dV; Q, K,
 * V, dO of the window resident in LDS), ws is not touched and may be NULL; profiler row "attn_win_bwd_bf16". */
int vah_attn_win_bwd_bf16(const void *q, const void *k, const void *v, int64_t ld,
                          const void *out, const void *dout, int64_t ld_out, const float *lse,
                          int64_t B, int64_t grid_h, int64_t grid_w, int64_t win, int64_t H, float scale,
                          void *ws, void *dq, void *dk, void *dv, int64_t ld_d, void *stream);

/* ------------------------------------------------------------------------------------
 * Fused memory-bound operators of the blocks.  They replace
 * chains of separate PyTorch elementwise / cast / copy kernels in the reference's graph; each is
 * one pass over its operands.  All tensors dense row-major; C % 4 == 0.
 *
 * LayerNorm over the last dim, fp32 in -> bf16 out (reference: nn.LayerNorm(eps=1e-6) followed by
 * autocast's fp32->bf16 cast in front of every Linear; detection/.../base/vit.py:301-306).
 * mean / rstd (rows) are saved for the backward.  Column reductions (dw, db, dgamma, conv weight
 * grads) go through per-workgroup partial rows in a scratch buffer `ws` of
 * vah_reduce_ws_floats(K) floats (K = number of reduced columns) and a finalize kernel: no atomics,
 * bitwise reproducible, outputs are overwritten.
 */
int64_t vah_reduce_ws_floats(int64_t K);
int vah_layernorm_fwd_f32_bf16(const float *x, const float *w, const float *b, int64_t rows, int64_t C,
                               float eps, void *y_bf16, float *mean, float *rstd, void *stream);
/* gres (fp32 (rows, C) or NULL): gradient that reaches x along the residual branch; when given,
 * dx = gres + LayerNorm'(g) - the sum autograd would otherwise form with a separate add. */
int vah_layernorm_bwd_f32_bf16(const float *x, const void *g_bf16, const float *w, const float *mean,
                               const float *rstd, const float *gres, int64_t rows, int64_t C,
                               float *dx, float *dw, float *db, float *ws /* K = 2C */, void *stream);
/* Residual update + LayerNorm in one pass (the pattern  x = x + drop_path(gamma * f(..)); h = norm(x)
 * of consecutive sub-blocks, base/vit.py:301-306, adapter_modules.py:112-117):
 *   fwd:  t = x + sc[b] * gamma * z (fp32, written),  h = LayerNorm(t) (bf16)
 *   bwd:  dt = gt + LayerNorm'(gh)  (= dx),  dz = sc * gamma * dt (bf16),  dgamma = sum sc * dt * z, dw, db
 * x, t (batch, rows_per_batch, C) fp32; z bf16; gamma (C), sc (batch), gt optional (NULL).
 * ws: vah_reduce_ws_floats(3 * C)  (partial rows [dw | db | dgamma]). */
int vah_residual_layernorm_fwd(const float *x, const void *z_bf16, const float *gamma, const float *sc,
                               int64_t batch, int64_t rows_per_batch, int64_t C, const float *w, const float *b,
                               float eps, float *t, void *h_bf16, float *mean, float *rstd, void *stream);
int vah_residual_layernorm_bwd(const float *t, const void *gh_bf16, const float *w, const float *mean,
                               const float *rstd, const float *gt, const void *z_bf16, const float *gamma,
                               const float *sc, int64_t batch, int64_t rows_per_batch, int64_t C, float *dt,
                               void *dz_bf16, float *dgamma, float *dw, float *db, float *ws, void *stream);
/* Two LayerNorms of the SAME fp32 rows with different affine parameters and equal eps (the adapter
 * normalises c with injector.feat_norm and again, unchanged, with extractor.query_norm,
 * adapter_modules.py:112-117,141-146): shared statistics, one read of x for both bf16 outputs; backward
 * dx = gres + LN_a'(ga) + LN_b'(gb) in one pass (ga / gb / gres optional), dparams (4, C) = [dwa|dba|dwb|dbb].
 * ws: vah_reduce_ws_floats(2 * C). */
int vah_layernorm_dual_fwd(const float *x, const float *wa, const float *ba, const float *wb, const float *bb,
                           int64_t rows, int64_t C, float eps, void *ya_bf16, void *yb_bf16, float *mean,
                           float *rstd, void *stream);
int vah_layernorm_dual_bwd(const float *x, const void *ga_bf16, const void *gb_bf16, const float *wa, const float *wb,
                           const float *mean, const float *rstd, const float *gres, int64_t rows, int64_t C,
                           float *dx, float *dparams, float *ws, void *stream);
/* out[c] = sum_r g[r][c] of a bf16 (rows, C) matrix, C % 8 == 0: the bias gradient of nn.Linear
 * (what autograd computes as grad_output.sum(0)); ws K = C. */
int vah_colsum_bf16(const void *g_bf16, int64_t rows, int64_t C, float *out, float *ws, void *stream);
/* Only the partial rows (ws: vah_reduce_ws_floats(C), *nparts rows of C floats), for a consumer that sums them
 * itself: vah_gemm_bf16_fin does it on its last launch. */
int vah_colsum_bf16_partials(const void *g_bf16, int64_t rows, int64_t C, float *ws, int64_t *nparts, void *stream);
/* fp32, over `batch` row blocks of a strided tensor: out[c] = sum_{b, r < rows} g[b * batch_stride + r * C + c]
 * (the gradient of a per-channel vector added to a token range of a (B, T, C) tensor); C % 4 == 0. */
int vah_colsum_f32(const float *g, int64_t batch, int64_t batch_stride, int64_t rows, int64_t C, float *out,
                   float *ws, void *stream);

/* ---- output tail: BatchNorm(a + b + bilinear_upsample_s(x)) (csrc/tail_ops.hip) ---------------
 * Reference: vit_adapter.py:106-127 (seg) / :101-120 (det):  c1 = up(c2) + c1;  c1 = c1 +
 * F.interpolate(x1, scale_factor=4, mode='bilinear', align_corners=False);  f1 = norm1(c1)  and the
 * same for the stride-8 / stride-16 maps.  The sum is never stored: every pass recomputes it.
 * a (N, C, H, W) bf16 or fp32; b optional, same shape; x optional fp32 (N, C, H/s, W/s),
 * s in {1, 2, 4, 8}; W % 4 == 0 and (W/s) % 4 == 0.
 *   stats:      sums[0:C] = sum t, sums[C:2C] = sum t^2 over (N, H, W)          (ws: vah_bn_tail_ws_floats(C))
 *   apply:      y = (t - mean) * rstd * gamma + beta          (fp32; gamma / beta may be NULL)
 *   bwd_stats:  sums[0:C] = sum dy, sums[C:2C] = sum dy * xhat
 *   bwd_apply:  dt = gamma * rstd * (dy - mdy - xhat * mdyx);  da, db <- dt (own dtypes, may be NULL);
 *               dxlo += upsample^T(dt)  (fp32, zero-filled by the caller; NULL to skip)
 * The caller owns the statistics between the passes (SyncBatchNorm all-reduces them there). */
int64_t vah_bn_tail_ws_floats(int64_t C);
int vah_bn_tail_stats(const void *a, int a_bf16, const void *b, int b_bf16, const float *x, int scale,
                      int64_t N, int64_t C, int64_t H, int64_t W, const float *shift, float *sums, float *ws,
                      void *stream);
int vah_bn_tail_apply(const void *a, int a_bf16, const void *b, int b_bf16, const float *x, int scale,
                      int64_t N, int64_t C, int64_t H, int64_t W, const float *mean, const float *rstd,
                      const float *gamma, const float *beta, int relu, const float *shift, void *y, int y_bf16,
                      void *stream);
int vah_bn_tail_bwd_stats(const void *a, int a_bf16, const void *b, int b_bf16, const float *x, int scale,
                          int64_t N, int64_t C, int64_t H, int64_t W, const float *mean, const float *rstd,
                          const float *gamma, const float *beta, int relu, const float *shift, const void *dy,
                          int dy_bf16, float *sums, float *ws, void *stream);
int vah_bn_tail_bwd_apply(const void *a, int a_bf16, const void *b, int b_bf16, const float *x, int scale,
                          int64_t N, int64_t C, int64_t H, int64_t W, const float *mean, const float *rstd,
                          const float *gamma, const float *beta, int relu, const float *shift, const void *dy,
                          int dy_bf16, const float *mdy, const float *mdyx, void *da, void *db, float *dxlo,
                          void *stream);
/* relu != 0: y = max(0, BatchNorm(t)) - the conv -> SyncBN -> ReLU triples of the SpatialPriorModule
 * (adapter_modules.py:217-241) with b = x = NULL; the backward recomputes y's sign from a, no mask is
 * stored.  y / dy are fp32 or bf16 (y_bf16 / dy_bf16).
 * shift (C floats or NULL): a per-channel constant added to the sum - the biases of the convolutions
 * that produce a and b (ConvTranspose2d `up`, the SPM's 1x1 `fc1`): applying them here instead of in
 * 100 M-element bias-add passes (and their bias-gradient reductions) changes nothing, BatchNorm
 * subtracts the channel mean anyway.
 * vah_bn_finalize_stats: sums = [sum (C) | sum of squares (C) | element count (1)] -> mean, rstd (biased
 * variance) and, when given, the running statistics (momentum, unbiased variance) in one launch. */
int vah_bn_finalize_stats(const float *sums, int64_t C, float eps, float momentum, float *running_mean,
                          float *running_var, float *mean, float *rstd, void *stream);

/* ConvTranspose2d(k = 2, stride 2) - the backbone's `up` (vit_adapter.py:46, 106-109) - runs as GEMMs on token rows
 * (vah_gemm_bf16: U (B, 4*C, h*w), rows (dy, dx, co), = Wcat x rows^T); this pass interleaves the 2 x 2 sub-pixels:
 * planes[b][co][2y+dy][2x+dx] = U[b][(2dy+dx)*C + co][y*w + x]  (inverse != 0: U <- planes, for the backward). */
int vah_pixel_shuffle2_bf16(const void *src, int64_t B, int64_t C, int64_t h, int64_t w, void *dst, int inverse, const void *add,
                            void *stream);      /* add: planes-shaped bf16 addend of the forward (`up(c2) + c1`), or NULL */

/* Token rows <-> NCHW planes.  to_planes != 0: dst (B, C, T) <- src (B, T_total, C) rows [t0, t0 + T);
 * else dst (B, T_total, C) rows [t0, t0 + T) <- src (B, C, T) + vec[C] (vec optional).  Tokens fp32, planes
 * fp32 or bf16.  Replaces c[:, a:b].transpose(1, 2).view(B, C, H, W).contiguous() of the pyramid assembly
 * (vit_adapter.py:113-119), cat([fc_l(c_l).flatten(2).transpose(1, 2) + level_embed[l]]) of the SPM
 * output (vit_adapter.py:94-97, adapter_modules.py:262-268) and their backward passes. */
int vah_transpose_tokens(const void *src, int64_t B, int64_t T_total, int64_t t0, int64_t T, int64_t C, void *dst,
                         int to_planes, int planes_bf16, const float *vec, void *stream);

/* MaxPool2d(kernel 3, stride 2, padding 1) of the SPM stem (adapter_modules.py:229-230) on bf16 NCHW, planes =
 * N * C, output (H-1)/2+1 x (W-1)/2+1.  idx: one byte per output = window position (0..8, row-major) of the
 * first maximum, the element torch's max_pool2d sends the gradient to; the backward gathers (no atomics). */
int vah_maxpool3s2_fwd_bf16(const void *x, int64_t planes, int64_t H, int64_t W, void *y, void *idx, void *stream);
int vah_maxpool3s2_bwd_bf16(const void *gy, const void *idx, int64_t planes, int64_t H, int64_t W, void *gx,
                            void *stream);

/* ---- 3x3 convolutions of the SpatialPriorModule as implicit GEMMs (csrc/conv.hip) -----------------------------
 * Replaces nn.Conv2d(k=3, padding=1, stride 1 | 2, bias=False) forward and both gradients of the stem / conv2-4
 * (adapter_modules.py:217-260), NHWC bf16 operands, fp32 accumulation.
 * vah_conv_taps_nhwc_bf16 - the gather-GEMM both the forward and the input gradient are instances of:
 *   out[n][oy*OS + oy0][ox*OS + ox0][co] = sum_t sum_c w[co][t][c] * in[n][oy*S + ty[t]][ox*S + tx[t]][c]
 *   for oy < ny, ox < nx (input read as zero outside IH x IW).  in (N, IH, IW, Cin), w (Cout, T, Cin), out (N, OH, OW,
 *   Cout), all bf16; Cin = 16 or a multiple of 64, Cout a multiple of 64; T <= 9 taps, S, OS in {1, 2}; ty / tx are
 *   HOST arrays of T offsets in [-4, 4].
 *   forward, stride s:          S = s, T = 9, (ty, tx) = (dy - 1, dx - 1), w = weight.permute(0, 2, 3, 1), OS = 1;
 *   input gradient, stride 1:   in = dY, (ty, tx) = (1 - dy, 1 - dx), w[ci][t][co] = weight[co][ci][dy][dx];
 *   input gradient, stride 2:   one call per output parity (a, b): OS = 2, (oy0, ox0) = (a, b), S = 1, taps
 *                               {dy : dy = a + 1 mod 2} x {dx : ...} with offsets (a + 1 - dy) / 2, (b + 1 - dx) / 2.
 * vah_conv3x3_dgrad_nhwc_bf16 - the input gradient as ONE launch: gx (N, H, W, Cin) from gy (N, OH, OW, Cout) and
 *   wt[ci][dy*3 + dx][co] = weight[co][ci][dy][dx] (bf16); stride 2 runs the four output parities as launch slices.
 * vah_conv3x3_wgrad_nhwc_bf16 - dw[co][dy][dx][c] (fp32, (Cout, 3, 3, Cin)) = sum_pixels dy[p][co] x[p*S + tap - 1][c];
 *   ws: vah_conv3x3_wgrad_ws_floats(Cin, Cout) floats (per-workgroup partials, summed in a fixed order). */
int vah_conv_taps_nhwc_bf16(const void *in, int64_t N, int64_t IH, int64_t IW, int64_t Cin, const void *w, int64_t Cout,
                            int T, const int *ty, const int *tx, int S, void *out, int64_t ny, int64_t nx, int64_t OH,
                            int64_t OW, int OS, int oy0, int ox0, void *stream);
int vah_conv3x3_dgrad_nhwc_bf16(const void *gy, int64_t N, int64_t OH, int64_t OW, int64_t Cout, const void *wt, int64_t Cin,
                                int S, void *gx, int64_t H, int64_t W, void *stream);
int64_t vah_conv3x3_wgrad_ws_floats(int64_t Cin, int64_t Cout);
int vah_conv3x3_wgrad_nhwc_bf16(const void *x, int64_t N, int64_t IH, int64_t IW, int64_t Cin, const void *dy, int64_t OH,
                                int64_t OW, int64_t Cout, int S, float *ws, int64_t ws_floats, float *dw, void *stream);

/* ---- memory-bound operators of the SpatialPriorModule on NHWC bf16 (csrc/spm_nhwc.hip) -----------------------
 * The layout the convolutions above read and write: no NCHW <-> NHWC conversion anywhere in the module, and its
 * stride-8/16/32 outputs ARE the token rows the adapter consumes.
 * vah_image_to_nhwc16_bf16: x (N, 3, H, W) fp32 -> y (N, H, W, 16) bf16, channels 3..15 zero.
 * BatchNorm(+ReLU) over the rows of x (rows = N*H*W, C a power of two <= 256), training statistics:
 *   stats      sums[2C] = [sum x | sum x^2]            (ws: vah_bn_nhwc_ws_floats(C); fixed summation order)
 *   apply      y = [relu](x * rstd * w + b - mean * rstd * w)   (mean, rstd from vah_bn_finalize_stats)
 *   bwd_stats  sums[2C] = [sum g' | sum g' xhat], g' = dy where the forward output was positive (all of dy if !relu)
 *   bwd_apply  dx = w rstd (g' - mean_g - xhat mean_gx)
 * vah_maxpool3s2_nhwc_*: MaxPool2d(3, stride 2, padding 1) of the stem; idx (same shape as y, one byte per element) =
 *   window position 0..8 of the first maximum; the backward gathers (no atomics). */
int vah_image_to_nhwc16_bf16(const float *x, int64_t N, int64_t H, int64_t W, void *y, void *stream);
/* x (N, C, H, W) fp32 -> y (N * H/ps * W/ps, C*ps*ps) bf16, column (c, ky, kx): the patch embedding
 * (PatchEmbed.proj, base/vit.py:169-190: Conv2d(k = stride = ps)) then is vah_gemm_bf16 of y with the flattened filters. */
int vah_patchify_bf16(const float *x, int64_t N, int64_t C, int64_t H, int64_t W, int64_t ps, void *y, void *stream);
int64_t vah_bn_nhwc_ws_floats(int64_t C);
int vah_bn_nhwc_stats(const void *x, int64_t rows, int64_t C, float *sums, float *ws, void *stream);
int vah_bn_nhwc_apply(const void *x, int64_t rows, int64_t C, const float *mean, const float *rstd, const float *w,
                      const float *b, int relu, void *y, void *stream);
int vah_bn_nhwc_bwd_stats(const void *x, const void *dy, int64_t rows, int64_t C, const float *mean, const float *rstd,
                          const float *w, const float *b, int relu, float *sums, float *ws, void *stream);
int vah_bn_nhwc_bwd_apply(const void *x, const void *dy, int64_t rows, int64_t C, const float *mean, const float *rstd,
                          const float *w, const float *b, int relu, const float *mean_g, const float *mean_gx, void *dx,
                          void *stream);
int vah_maxpool3s2_nhwc_fwd_bf16(const void *x, int64_t N, int64_t H, int64_t W, int64_t C, void *y, void *idx, void *stream);
int vah_maxpool3s2_nhwc_bwd_bf16(const void *gy, const void *idx, int64_t N, int64_t H, int64_t W, int64_t C, void *gx,
                                 void *stream);

/* ---- bf16 GEMMs of the Linear layers (csrc/gemm.hip) ----------------------------------------
 * D (M x N, row-major, leading dimension ldd; bf16, or fp32 when d_is_f32) = op(A) op(B), bf16
 * operands, fp32 accumulation.  trans_a: A is stored (K x M) row-major and used transposed;
 * trans_b likewise (B stored (N x K)).  Replaces F.linear and its backward products
 * (torch addmm / mm on hipBLASLt): forward  y = x W^T + b   (trans_b = 1, EPI_BIAS),
 * input gradient  dx = g W   and weight gradient  dW = g^T x  (trans_a = 1, fp32 output).
 * Epilogue BIAS adds bias[N] (bf16 or fp32).  (hipBLASLt's GELU / bias-gradient epilogues were
 * measured on gfx950 and are slower than the separate kernels or unsupported for these types.)
 * The library times the hipBLASLt candidates of every new problem once on the caller's stream
 * (vah_gemm_set_tuning: mode 0 = first heuristic answer, 1 = time `candidates` heuristic answers
 * [default, 32], 2 = time every algorithm of the library) and caches the winner; the cache can be
 * dumped / loaded as text ("ta tb d32 epi bias32 M N K lda ldb ldd algo_index split us" per line).
 * Long reductions into few output tiles (K >= 4096, no bias) may run as `split` strided-batch
 * slices of K with fp32 partial products in the workspace and one reduction pass; the split is
 * part of the timed choice and bounded by the workspace.
 * workspace: caller-provided scratch private to the stream; 32 MiB for the library plus, to allow
 * split-K, split * M * N * 4 bytes. */
#define VAH_GEMM_EPI_NONE 0
#define VAH_GEMM_EPI_BIAS 1
int vah_gemm_set_tuning(int mode, int candidates);
int vah_gemm_bf16(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, const void *A, int64_t lda,
                  const void *B, int64_t ldb, void *D, int64_t ldd, int d_is_f32, int epilogue,
                  const void *bias, int bias_is_f32, void *workspace, int64_t workspace_bytes, void *stream);
int vah_gemm_bf16_fin(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, const void *A, int64_t lda,
                      const void *B, int64_t ldb, void *D, int64_t ldd, int d_is_f32, void *workspace,
                      int64_t workspace_bytes, const float *fin_part, int64_t fin_nparts, int64_t fin_C,
                      float *fin_out, void *stream);   /* + fin_out[c] = sum_p fin_part[p * fin_C + c] */
int64_t vah_gemm_library_version(void);                /* hipBLASLt build the algorithm indices belong to */
/* Tuning candidates dropped so far because the 64 x 64 corner of their result differed from the corner the heuristic's
 * first answer computes for the same operands (an algorithm that runs without an error status may still be wrong). */
int64_t vah_gemm_rejected_candidates(void);
int64_t vah_gemm_table_dump(char *buf, int64_t cap);   /* returns the size needed (incl. NUL) */
int vah_gemm_table_load(const char *text);             /* returns the number of entries, < 0 on error */
/* y = x + s[b] * gamma[c] * z   (x, y fp32 (batch, rows_per_batch, C); z bf16; gamma (C) or NULL;
 * s (batch) or NULL): the residual update  x + drop_path(gamma * branch(x))  of base/vit.py:301-306
 * and adapter_modules.py:112-117,145.  Backward: dz (bf16) and dgamma (NULL when gamma is NULL;
 * ws K = C); dx is the incoming gradient itself. */
int vah_scale_residual_fwd(const float *x, const void *z_bf16, const float *gamma, const float *s,
                           int64_t batch, int64_t rows_per_batch, int64_t C, float *y, void *stream);
int vah_scale_residual_bwd(const float *g, const void *z_bf16, const float *gamma, const float *s,
                           int64_t batch, int64_t rows_per_batch, int64_t C,
                           void *dz_bf16, float *dgamma, float *ws, void *stream);
/* ConvFFN's depthwise 3x3 (+bias) directly on the (B, 21n, C) bf16 token tensor that holds the
 * (2H,2W), (H,W), (H/2,W/2) maps back to back (segmentation/.../adapter_modules.py:72-87).
 * mode 0: y = conv(x) + bias; mode 1: input gradient (x = grad of y, flipped taps).
 * wgrad writes dw (C*9) and db (C or NULL); ws K = 10C. */
int vah_dwconv3x3_tokens_bf16(const void *x, const float *w, const float *bias, int64_t B, int64_t H,
                              int64_t W, int64_t C, int mode, void *y, void *stream);
int vah_dwconv3x3_tokens_wgrad_bf16(const void *x, const void *g, int64_t B, int64_t H, int64_t W,
                                    int64_t C, float *dw, float *db, float *ws, void *stream);

/* ------------------------------------------------------------------------------------
 * Launch timing (bench.py's roofline leg).  While enabled, every kernel launched through
 * this library is bracketed by two hipEvents recorded on the launch's own stream.
 *   vah_prof_enable(1)  : start collecting (drops anything collected before)
 *   vah_prof_enable(0)  : stop collecting
 *   vah_prof_filter(p)  : time only the entry points whose name starts with p, or with one of the
 *                         comma-separated prefixes in p ("" = all); an
 *                         event pair per launch is not free (about 2 % of a training step when every
 *                         entry point is timed), so a benchmark times the kernel it reports on
 *   vah_prof_report(..) : synchronises the recorded events and writes one text line per
 *                         kernel name:  "<name> <calls> <total_ms> <bytes> <def_bytes> <flops>\n"
 *                         (bytes: algorithmic bytes for the IO dtypes the launches ran with; def_bytes:
 *                         the operator's fp32-definition bytes; flops: matrix-core work,
 *                         0 for the HBM-bound entry points)
 *                         into buf_host (NUL terminated); returns the number of bytes the
 *                         full report needs (call again with a bigger buffer if > cap).
 * ------------------------------------------------------------------------------------ */
int vah_prof_enable(int on);
int vah_prof_filter(const char *name_prefix);
int64_t vah_prof_report(char *buf_host, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif /* VITADAPTER_HIP_H */
