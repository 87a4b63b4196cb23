/* mmae_hip.h -- C ABI of libmmae_hip.so: the MI355X (gfx950) kernels behind the MultiMAE fusion-token hot path.
 *
 * Drop-in boundary.  The reference has no FFI on this path: every entry point below replaces a
 * *composition of stock ATen ops* inside the reference's Python modules; the citation on each function names the
 * reference lines it stands in for (paths relative to the reference checkout; DSI-MM = downstream/
 * instance_segmentation/modeling/multimae, MM = pretraining/multimae, PT = pretraining).  The only consumer is
 * incomplete_multimodal_fusion_amd/_lib.py (ctypes), which wraps them in torch.autograd.Functions inside module
 * classes that keep the reference's constructor / forward signatures and state-dict names.
 *
 * Conventions
 *   - plain C: raw DEVICE pointers, sizes, strides in ELEMENTS; no torch types.  `stream` is a hipStream_t.
 *   - dtype: MMAE_F32 (0) or MMAE_BF16 (1).  fp32 accumulation everywhere.
 *   - pointers of vectorised operands must be 16-byte aligned, strides multiples of 8 (attention) or 4 (row ops).
 *   - every function returns 0 on success, MMAE_ERR_ARG (-1) for invalid arguments (nothing is launched),
 *     MMAE_ERR_LAUNCH (-2) when the HIP launch failed.  Asynchronous: no host synchronisation, no allocation.
 */
#ifndef MMAE_HIP_H
#define MMAE_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

#define MMAE_F32 0
#define MMAE_BF16 1
#define MMAE_ABI_VERSION 7   /* 2: mmae_mha_fwd takes max_k_rows; the attention stamp / variant entry points moved to csrc/mmae_internal.h.  3: + mmae_add_ln_fwd_cast;
                                mmae_mha_bwd's workspace delta_ws grew from (H, rows) to (3, H, rows) floats (see mmae_mha_bwd_ws_floats).  4: + mmae_scale_rows,
                                mmae_mha_bwd_ws_floats, mmae_gemm_nt, mmae_gemm_geglu, mmae_gemm_tn, mmae_splitk_sum_multi.  5: the optimizer control block grew
                                from 4 to 8 floats (mmae_adamw_control / _step_ctl read [4], [5]); + mmae_adamw_tick, mmae_mha_fwd_route.  6: + mmae_pad_copy_bf16_batched
                                (additive: no existing signature changed).  7: + mmae_mha_bwd_fused, mmae_mha_bwd_fused_supported, mmae_mha_bwd_fused_ws_floats
                                (additive) */
int mmae_abi_version(void);
/* hipError_t of this thread's most recent launch that returned MMAE_ERR_LAUNCH (0: none); reading resets it. */
int mmae_last_hip_error(void);

/* ---- masked multi-head attention (DSI-MM/zorro_utils.py:181-193; decoder core MM/multimae_utils.py:172-179) ----------
 * Segment ("Zorro") mask as data: sample b has nseg query segments and nseg key segments (arrays (B, nseg), global row
 * index + length).  A query in segment s < nseg-1 attends key segment s; a query in the LAST segment attends every key
 * of its sample (fusion rule, MM/multimae_crossattn.py:441-447; pool rule :489-493).  Empty key segment:
 * empty_mode 0 -> uniform over all keys (finite masked_fill, zorro_utils.py:187), 1 -> zeros (empty context, :530-543).
 * q/k/v/out element (row, head h, d) lives at ptr[row*stride + h*head_dim + d]; head_dim in {32, 64}.
 * lse: (H, q_rows_total) fp32, natural log of the softmax denominator (scaled scores).  max_q_rows / max_k_rows: upper
 * bound of the per-sample total rows (sizes the grid). */
int mmae_mha_fwd(int dtype, int head_dim, int B, int H, int nseg, const void* q, const void* k, const void* v, void* out,
                 float* lse, long q_stride, long k_stride, long v_stride, long o_stride, long q_rows_total,
                 const int* q_seg_start, const int* q_seg_len, const int* k_seg_start, const int* k_seg_len,
                 int max_q_rows, int max_k_rows, float scale, int empty_mode, void* stream);
/* which forward kernel mmae_mha_fwd launches for these arguments: 0 the fp32 kernel, 1 the bf16 tile-per-block kernel, 2 the bf16
 * sample-head kernel (mha_sh_fwd_kernel).  A query (no launch): bench.py labels its attention roofline with it. */
int mmae_mha_fwd_route(int dtype, int head_dim, int nseg, int max_q_rows, int max_k_rows, long k_stride, long v_stride);
/* backward of the above (autograd of the same lines).  delta_ws: mmae_mha_bwd_ws_floats(H, q_rows_total) fp32 scratch -- three
 * (H, q_rows_total) planes: delta and the row constants handed from the dQ kernel to the dK/dV kernel (ABI 3; ABI 2 took ONE plane:
 * an old-size buffer is a device out-of-bounds write, the library cannot check it -- size it with the function). */
long mmae_mha_bwd_ws_floats(int H, long q_rows_total);
int mmae_mha_bwd(int dtype, int head_dim, int B, int H, int nseg, const void* q, const void* k, const void* v,
                 const void* out, const void* dout, const float* lse, float* delta_ws, void* dq, void* dk, void* dv,
                 long q_stride, long k_stride, long v_stride, long o_stride, long do_stride, long dq_stride,
                 long dk_stride, long dv_stride, long q_rows_total, const int* q_seg_start, const int* q_seg_len,
                 const int* k_seg_start, const int* k_seg_len, int max_q_rows, int max_k_rows, float scale,
                 int empty_mode, void* stream);
/* The same backward as ONE kernel (bf16, head_dim 64; autograd of DSI-MM/zorro_utils.py:181-193 again): dQ, dK and dV from five tile
 * products -- S and dP once, key-stationary, dS handed through LDS for dQ (mha_sh_bwd_kernel) -- after a pre-pass that writes the row
 * constants delta = rowsum(dO o O), -lse log2 e, -delta.  Arguments as mmae_mha_bwd; delta_ws must hold
 * mmae_mha_bwd_fused_ws_floats(B, H, nseg, q_rows_total, max_q_rows) floats: the three planes plus one fp32 64 x 64 partial dQ tile
 * per (sample, 64-row query tile, head), through which a query tile that meets several key passes (the fusion queries see every key)
 * carries its sum from pass to pass -- same workgroup, program order: bitwise reproducible, no atomics.
 * mmae_mha_bwd_fused_supported: 1 when these arguments fit the kernel's schedule tables (else use mmae_mha_bwd). */
long mmae_mha_bwd_fused_ws_floats(int B, int H, int nseg, long q_rows_total, int max_q_rows);
int mmae_mha_bwd_fused_supported(int dtype, int head_dim, int B, int H, int nseg, int max_q_rows, int max_k_rows);
int mmae_mha_bwd_fused(int dtype, int head_dim, int B, int H, int nseg, const void* q, const void* k, const void* v,
                       const void* out, const void* dout, const float* lse, float* delta_ws, void* dq, void* dk, void* dv,
                       long q_stride, long k_stride, long v_stride, long o_stride, long do_stride, long dq_stride,
                       long dk_stride, long dv_stride, long q_rows_total, const int* q_seg_start, const int* q_seg_len,
                       const int* k_seg_start, const int* k_seg_len, int max_q_rows, int max_k_rows, float scale,
                       int empty_mode, void* stream);

/* ---- modality attention of Block_Fusion (DSI-MM/zorro_utils.py:252-256 on MM/multimae_crossattn.py:454-462) ---------
 * For each of the B*P (sample, patch) rows: the fusion query (row of q) attends `ns` = M+1 key/value rows of kv
 * (kv[row] = [K (inner) | V (inner)]) named by slot_row (B*P, ns).  Rows < shared_base are token rows, each used by
 * exactly one slot; a masked slot of patch p uses the shared mask-embedding row shared_base + p. */
int mmae_modattn_fwd(int dtype, int head_dim, int B, int P, int ns, int inner, const void* q, long q_stride,
                     const void* kv, long kv_stride, const int* slot_row, void* out, long out_stride, float scale,
                     void* stream);
int mmae_modattn_bwd(int dtype, int head_dim, int B, int P, int ns, int inner, const void* q, long q_stride,
                     const void* kv, long kv_stride, const int* slot_row, const void* dout, long do_stride, void* dq,
                     long dq_stride, void* dkv, long dkv_stride, int shared_base, float scale, float* ws, void* stream);
/* ws: fp32 workspace of mmae_modattn_bwd_nsplit(B) * P * 2*inner floats (partial sums of the shared rows). */
int mmae_modattn_bwd_nsplit(int B);

/* ---- fused residual add + LayerNorm / double LayerNorm (DSI-MM/zorro_utils.py:103-110, :176, :124, :238-239, :255-257;
 *      decoder nn.LayerNorm eps 1e-6, MM/output_adapters_simple.py:75; final norm MM/multimae_crossattn.py:472) ------
 * x_new = x + delta (fp32 residual stream; delta optional, dtype_delta);  y = LN2(LN1(x_new)) (gamma2 == NULL: single
 * LN), betas optional.  stats: (rows, 4) = mean1, rstd1, mean2, rstd2.  D multiple of 4, <= 1024. */
int mmae_add_ln_fwd(int dtype_delta, int dtype_y, long rows, int D, const float* x, const void* delta, float* x_new,
                    void* y, const float* gamma1, const float* beta1, float eps1, const float* gamma2,
                    const float* beta2, float eps2, float* stats, void* stream);
/* The same pass with an fp32 y AND its bf16 copy y_bf16: the final norm (MM/multimae_crossattn.py:472), whose fp32 output the
 * model returns (ori_tokens / enc_fus, :495, :504) while the attention pool and the decoders consume it in the compute dtype
 * (the cast an autocast-ed nn.Linear applies to its input, :475-527).  Bias-less LayerNorms of width 768 / 1024 only
 * (MMAE_ERR_ARG otherwise: the caller casts separately). */
int mmae_add_ln_fwd_cast(int dtype_delta, long rows, int D, const float* x, const void* delta, float* x_new, float* y,
                         void* y_bf16, const float* gamma1, float eps1, const float* gamma2, float eps2, float* stats,
                         void* stream);
/* gx = LN-backward(gy) + gx_up (optional); written as fp32 (gx) and/or dtype_delta (gdelta).  Column sums
 * dgamma1/2, dbeta1/2 (fp32, D; += when accumulate != 0) via workspace ws of mmae_add_ln_bwd_ws_floats(rows, D) floats. */
int mmae_add_ln_bwd_ws_floats(long rows, int D);
int mmae_add_ln_bwd(int dtype_delta, int dtype_y, long rows, int D, const float* x_new, const void* gy,
                    const float* gx_up, const float* gamma1, const float* beta1, const float* gamma2,
                    const float* stats, float* gx, void* gdelta, float* dgamma1, float* dbeta1, float* dgamma2,
                    float* dbeta2, float* ws, int accumulate, void* stream);

/* Two double LayerNorms of the SAME residual rows in one pass: path a = Block_Fusion's (norm1, attn.norm) for the modality
 * attention's K/V, path b = Block's (norm1, attn.norm) for the Zorro attention (MM/multimae_crossattn.py:454-470 with
 * DSI-MM/zorro_utils.py:238, :255; the modality rows are not changed between the two).  Bias-less LayerNorms only.
 * Same results as two mmae_add_ln_fwd / mmae_add_ln_bwd calls (the first with delta, the second without). */
int mmae_add_ln_fwd_dual(int dtype_delta, int dtype_y, long rows, int D, const float* x, const void* delta, float* x_new,
                         void* y_a, void* y_b, const float* gamma1_a, const float* gamma2_a, const float* gamma1_b,
                         const float* gamma2_b, float eps1, float eps2, float* stats_a, float* stats_b, void* stream);
int mmae_add_ln_bwd_dual(int dtype_delta, int dtype_y, long rows, int D, const float* x_new, const void* gy_a,
                         const void* gy_b, const float* gx_up, const float* gamma1_a, const float* gamma2_a,
                         const float* gamma1_b, const float* gamma2_b, const float* stats_a, const float* stats_b,
                         float* gx, void* gdelta, float* dgamma1_a, float* dgamma2_a, float* dgamma1_b, float* dgamma2_b,
                         float* ws, int accumulate, void* stream);

/* ---- GEGLU (DSI-MM/zorro_utils.py:115-118): out[r, j] = gelu(h[r, F + j]) * h[r, j]; erf GELU, Phi to 1.5e-7 - */
int mmae_geglu_fwd(int dtype, long rows, int F, const void* h, void* out, void* stream);
int mmae_geglu_bwd(int dtype, long rows, int F, const void* h, const void* gout, void* dh, void* stream);
/* ---- dense projections (the nn.Linear calls of DSI-MM/zorro_utils.py:181-182,192 (to_q / to_kv / to_out), :125-127 (FeedForward) and
 * autograd's dX = dY . W for them): C[M, N] = A[M, K] . W[N, K]^T, bf16 in / bf16 out, fp32 accumulate, rows contiguous with leading
 * dimensions lda / ldw / ldc (elements).  Shapes: any M, N % 256 == 0, K % 128 == 0, K >= 384, every byte offset < 2 GiB --
 * mmae_gemm_nt_supported() says whether a shape qualifies (callers use the library GEMM otherwise).  One persistent launch, 128 KB LDS. */
int mmae_gemm_nt_supported(long M, long N, long K, long lda, long ldw, long ldc);
int mmae_gemm_nt(long M, long N, long K, const void* A, long lda, const void* W, long ldw, void* C, long ldc, void* stream);
/* Weight gradients of the same layers (autograd's dW = dY^T X): out[N, Kin] (fp32) = G[rows, N]^T . X[rows, Kin], bf16 operands contracted
 * over their rows (transposing LDS reads), split-K over workgroups with the fp32 slabs summed in a fixed order (bitwise reproducible).
 * N % 256 == 0, Kin % 256 == 0, rows >= 128, byte offsets < 2 GiB.  ws: mmae_gemm_tn_ws_floats(rows, N, Kin) floats (may be 0 -> NULL). */
int mmae_gemm_tn_supported(long rows, long N, long Kin, long ldg, long ldx);
long mmae_gemm_tn_ws_floats(long rows, long N, long Kin);
int mmae_gemm_tn(long rows, long N, long Kin, const void* G, long ldg, const void* X, long ldx, float* out, float* ws, void* stream);
/* FeedForward[1] + GEGLU in one kernel (DSI-MM/zorro_utils.py:115-118,125-126): W1 is (2 F, K), val rows [0, F), gate rows [F, 2 F);
 * h[M, 2 F] = A . W1^T (kept: the backward's GEGLU' reads it) and g[M, F] = gelu(h[:, F + n]) * h[:, n] from the bf16-rounded h, i.e.
 * exactly what mmae_geglu_fwd computes from h.  F % 128 == 0; the other constraints as above. */
int mmae_gemm_geglu_supported(long M, long F, long K, long lda, long ldw, long ldh, long ldg);
int mmae_gemm_geglu(long M, long F, long K, const void* A, long lda, const void* W1, long ldw, void* h, long ldh, void* g, long ldg, void* stream);
/* ---- DropPath / stochastic depth (DSI-MM/zorro_utils.py:69-84 on :238-239; MM/multimae_utils.py:105-132): out[r, :] =
 * x[r, :] * row_scale[r], row_scale = floor(keep + u_sample) / keep per sample, rows in the packed row space.  Its own backward. */
int mmae_scale_rows(int dtype, long rows, int W, const void* x, const float* row_scale, void* out, void* stream);
/* ---- GELU of Mlp (DSI-MM/zorro_utils.py:141-143, MM/multimae_utils.py:148-150) -------------------------------------- */
int mmae_gelu_fwd(int dtype, long n, const void* x, void* y, void* stream);
int mmae_gelu_bwd(int dtype, long n, const void* x, const void* g, void* dx, void* stream);

/* ---- column sums: out[c] = sum_r x[r*ld + c] in fp32, deterministic two-stage (bias gradients = autograd of the `+ bias` of
 *      nn.Linear in the decoders / Mlp heads, MM/multimae_utils.py:138-182, MM/output_adapters_simple.py:166-181).
 *      cols and ld multiples of 8 (bf16) / 4 (fp32); ws: mmae_colsum_ws_floats(rows, cols) floats. */
long mmae_colsum_ws_floats(long rows, int cols);
int mmae_colsum(int dtype, long rows, int cols, const void* x, long ld, float* out, float* ws, void* stream);

/* ---- row gather / scatter (token selection MM/multimae_crossattn.py:402-407, :531-541; pos-emb lookup) ------------- */
int mmae_gather_rows(int dtype, long rows, int W, const void* src, long src_stride, const int* idx, void* out,
                     long out_stride, void* stream);
/* dst[idx[r]] (+)= src[r] for rows with idx[r] >= 0 and (filter == NULL or filter[r] == filter_value); plain
 * read-modify-write: destination rows must be unique within one call (call once per modality when they are not). */
int mmae_scatter_rows(int dtype, long rows, int W, const void* src, long src_stride, const int* idx, void* dst,
                      long dst_stride, int accumulate, const int* filter, int filter_value, void* stream);

/* ---- patchify of KEPT patches (MM/input_adapters.py:104-110 + MM/multimae_crossattn.py:402-407) ---------------------
 * out (B*tokens_per_sample, Kcat): row r of sample r / tokens_per_sample, modality tok_mod[r], patch tok_patch[r]:
 * pixels in (c ph pw) order at columns [col_offsets[m], +C_m*patch^2), zeros elsewhere, optional one-hot(modality) at
 * onehot_offset (bias column of the concatenated conv weight).  tok_mod == tok_patch == NULL: dense, one modality. */
int mmae_patchify_gather(int dtype_out, int nmod, const float* const* images, const int* channels,
                         const int* col_offsets, int onehot_offset, int Kcat, int B, int H, int W, int patch,
                         const int* tok_mod, const int* tok_patch, int tokens_per_sample, void* out, void* stream);
/* ---- unpatchify (MM/output_adapters_simple.py:183-186) -------------------------------------------------------------- */
int mmae_unpatchify(int dtype_in, int B, int C, int H, int W, int patch, const void* tokens, float* image, void* stream);

/* ---- masked reconstruction loss (MM/criterion.py:98-111 MSE kind 0, :155-168 L1 kind 1) -----------------------------
 * pred: image (B,C,H,W) fp32 (pred_is_tokens 0) or decoder tokens (B*P, C*patch^2) of pred_dtype (fused unpatchify).
 * mask (B, P) int64 {0,1} or NULL (= all ones).  stats[0] = loss, stats[1] = number of samples with a non-empty mask;
 * partial_ws (B*P) and den (B) fp32 are kept for the backward.  Nothing masked -> loss 0 (criterion.py:101-102). */
int mmae_masked_loss_fwd(int pred_dtype, int pred_is_tokens, int kind, int B, int C, int H, int W, int patch,
                         const void* pred, const float* target, const long long* mask, float* partial_ws, float* den,
                         float* stats, void* stream);
int mmae_masked_loss_bwd(int pred_dtype, int pred_is_tokens, int kind, int B, int C, int H, int W, int patch,
                         const void* pred, const float* target, const long long* mask, const float* den,
                         const float* stats, const float* gloss, void* gpred, void* stream);

/* ---- masked cross-entropy (MM/criterion.py:24-58; the `dnw` class-map modality of PT/pretrain_mmae_my.py:67-74) ---------
 * pred: logits, image (B,C,H,W) fp32 or decoder tokens (B*P, C*patch^2) of pred_dtype; target (B,H,W) int64 class ids;
 * per-pixel CE with label smoothing, masked / reduced exactly like mmae_masked_loss_* (no mean over C). */
int mmae_masked_ce_loss_fwd(int pred_dtype, int pred_is_tokens, int B, int C, int H, int W, int patch, const void* pred,
                            const long long* target, const long long* mask, float label_smoothing, float* partial_ws,
                            float* den, float* stats, void* stream);
int mmae_masked_ce_loss_bwd(int pred_dtype, int pred_is_tokens, int B, int C, int H, int W, int patch, const void* pred,
                            const long long* target, const long long* mask, float label_smoothing, const float* den,
                            const float* stats, const float* gloss, void* gpred, void* stream);

/* ---- contrastive heads: dino_loss_func (MM/criterion.py:328-335), HardNegtive_loss (MM/criterion.py:233-268) -------- */
int mmae_dino_loss_fwd(int B, int D, const float* student, const float* teacher, float student_temp,
                       float teacher_temp, float* row_loss_ws, float* loss, void* stream);
int mmae_dino_loss_bwd(int B, int D, const float* student, const float* teacher, float student_temp,
                       float teacher_temp, const float* gloss, float* gstudent, void* stream);
long mmae_hardneg_ws_floats(int B, int D);
int mmae_hardneg_loss_fwd(int B, int D, const float* out_1, const float* out_2, float tau_plus, float beta,
                          float temperature, float* ws, float* loss, void* stream);
int mmae_hardneg_loss_bwd(int B, int D, const float* out_1, const float* out_2, float tau_plus, float beta,
                          float temperature, float* ws, const float* gloss, float* g1, float* g2, void* stream);

/* ---- fused AdamW over flat buffers (PT/utils/optim_factory.py:136-179 + PT/utils/native_scaler.py:20-40, :49-62) ------
 * torch.optim.AdamW update rule on n fp32 elements (n % 4 == 0, 16-byte aligned); g is multiplied by grad_scale first
 * (loss-scale / averaging); `shadow_bf16` (optional) receives bf16(p) -- the weights the next forward's GEMMs read. */
int mmae_adamw_step(long n, float* p, const float* g, float* m, float* v, void* shadow_bf16, float lr, float beta1,
                    float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream);
/* Device-side gradient clipping / step skipping (PT/utils/native_scaler.py:20-40: clip_grad -> clip_grad_norm_, skip_grad ->
 * no optimizer step when norm >= skip_grad; torch GradScaler: no step on a non-finite gradient) without the host round trip.
 * mmae_adamw_control reads the norm mmae_grad_norm left on the device and fills ctl8 (8 floats since ABI 5 -- ABI 4 took 4 --,
 * zero-initialised once by the caller): [0] gradient multiplier = grad_scale * min(1, max_norm / (norm*|grad_scale| + 1e-6))
 * (max_norm 0: no clipping), [1] 1 when the step is skipped (norm non-finite and check_finite, or skip_norm > 0 and norm >=
 * skip_norm), [2] running count of skipped steps, [3] the unscaled norm.  mmae_adamw_step_ctl is mmae_adamw_step taking the
 * multiplier / skip flag from ctl8 and bias-correcting with step - ctl8[2], i.e. exactly as if optimizer.step() had not been
 * called for skipped steps.
 * Steps captured in a hipGraph replay their launches with the captured by-value arguments, so the two that change per step live
 * in ctl8 as well: [4] replays so far -- mmae_adamw_tick adds 1, captured at the top of the step; mmae_adamw_step_ctl adds it to
 * `step` --, [5] / [6] the learning rate / weight decay of this replay, used when [7] != 0 (the host writes them before the
 * replay).  All stay 0 outside a graph. */
int mmae_adamw_control(const float* grad_norm, float max_norm, float skip_norm, float grad_scale, int check_finite,
                       float* ctl8, void* stream);
int mmae_adamw_tick(float* ctl8, void* stream);
int mmae_adamw_step_ctl(long n, float* p, const float* g, float* m, float* v, void* shadow_bf16, float lr, float beta1,
                        float beta2, float eps, float weight_decay, int step, const float* ctl8, void* stream);
int mmae_shadow_bf16(long n, const float* p, void* shadow_bf16, void* stream);
/* Transposed bf16 copies of many 2-D weights in ONE launch (the data-gradient GEMMs of ops._Linear read W^T; torch did one
 * `w.t().contiguous()` kernel per weight per step).  One 64x64 tile per row of `tiles`, 32 bytes each:
 * {int64 src element offset of the tile, int64 dst element offset, int32 ld_src, int32 ld_dst, int32 rows, int32 cols}
 * with rows, cols <= 64 and multiples of 8, all offsets multiples of 8 elements. */
int mmae_transpose_bf16_batched(const void* src_bf16, void* dst_bf16, const void* tiles, int n_tiles, void* stream);
/* Zero-padded (optionally transposed) bf16 copies of weight blocks in ONE launch: the own GEMM needs K % 128 == 0 and N % 256 == 0, and
 * ViT-L's GEGLU width ffi = int(1024 * 8 / 3) = 2730 (pretraining/multimae/multimae_crossattn.py:584-599 building
 * downstream/.../multimae/zorro_utils.py:121-128 FeedForward) fits neither; the optimizer engine keeps FeedForward[1] / [3] weights padded
 * to 2816 (pads zero: inert) beside the flat bf16 shadow.  One tile of <= 64 x 64 SOURCE elements per row of `tiles`, 48 bytes each:
 * {uint64 src address, uint64 dst address, int32 ld_src, int32 ld_dst (elements), int32 rows, int32 cols (any value <= 64),
 *  int32 transpose, 12 bytes reserved}.  dst[r][c] (or dst[c][r]) = src[r][c] for r < rows, c < cols; nothing else is written. */
int mmae_pad_copy_bf16_batched(const void* tiles, int n_tiles, void* stream);
/* out[i] = sum_s partials[s*n + i], fp32 accumulation in fixed order: reduction of the S bf16 partial products of a split-K
 * weight-gradient GEMM (autograd of nn.Linear in the reference) into its fp32 destination.  n % 8 == 0. */
int mmae_splitk_sum(int S, long n, const void* partials_bf16, float* out, void* stream);
/* The same for `count` gradients in one launch per 16 (host arrays of count device pointers / sizes; bitwise identical to count calls of
 * mmae_splitk_sum): a layer's weight gradients are summed together when its backward has run. */
int mmae_splitk_sum_multi(int count, const void* const* partials_bf16, float* const* outs, const int* S, const long* n, void* stream);
/* L2 norm of a flat fp32 gradient buffer (deterministic two-stage sum); partial_ws: 2048 floats. */
int mmae_grad_norm(long n, const float* g, float* partial_ws_2048, float* out_norm, void* stream);

/* ---- input staging (PT/utils/multimodal_dfc2023.py:99-139 with :25-49) ---------------------------------
 * raw (B, C, H*factor, W*factor) of in_dtype -> out (B, C, H, W) fp32.  `factor` = integer shrink (cv2.INTER_AREA = block
 * mean; uint8 input rounds the mean like cv2).  kind SAR_DB: 10*log10(x+1e-7), clip [-25,0], NaN->0, then (x-mean)/std;
 * AFFINE: nan_to_num then per-channel (x-mean[c])/std[c]; ZSCORE: nan_to_num then per-(sample, channel) tile
 * (x - mean(x)) / sqrt(var(x) + 1e-6) (mean/std ignored, may be null).  mean/std: C host floats. */
#define MMAE_STAGE_SAR_DB 0
#define MMAE_STAGE_AFFINE 1
#define MMAE_STAGE_ZSCORE 2
#define MMAE_RAW_F32 0
#define MMAE_RAW_U8 1
int mmae_stage_tiles(int kind, int in_dtype, int B, int C, int H, int W, int factor, const void* raw, float* out,
                     const float* mean, const float* stdv, void* stream);

/* ---- mask bookkeeping (MM/multimae_crossattn.py:233-272 with injected draws; :402-447, :454-462, :489-493) ---------- */
int mmae_masks_from_draws(int R, int M, int P, int N, const float* dirichlet, const float* noise,
                          const float* noise_all, long long* mask_all, long long* ids_keep, long long* ids_restore,
                          void* stream);
/* int32 descriptor buffer; section offsets (15 entries, last = total ints) by mmae_descriptor_layout (host pointer;
 * returns the total, or MMAE_ERR_ARG). */
long mmae_descriptor_layout(int B, int M, int P, int N, long* offsets15);
int mmae_build_descriptors(int B, int R, int M, int P, int N, const long long* mask_all, int* desc, void* stream);

#ifdef __cplusplus
}
#endif
#endif
