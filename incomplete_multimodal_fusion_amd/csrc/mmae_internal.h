/* mmae_internal.h -- NOT part of the product ABI (include/mmae_hip.h).  Test / tuning entry points of libmmae_hip.so:
 * the attention launchers with the kernel variant as a PER-CALL argument (no process-global state, so concurrent streams
 * and threads cannot change each other's kernel).  Consumers: tests/ (generic-vs-fast-path agreement) and
 * tools/bench_attn.py (A/B of tilings inside one process).
 *
 * variant  0  what mmae_mha_fwd / mmae_mha_bwd run
 *         -1  bf16 through the generic dtype-templated kernels of mha.hip instead of the bf16 fast path
 *        > 0  bits 0..7: alternative tilings of the bf16 fast path (see mha_bf16.hip: mha_bf16_fwd / mha_bf16_bwd; 0 = product);
 *             bits 8..11: heads ONE workgroup of the sample-head kernels (mha_sh.hip) walks, a divisor of H -- the product path
 *             picks it from (B, H) (all H heads once B >= 256); tests force 1 / 2 / H at small B to reach the head loop, the
 *             cross-head K/V prefetch and the ring accounting of the bench configuration.  0 = product choice. */
#ifndef MMAE_INTERNAL_H
#define MMAE_INTERNAL_H
#ifdef __cplusplus
extern "C" {
#endif
int mmae_mha_fwd_variant(int dtype, int head_dim, int B, int H, int nseg, const void* q, const void* k, const void* v, void* out,
                         float* lse, long q_stride, long k_stride, long v_stride, long o_stride, long q_rows_total,
                         const int* q_seg_start, const int* q_seg_len, const int* k_seg_start, const int* k_seg_len,
                         int max_q_rows, int max_k_rows, float scale, int empty_mode, int variant, void* stream);
int mmae_mha_bwd_variant(int dtype, int head_dim, int B, int H, int nseg, const void* q, const void* k, const void* v,
                         const void* out, const void* dout, const float* lse, float* delta_ws, void* dq, void* dk, void* dv,
                         long q_stride, long k_stride, long v_stride, long o_stride, long do_stride, long dq_stride,
                         long dk_stride, long dv_stride, long q_rows_total, const int* q_seg_start, const int* q_seg_len,
                         const int* k_seg_start, const int* k_seg_len, int max_q_rows, int max_k_rows, float scale,
                         int empty_mode, int variant, void* stream);
/* backward variants 50..55: the fused dQ + dK + dV kernel (product entry: mmae_mha_bwd_fused) and its diagnostic builds -- 51 no dQ part,
 * 52 no partial-tile workspace traffic, 53 no row-constant pre-pass, 54 dS exchange without the dQ products (51..54: wrong dQ, timing only);
 * 55 the row constants formed inside the kernel instead of by the pre-pass (correct results; an experiment that did not pay in the step);
 * delta_ws sized by mmae_mha_bwd_fused_ws_floats (include/mmae_hip.h). */
/* diagnostic (variant 9 of the forward): copies the 8 per-launch stamp sums to host8 and clears them (host sync). */
int mmae_debug_mha_stamps(unsigned long long* host8);
/* diagnostic (variant 8 of the forward, mha_sh.hip): 2 x 16 stamp sums (global-role waves, local-role waves), cleared on read */
int mmae_debug_sh_stamps(unsigned long long* host32);
/* bench.py's event-bracket calibration: a streaming copy of n_bytes (multiple of 16) under its own kernel name. */
int mmae_debug_stream_copy(long n_bytes, const void* src, void* dst, void* stream);
#ifdef __cplusplus
}
#endif
#endif
