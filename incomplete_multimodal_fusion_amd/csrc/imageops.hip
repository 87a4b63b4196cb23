// Patch <-> image kernels and the masked reconstruction loss (gfx950, HBM-bound, 16 B per lane).
//
// Reference compositions replaced:
//   PatchedInputAdapter.forward (pretraining/multimae/input_adapters.py:104-110): Conv2d(C->D, k=s=patch) over the whole
//     image followed by a gather of the kept tokens (multimae_crossattn.py:402-407).  Rows of a k=s conv are independent,
//     so only KEPT patches are read: patchify_gather writes one GEMM row per kept token.  All modalities share one GEMM:
//     row = [0 .. pixels (c ph pw) in the modality's column slot .. 0 | one-hot(modality)], the weight is the column
//     concatenation of the per-modality conv weights and biases -> static shapes, no host sync on the kept counts.
//   SpatialOutputAdapter.forward tail (output_adapters_simple.py:183-186): 'b (nh nw) (c ph pw) -> b c (nh ph) (nw pw)'.
//   MaskedMSELoss / MaskedL1Loss (pretraining/multimae/criterion.py:98-111, 155-168): elementwise loss, mean over C,
//     nearest-upsampled patch mask, per-sample masked mean, nanmean over samples.  Fused form reads the decoder's
//     token-major output directly (no prediction image round trip).
#include "common.hpp"
#include "mmae_hip.h"

#define IMG_MAXMOD 8

template <typename T> __device__ __forceinline__ void st4(T* p, const f32x4& v);
template <> __device__ __forceinline__ void st4<float>(float* p, const f32x4& v) { *reinterpret_cast<f32x4*>(p) = v; }
template <> __device__ __forceinline__ void st4<bf16>(bf16* p, const f32x4& v) {
    bf16x4 o; o[0] = (bf16)v[0]; o[1] = (bf16)v[1]; o[2] = (bf16)v[2]; o[3] = (bf16)v[3];
    *reinterpret_cast<bf16x4*>(p) = o;
}
template <typename T> __device__ __forceinline__ f32x4 ld4(const T* p);
template <> __device__ __forceinline__ f32x4 ld4<float>(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
template <> __device__ __forceinline__ f32x4 ld4<bf16>(const bf16* p) {
    const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}

struct PatchifyDesc {
    const float* img[IMG_MAXMOD];   // (B, C_m, H, W) fp32 NCHW
    int C[IMG_MAXMOD];
    int koff[IMG_MAXMOD];           // first column of modality m's pixel slot
    int nmod, H, W, ps, Kcat, onehot_off;   // onehot_off < 0: no one-hot columns
    const int* tok_mod; const int* tok_patch;    // (rows); null -> dense single-modality: patch = r % P, mod 0
    int tokens_per_sample;          // rows per sample (N, or P in dense mode)
    long rows;
    void* out;                      // (rows, Kcat)
};

template <typename T>
__global__ __launch_bounds__(256) void patchify_gather_kernel(PatchifyDesc d) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= d.rows) return;
    const int b = (int)(r / d.tokens_per_sample);
    const int m = d.tok_mod ? d.tok_mod[r] : 0;
    const int patch = d.tok_patch ? d.tok_patch[r] : (int)(r % d.tokens_per_sample);
    const int nw = d.W / d.ps;
    const int py = (patch / nw) * d.ps, px = (patch % nw) * d.ps;
    const int ps2 = d.ps * d.ps;
    const int k0 = d.koff[m], k1 = k0 + d.C[m] * ps2;
    const float* img = d.img[m] + (long)b * d.C[m] * d.H * d.W;
    T* out = reinterpret_cast<T*>(d.out) + r * d.Kcat;
    for (int e = 4 * lane; e < d.Kcat; e += 256) {
        f32x4 v{0.f, 0.f, 0.f, 0.f};
        if (e >= k0 && e < k1) {
            const int le = e - k0;
            const int c = le / ps2, rem = le % ps2, y = rem / d.ps, x = rem % d.ps;
            v = *reinterpret_cast<const f32x4*>(img + ((long)c * d.H + py + y) * d.W + px + x);
        } else if (d.onehot_off >= 0 && e >= d.onehot_off) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (e + j - d.onehot_off) == m ? 1.f : 0.f;
        }
        st4<T>(out + e, v);
    }
}

// tokens (B*P, C*ps*ps) in (c ph pw) order -> image (B, C, H, W) fp32
template <typename T>
__global__ __launch_bounds__(256) void unpatchify_kernel(const T* __restrict__ tok, float* __restrict__ img, long rows,
                                                         int P, int C, int H, int W, int ps) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int b = (int)(r / P), patch = (int)(r % P);
    const int nw = W / ps, py = (patch / nw) * ps, px = (patch % nw) * ps, ps2 = ps * ps, K = C * ps2;
    float* im = img + (long)b * C * H * W;
    for (int e = 4 * lane; e < K; e += 256) {
        const int c = e / ps2, rem = e % ps2, y = rem / ps, x = rem % ps;
        *reinterpret_cast<f32x4*>(im + ((long)c * H + py + y) * W + px + x) = ld4<T>(tok + r * K + e);
    }
}

extern "C" int mmae_patchify_gather(int dtype_out, int nmod, const float* const* images, const int* channels,
                                    const int* col_offsets, int onehot_offset, int Kcat, int B, int H, int W, int patch,
                                    const int* tok_mod, const int* tok_patch, int tokens_per_sample, void* out, void* stream) {
    if (dtype_out != MMAE_F32 && dtype_out != MMAE_BF16) return MMAE_ERR_ARG;
    if (nmod <= 0 || nmod > IMG_MAXMOD || !images || !channels || !col_offsets || !out) return MMAE_ERR_ARG;
    if (patch <= 0 || (patch % 4) || (H % patch) || (W % patch) || (Kcat % 4) || B <= 0 || tokens_per_sample <= 0) return MMAE_ERR_ARG;
    if ((tok_mod == nullptr) != (tok_patch == nullptr)) return MMAE_ERR_ARG;
    PatchifyDesc d{};
    for (int m = 0; m < nmod; ++m) {
        if (!images[m] || channels[m] <= 0 || (col_offsets[m] % 4) || col_offsets[m] + channels[m] * patch * patch > Kcat) return MMAE_ERR_ARG;
        d.img[m] = images[m]; d.C[m] = channels[m]; d.koff[m] = col_offsets[m];
    }
    if (onehot_offset >= 0 && ((onehot_offset % 4) || onehot_offset + nmod > Kcat)) return MMAE_ERR_ARG;
    d.nmod = nmod; d.H = H; d.W = W; d.ps = patch; d.Kcat = Kcat; d.onehot_off = onehot_offset;
    d.tok_mod = tok_mod; d.tok_patch = tok_patch; d.tokens_per_sample = tokens_per_sample;
    d.rows = (long)B * tokens_per_sample; d.out = out;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype_out == MMAE_BF16) MMAE_LAUNCH((patchify_gather_kernel<bf16>), dim3(cdiv(d.rows, 4)), dim3(256), 0, st, d);
    else MMAE_LAUNCH((patchify_gather_kernel<float>), dim3(cdiv(d.rows, 4)), dim3(256), 0, st, d);
    MMAE_CHECK_LAUNCH();
    return MMAE_OK;
}

extern "C" int mmae_unpatchify(int dtype_in, int B, int C, int H, int W, int patch, const void* tokens, float* image,
                               void* stream) {
    if (dtype_in != MMAE_F32 && dtype_in != MMAE_BF16) return MMAE_ERR_ARG;
    if (B <= 0 || C <= 0 || patch <= 0 || (patch % 4) || (H % patch) || (W % patch) || !tokens || !image) return MMAE_ERR_ARG;
    const int P = (H / patch) * (W / patch);
    const long rows = (long)B * P;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype_in == MMAE_BF16) MMAE_LAUNCH((unpatchify_kernel<bf16>), dim3(cdiv(rows, 4)), dim3(256), 0, st, (const bf16*)tokens, image, rows, P, C, H, W, patch);
    else MMAE_LAUNCH((unpatchify_kernel<float>), dim3(cdiv(rows, 4)), dim3(256), 0, st, (const float*)tokens, image, rows, P, C, H, W, patch);
    MMAE_CHECK_LAUNCH();
    return MMAE_OK;
}

// ------------------------------------------------------------------------------------------ masked reconstruction loss
// kind 0 = MSE, 1 = L1.  pred is either an image (B,C,H,W) fp32 (TOKENS=false) or decoder tokens (B*P, C*ps*ps) of type T.
// Stage 1: one wave per (b, patch) with mask == 1: partial[b*P+patch] = sum_{c,y,x} l(pred - tgt) / C  (0 for unmasked).
// Stage 2: one block: per-sample num/den, nanmean over samples with den > 0; stats = [loss, n_valid_samples]; den[b] kept.
struct LossDesc {
    const void* pred; const float* tgt; const long long* mask;   // mask (B, P) int64 {0,1}; null -> all ones
    float* partial; long rows; int P, C, H, W, ps, kind;
};

template <typename T, bool TOKENS>
__global__ __launch_bounds__(256) void masked_loss_partial_kernel(LossDesc d) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= d.rows) return;
    if (d.mask && d.mask[r] == 0) { if (lane == 0) d.partial[r] = 0.f; return; }
    const int b = (int)(r / d.P), patch = (int)(r % d.P);
    const int nw = d.W / d.ps, py = (patch / nw) * d.ps, px = (patch % nw) * d.ps, ps2 = d.ps * d.ps, K = d.C * ps2;
    const float* tg = d.tgt + (long)b * d.C * d.H * d.W;
    float acc = 0.f;
    for (int e = 4 * lane; e < K; e += 256) {
        const int c = e / ps2, rem = e % ps2, y = rem / d.ps, x = rem % d.ps;
        const long off = ((long)c * d.H + py + y) * d.W + px + x;
        const f32x4 t = *reinterpret_cast<const f32x4*>(tg + off);
        f32x4 pv;
        if (TOKENS) pv = ld4<T>(reinterpret_cast<const T*>(d.pred) + r * K + e);
        else pv = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(d.pred) + (long)b * d.C * d.H * d.W + off);
        const f32x4 df = pv - t;
#pragma unroll
        for (int j = 0; j < 4; ++j) acc += d.kind == 0 ? df[j] * df[j] : fabsf(df[j]);
    }
    acc = wave_sum(acc);
    if (lane == 0) d.partial[r] = acc / (float)d.C;
}

__global__ __launch_bounds__(256) void masked_loss_finish_kernel(const float* partial, const long long* mask, int B, int P,
                                                                 int ps, float* den, float* stats) {
    __shared__ float s_sum[256];
    __shared__ float s_cnt[256];
    float lsum = 0.f, lcnt = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) {
        float num = 0.f, dn = 0.f;
        for (int p = 0; p < P; ++p) {
            num += partial[(long)b * P + p];
            dn += mask ? (float)mask[(long)b * P + p] : 1.f;
        }
        dn *= (float)(ps * ps);
        den[b] = dn;
        if (dn > 0.f) { lsum += num / dn; lcnt += 1.f; }
    }
    s_sum[threadIdx.x] = lsum; s_cnt[threadIdx.x] = lcnt;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) { s_sum[threadIdx.x] += s_sum[threadIdx.x + o]; s_cnt[threadIdx.x] += s_cnt[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float n = s_cnt[0];
        stats[0] = n > 0.f ? s_sum[0] / n : 0.f;    // reference returns 0 when nothing is masked (criterion.py:101-102)
        stats[1] = n;
    }
}

// grad: gpred = gloss / n_valid * mask / den[b] / C * dl/dpred   (0 where den[b] == 0: the reference yields NaN there)
template <typename T, bool TOKENS>
__global__ __launch_bounds__(256) void masked_loss_bwd_kernel(LossDesc d, const float* den, const float* stats,
                                                              const float* gloss, void* gpred) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= d.rows) return;
    const int b = (int)(r / d.P), patch = (int)(r % d.P);
    const int nw = d.W / d.ps, py = (patch / nw) * d.ps, px = (patch % nw) * d.ps, ps2 = d.ps * d.ps, K = d.C * ps2;
    const bool on = !d.mask || d.mask[r] != 0;
    const float dn = den[b], nv = stats[1];
    const float coef = (on && dn > 0.f && nv > 0.f) ? gloss[0] / (nv * dn * (float)d.C) : 0.f;
    const float* tg = d.tgt + (long)b * d.C * d.H * d.W;
    for (int e = 4 * lane; e < K; e += 256) {
        const int c = e / ps2, rem = e % ps2, y = rem / d.ps, x = rem % d.ps;
        const long off = ((long)c * d.H + py + y) * d.W + px + x;
        f32x4 g{0.f, 0.f, 0.f, 0.f};
        if (coef != 0.f) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(tg + off);
            f32x4 pv;
            if (TOKENS) pv = ld4<T>(reinterpret_cast<const T*>(d.pred) + r * K + e);
            else pv = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(d.pred) + (long)b * d.C * d.H * d.W + off);
            const f32x4 df = pv - t;
#pragma unroll
            for (int j = 0; j < 4; ++j) g[j] = d.kind == 0 ? 2.f * df[j] * coef : (df[j] > 0.f ? coef : (df[j] < 0.f ? -coef : 0.f));
        }
        if (TOKENS) st4<T>(reinterpret_cast<T*>(gpred) + r * K + e, g);
        else *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(gpred) + (long)b * d.C * d.H * d.W + off) = g;
    }
}

static int loss_check(int pred_dtype, int pred_is_tokens, int kind, int B, int C, int H, int W, int patch) {
    if (pred_dtype != MMAE_F32 && pred_dtype != MMAE_BF16) return MMAE_ERR_ARG;
    if (!pred_is_tokens && pred_dtype != MMAE_F32) return MMAE_ERR_ARG;
    if (kind != 0 && kind != 1) return MMAE_ERR_ARG;
    if (B <= 0 || C <= 0 || patch <= 0 || (patch % 4) || (H % patch) || (W % patch)) return MMAE_ERR_ARG;
    return MMAE_OK;
}

extern "C" int mmae_masked_loss_fwd(int pred_dtype, int pred_is_tokens, int kind, int B, int C, int H, int W, int patch,
                                    const void* pred, const float* target, const long long* mask, float* partial_ws,
                                    float* den, float* stats, void* stream) {
    int rc = loss_check(pred_dtype, pred_is_tokens, kind, B, C, H, W, patch);
    if (rc) return rc;
    if (!pred || !target || !partial_ws || !den || !stats) return MMAE_ERR_ARG;
    const int P = (H / patch) * (W / patch);
    LossDesc d{pred, target, mask, partial_ws, (long)B * P, P, C, H, W, patch, kind};
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    dim3 grid(cdiv(d.rows, 4)), blk(256);
    if (!pred_is_tokens) MMAE_LAUNCH((masked_loss_partial_kernel<float, false>), grid, blk, 0, st, d);
    else if (pred_dtype == MMAE_BF16) MMAE_LAUNCH((masked_loss_partial_kernel<bf16, true>), grid, blk, 0, st, d);
    else MMAE_LAUNCH((masked_loss_partial_kernel<float, true>), grid, blk, 0, st, d);
    MMAE_CHECK_LAUNCH();
    MMAE_LAUNCH(masked_loss_finish_kernel, dim3(1), dim3(256), 0, st, partial_ws, mask, B, P, patch, den, stats);
    MMAE_CHECK_LAUNCH();
    return MMAE_OK;
}

extern "C" int mmae_masked_loss_bwd(int pred_dtype, int pred_is_tokens, int kind, int B, int C, int H, int W, int patch,
                                    const void* pred, const float* target, const long long* mask, const float* den,
                                    const float* stats, const float* gloss, void* gpred, void* stream) {
    int rc = loss_check(pred_dtype, pred_is_tokens, kind, B, C, H, W, patch);
    if (rc) return rc;
    if (!pred || !target || !den || !stats || !gloss || !gpred) return MMAE_ERR_ARG;
    const int P = (H / patch) * (W / patch);
    LossDesc d{pred, target, mask, nullptr, (long)B * P, P, C, H, W, patch, kind};
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    dim3 grid(cdiv(d.rows, 4)), blk(256);
    if (!pred_is_tokens) MMAE_LAUNCH((masked_loss_bwd_kernel<float, false>), grid, blk, 0, st, d, den, stats, gloss, gpred);
    else if (pred_dtype == MMAE_BF16) MMAE_LAUNCH((masked_loss_bwd_kernel<bf16, true>), grid, blk, 0, st, d, den, stats, gloss, gpred);
    else MMAE_LAUNCH((masked_loss_bwd_kernel<float, true>), grid, blk, 0, st, d, den, stats, gloss, gpred);
    MMAE_CHECK_LAUNCH();
    return MMAE_OK;
}

// ------------------------------------------------------------------------------------------ masked cross-entropy loss
// MaskedCrossEntropyLoss (pretraining/multimae/criterion.py:24-58; the `dnw` land-cover modality of the 4-modality driver,
// pretrain_mmae_my.py:67-74): per-pixel F.cross_entropy over C classes (label smoothing eps), nearest-upsampled patch mask,
// per-sample masked mean, nanmean over samples.  Same two stages as the pixel losses above (no division by C here);
// pred is the (B,C,H,W) fp32 logit image or the decoder's token-major (B*P, C*ps*ps) output, target (B,H,W) int64.
struct CEDesc {
    const void* pred; const long long* tgt; const long long* mask;
    float* partial; long rows; int P, C, H, W, ps; float smooth;
};

template <typename T, bool TOKENS>
__device__ __forceinline__ f32x4 ce_logits4(const CEDesc& d, long r, int b, int c, int pix, int py, int px) {
    const int ps2 = d.ps * d.ps;
    if (TOKENS) return ld4<T>(reinterpret_cast<const T*>(d.pred) + r * ((long)d.C * ps2) + (long)c * ps2 + pix);
    const int y = pix / d.ps, x = pix % d.ps;
    return *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(d.pred) +
                                            (((long)b * d.C + c) * d.H + py + y) * d.W + px + x);
}

template <typename T, bool TOKENS>
__global__ __launch_bounds__(256) void masked_ce_partial_kernel(CEDesc d) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= d.rows) return;
    if (d.mask && d.mask[r] == 0) { if (lane == 0) d.partial[r] = 0.f; return; }
    const int b = (int)(r / d.P), patch = (int)(r % d.P);
    const int nw = d.W / d.ps, py = (patch / nw) * d.ps, px = (patch % nw) * d.ps, ps2 = d.ps * d.ps;
    float acc = 0.f;
    for (int pix = 4 * lane; pix < ps2; pix += 256) {
        const int y = pix / d.ps, x = pix % d.ps;
        const long long* tp = d.tgt + ((long)b * d.H + py + y) * d.W + px + x;
        f32x4 mx{-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
        for (int c = 0; c < d.C; ++c) {
            const f32x4 v = ce_logits4<T, TOKENS>(d, r, b, c, pix, py, px);
#pragma unroll
            for (int j = 0; j < 4; ++j) mx[j] = fmaxf(mx[j], v[j]);
        }
        f32x4 se{0.f, 0.f, 0.f, 0.f}, sx{0.f, 0.f, 0.f, 0.f}, xt{0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < d.C; ++c) {
            const f32x4 v = ce_logits4<T, TOKENS>(d, r, b, c, pix, py, px);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                se[j] += __expf(v[j] - mx[j]);
                sx[j] += v[j];
                if ((long long)c == tp[j]) xt[j] = v[j];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // a target outside [0, C) -- F.cross_entropy's ignore_index (-100) in particular -- contributes no loss (and no
            // gradient below), as reduction='none' gives 0 for an ignored pixel; it used to be read as "logit 0"
            const bool valid = tp[j] >= 0 && tp[j] < (long long)d.C;
            const float lse = mx[j] + __logf(se[j]);
            acc += valid ? (1.f - d.smooth) * (lse - xt[j]) + d.smooth * (lse - sx[j] / (float)d.C) : 0.f;
        }
    }
    acc = wave_sum(acc);
    if (lane == 0) d.partial[r] = acc;
}

template <typename T, bool TOKENS>
__global__ __launch_bounds__(256) void masked_ce_bwd_kernel(CEDesc d, const float* den, const float* stats,
                                                            const float* gloss, void* gpred) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= d.rows) return;
    const int b = (int)(r / d.P), patch = (int)(r % d.P);
    const int nw = d.W / d.ps, py = (patch / nw) * d.ps, px = (patch % nw) * d.ps, ps2 = d.ps * d.ps;
    const bool on = !d.mask || d.mask[r] != 0;
    const float dn = den[b], nv = stats[1];
    const float coef = (on && dn > 0.f && nv > 0.f) ? gloss[0] / (nv * dn) : 0.f;
    for (int pix = 4 * lane; pix < ps2; pix += 256) {
        const int y = pix / d.ps, x = pix % d.ps;
        const long long* tp = d.tgt + ((long)b * d.H + py + y) * d.W + px + x;
        f32x4 mx{-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f}, se{0.f, 0.f, 0.f, 0.f};
        if (coef != 0.f) {
            for (int c = 0; c < d.C; ++c) {
                const f32x4 v = ce_logits4<T, TOKENS>(d, r, b, c, pix, py, px);
#pragma unroll
                for (int j = 0; j < 4; ++j) mx[j] = fmaxf(mx[j], v[j]);
            }
            for (int c = 0; c < d.C; ++c) {
                const f32x4 v = ce_logits4<T, TOKENS>(d, r, b, c, pix, py, px);
#pragma unroll
                for (int j = 0; j < 4; ++j) se[j] += __expf(v[j] - mx[j]);
            }
        }
        for (int c = 0; c < d.C; ++c) {
            f32x4 g{0.f, 0.f, 0.f, 0.f};
            if (coef != 0.f) {
                const f32x4 v = ce_logits4<T, TOKENS>(d, r, b, c, pix, py, px);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool valid = tp[j] >= 0 && tp[j] < (long long)d.C;
                    g[j] = valid ? coef * (__expf(v[j] - mx[j]) / se[j] - ((long long)c == tp[j] ? 1.f - d.smooth : 0.f) - d.smooth / (float)d.C) : 0.f;
                }
            }
            if (TOKENS) st4<T>(reinterpret_cast<T*>(gpred) + r * ((long)d.C * ps2) + (long)c * ps2 + pix, g);
            else *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(gpred) + (((long)b * d.C + c) * d.H + py + y) * d.W + px + x) = g;
        }
    }
}

static int ce_check(int pred_dtype, int pred_is_tokens, int B, int C, int H, int W, int patch, float smooth) {
    if (pred_dtype != MMAE_F32 && pred_dtype != MMAE_BF16) return MMAE_ERR_ARG;
    if (!pred_is_tokens && pred_dtype != MMAE_F32) return MMAE_ERR_ARG;
    if (B <= 0 || C <= 0 || patch <= 0 || (patch % 4) || (H % patch) || (W % patch)) return MMAE_ERR_ARG;
    if (!(smooth >= 0.f && smooth <= 1.f)) return MMAE_ERR_ARG;
    return MMAE_OK;
}

extern "C" int mmae_masked_ce_loss_fwd(int pred_dtype, int pred_is_tokens, int B, int C, int H, int W, int patch,
                                       const void* pred, const long long* target, const long long* mask,
                                       float label_smoothing, float* partial_ws, float* den, float* stats, void* stream) {
    int rc = ce_check(pred_dtype, pred_is_tokens, B, C, H, W, patch, label_smoothing);
    if (rc) return rc;
    if (!pred || !target || !partial_ws || !den || !stats) return MMAE_ERR_ARG;
    const int P = (H / patch) * (W / patch);
    CEDesc d{pred, target, mask, partial_ws, (long)B * P, P, C, H, W, patch, label_smoothing};
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    dim3 grid(cdiv(d.rows, 4)), blk(256);
    if (!pred_is_tokens) MMAE_LAUNCH((masked_ce_partial_kernel<float, false>), grid, blk, 0, st, d);
    else if (pred_dtype == MMAE_BF16) MMAE_LAUNCH((masked_ce_partial_kernel<bf16, true>), grid, blk, 0, st, d);
    else MMAE_LAUNCH((masked_ce_partial_kernel<float, true>), grid, blk, 0, st, d);
    MMAE_CHECK_LAUNCH();
    MMAE_LAUNCH(masked_loss_finish_kernel, dim3(1), dim3(256), 0, st, partial_ws, mask, B, P, patch, den, stats);
    MMAE_CHECK_LAUNCH();
    return MMAE_OK;
}

extern "C" int mmae_masked_ce_loss_bwd(int pred_dtype, int pred_is_tokens, int B, int C, int H, int W, int patch,
                                       const void* pred, const long long* target, const long long* mask,
                                       float label_smoothing, const float* den, const float* stats, const float* gloss,
                                       void* gpred, void* stream) {
    int rc = ce_check(pred_dtype, pred_is_tokens, B, C, H, W, patch, label_smoothing);
    if (rc) return rc;
    if (!pred || !target || !den || !stats || !gloss || !gpred) return MMAE_ERR_ARG;
    const int P = (H / patch) * (W / patch);
    CEDesc d{pred, target, mask, nullptr, (long)B * P, P, C, H, W, patch, label_smoothing};
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    dim3 grid(cdiv(d.rows, 4)), blk(256);
    if (!pred_is_tokens) MMAE_LAUNCH((masked_ce_bwd_kernel<float, false>), grid, blk, 0, st, d, den, stats, gloss, gpred);
    else if (pred_dtype == MMAE_BF16) MMAE_LAUNCH((masked_ce_bwd_kernel<bf16, true>), grid, blk, 0, st, d, den, stats, gloss, gpred);
    else MMAE_LAUNCH((masked_ce_bwd_kernel<float, true>), grid, blk, 0, st, d, den, stats, gloss, gpred);
    MMAE_CHECK_LAUNCH();
    return MMAE_OK;
}
