// Fused AdamW over flat parameter buffers (gfx950, HBM-bound: 30 B per element, one pass).
//
// Replaces the optimizer tail of the reference step (pretraining/utils/optim_factory.py:136-179 builds torch AdamW over
// all parameters; pretraining/utils/native_scaler.py:20-40 steps it, :49-62 computes the gradient norm per parameter with a
// stack of torch.norm calls).  One launch updates the fp32 master weights, both moments and the bf16 shadow copy the
// next forward's GEMMs read -- so there is no per-parameter cast kernel and no separate multi-tensor optimizer launch.
// Update rule = torch.optim.AdamW (decoupled weight decay, bias-corrected):
//   p *= 1 - lr*wd;  m = b1*m + (1-b1)*g;  v = b2*v + (1-b2)*g*g;  p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
#include "common.hpp"
#include "mmae_hip.h"

// Device-side step control (native_scaler.py:20-40 without the host round trip of `norm >= skip_grad` / GradScaler's
// found-inf check): ctl[0] = multiplier applied to every gradient element (grad_scale x clip coefficient), ctl[1] = 1 when
// this step must not be taken, ctl[2] = number of steps skipped so far (the bias correction uses step - ctl[2], exactly as if
// optimizer.step() had not been called), ctl[3] = the unscaled gradient norm.
// Captured steps (a hipGraph replays the launch with the SAME by-value arguments): ctl[4] = replays so far, advanced by
// adamw_tick_kernel at the top of every replay and added to the captured step count; ctl[7] != 0: ctl[5] / ctl[6] are the learning
// rate / weight decay of this replay (written by the host before it, overriding the captured ones).  All stay 0 outside a graph.
__global__ void adamw_control_kernel(const float* __restrict__ norm, float max_norm, float skip_norm, float grad_scale,
                                     int guard, float* __restrict__ ctl) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    // guard: skip the step when the norm is inf / NaN (FlatAdamW.step(check_finite=...)).  The caller passes skip_norm = 0
    // together with a clip norm: native_scaler.py:24-32 is `if clip_grad ... elif skip_grad`.
    const float nrm = norm[0] * fabsf(grad_scale);
    const bool finite = (nrm == nrm) && (nrm <= 3.0e38f);
    const bool skip = (guard && !finite) || (skip_norm > 0.f && nrm >= skip_norm);
    float coef = 1.f;
    if (max_norm > 0.f) { coef = max_norm / (nrm + 1e-6f); coef = coef > 1.f ? 1.f : coef; }   // torch clip_grad_norm_
    ctl[0] = skip ? 0.f : grad_scale * coef;
    ctl[1] = skip ? 1.f : 0.f;
    ctl[2] = ctl[2] + (skip ? 1.f : 0.f);
    ctl[3] = nrm;
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, bf16* __restrict__ shadow, long n, float lr,
                                                    float b1, float b2, float eps, float wd, int step, float grad_scale,
                                                    const float* __restrict__ ctl) {
    if (ctl) {
        if (ctl[1] != 0.f) return;                      // skipped step: weights, moments and shadow stay as they are
        grad_scale = ctl[0];
        step += (int)ctl[4] - (int)ctl[2];
        if (ctl[7] != 0.f) { lr = ctl[5]; wd = ctl[6]; }
    }
    const float inv_bc1 = 1.f / (1.f - powf(b1, (float)step));
    const float inv_sqrt_bc2 = 1.f / sqrtf(1.f - powf(b2, (float)step));
    // one pass (block b owns elements [1024 b, 1024 b + 1024)), streaming accesses: the five buffers are touched once per step
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    f32x4 P = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + i));
    f32x4 G = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g + i)) * grad_scale;
    f32x4 M = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(m + i));
    f32x4 V = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(v + i));
    P *= (1.f - lr * wd);
    M = M * b1 + G * (1.f - b1);
    V = V * b2 + G * G * (1.f - b2);
#pragma unroll
    for (int j = 0; j < 4; ++j) P[j] -= (lr * inv_bc1) * M[j] / (sqrtf(V[j]) * inv_sqrt_bc2 + eps);
    __builtin_nontemporal_store(P, reinterpret_cast<f32x4*>(p + i));
    __builtin_nontemporal_store(M, reinterpret_cast<f32x4*>(m + i));
    __builtin_nontemporal_store(V, reinterpret_cast<f32x4*>(v + i));
    if (shadow) {
        bf16x4 s; s[0] = (bf16)P[0]; s[1] = (bf16)P[1]; s[2] = (bf16)P[2]; s[3] = (bf16)P[3];
        *reinterpret_cast<bf16x4*>(shadow + i) = s;
    }
}

// fp32 -> bf16 copy of a flat buffer (initial shadow / after loading a checkpoint)
__global__ __launch_bounds__(256) void shadow_kernel(const float* __restrict__ p, bf16* __restrict__ shadow, long n) {
    const long stride = (long)gridDim.x * 256 * 4;
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
        const f32x4 P = *reinterpret_cast<const f32x4*>(p + i);
        bf16x4 s; s[0] = (bf16)P[0]; s[1] = (bf16)P[1]; s[2] = (bf16)P[2]; s[3] = (bf16)P[3];
        *reinterpret_cast<bf16x4*>(shadow + i) = s;
    }
}

// Transposed bf16 shadows of the 2-D weights, all in one launch: the data-gradient GEMMs run dx = g @ W as
// linear(g, W^T) (both operands contraction-contiguous), and W changes once per optimizer step.  One 64x64 tile per block;
// `tiles` rows = {src element offset of the tile, dst element offset, ld_src, ld_dst, rows, cols} (rows/cols % 8 == 0).
struct TrTile { long long src, dst; int ld_src, ld_dst, nr, nc; };
static_assert(sizeof(TrTile) == 32, "TrTile layout is part of the C ABI (mmae_transpose_bf16_batched)");

__global__ __launch_bounds__(256) void transpose_bf16_batched_kernel(const bf16* __restrict__ src, bf16* __restrict__ dst,
                                                                      const TrTile* __restrict__ tiles) {
    __shared__ bf16 lds[64][72];                       // +8 pad: 144-byte rows keep the 16-byte row writes aligned
    const TrTile t = tiles[blockIdx.x];
    const int v = (threadIdx.x & 7) * 8, r = threadIdx.x >> 3;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int row = r + 32 * pass;
        if (row < t.nr && v < t.nc)
            *reinterpret_cast<uint4*>(&lds[row][v]) = *reinterpret_cast<const uint4*>(src + t.src + (long)row * t.ld_src + v);
    }
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int col = r + 32 * pass;                 // source column = destination row
        if (col < t.nc && v < t.nr) {
            union { uint4 q; bf16 e[8]; } o;
#pragma unroll
            for (int i = 0; i < 8; ++i) o.e[i] = lds[v + i][col];
            *reinterpret_cast<uint4*>(dst + t.dst + (long)col * t.ld_dst + v) = o.q;
        }
    }
}

// Zero-PADDED (and optionally transposed) bf16 copies of weight blocks whose sizes do not fit the own GEMM's tiles -- ViT-L's GEGLU width
// ffi = int(1024 * 8 / 3) = 2730 (reference: MM/multimae_crossattn.py:584-599 -> DSI-MM/zorro_utils.py:121-128) is no multiple of the
// 64-deep K-tile nor of the 256-wide N-tile: the engine keeps copies padded to 2816 (pads are zero, i.e. mathematically inert) and refreshes
// them after every update with ONE launch.  One tile of up to 64 x 64 source elements per table row (48 bytes):
//   {src address, dst address (bytes, absolute), ld_src, ld_dst (elements), rows, cols (<= 64, ANY value), transpose (0 / 1)}
// dst[r][c] = src[r][c] (transpose 0) or dst[c][r] = src[r][c] (transpose 1) for r < rows, c < cols; nothing else is written, so the pad
// region of a destination that was zero-filled once stays zero.  Edges are element-granular (2730 % 8 = 2); full, 16-byte aligned chunks
// move as vectors.
struct PadTile { unsigned long long src, dst; int ld_src, ld_dst, nr, nc, tr, pad_; long long pad2_; };
static_assert(sizeof(PadTile) == 48, "PadTile layout is part of the C ABI (mmae_pad_copy_bf16_batched)");

__global__ __launch_bounds__(256) void pad_copy_bf16_batched_kernel(const PadTile* __restrict__ tiles) {
    __shared__ bf16 lds[64][72];
    const PadTile t = tiles[blockIdx.x];
    const bf16* src = reinterpret_cast<const bf16*>(t.src);
    bf16* dst = reinterpret_cast<bf16*>(t.dst);
    const int v = (threadIdx.x & 7) * 8, r = threadIdx.x >> 3;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int row = r + 32 * pass;
        if (row < t.nr && v < t.nc) {
            const bf16* sp = src + (long)row * t.ld_src + v;
            if (v + 8 <= t.nc && (reinterpret_cast<uintptr_t>(sp) & 15) == 0) {
                *reinterpret_cast<uint4*>(&lds[row][v]) = *reinterpret_cast<const uint4*>(sp);
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) lds[row][v + i] = (v + i < t.nc) ? sp[i] : (bf16)0.f;
            }
        }
    }
    __syncthreads();
    const int n_along = t.tr ? t.nr : t.nc, n_across = t.tr ? t.nc : t.nr;       // destination: `across` rows of `along` elements
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int drow = r + 32 * pass;
        if (drow < n_across && v < n_along) {
            union { uint4 q; bf16 e[8]; } o;
#pragma unroll
            for (int i = 0; i < 8; ++i) o.e[i] = t.tr ? lds[(v + i) & 63][drow] : lds[drow][(v + i) & 63];
            bf16* dp = dst + (long)drow * t.ld_dst + v;
            if (v + 8 <= n_along && (reinterpret_cast<uintptr_t>(dp) & 15) == 0) {
                *reinterpret_cast<uint4*>(dp) = o.q;
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (v + i < n_along) dp[i] = o.e[i];
            }
        }
    }
}

// out[i] = sum_s part[s*n + i]: fp32 reduction of the S bf16 partial products of a split-K weight-gradient GEMM, written
// straight to its fp32 destination (the flat gradient buffer).  8 elements per lane, fixed summation order.
__global__ __launch_bounds__(256) void splitk_sum_kernel(const bf16* __restrict__ part, int S, long n, float* __restrict__ out) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i >= n) return;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
    for (int s = 0; s < S; ++s) {
        union { uint4 q; bf16 e[8]; } v;
        v.q = *reinterpret_cast<const uint4*>(part + (long)s * n + i);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += (float)v.e[j];
    }
    f32x4 a, b;
    a[0] = acc[0]; a[1] = acc[1]; a[2] = acc[2]; a[3] = acc[3];
    b[0] = acc[4]; b[1] = acc[5]; b[2] = acc[6]; b[3] = acc[7];
    *reinterpret_cast<f32x4*>(out + i) = a;
    *reinterpret_cast<f32x4*>(out + i + 4) = b;
}

// The same for up to 16 weight gradients in ONE launch (a layer's worth: the per-gradient launches were ~140 dependent 6-us kernels per
// step): the entries travel by value in the kernel argument (no pointer table in memory, no host-to-device copy); block b serves the
// entry e with first_block[e] <= b < first_block[e + 1].  Same summation order as splitk_sum_kernel: bitwise identical results.
struct SplitkMulti {
    const bf16* part[16]; float* out[16]; long n[16]; int S[16]; int first_block[17]; int count;
};
__global__ __launch_bounds__(256) void splitk_sum_multi_kernel(SplitkMulti m) {
    int e = 0;
#pragma unroll
    for (int k = 1; k < 16; ++k) e += (k < m.count && (int)blockIdx.x >= m.first_block[k]) ? 1 : 0;
    const long n = m.n[e];
    const long i = ((long)(blockIdx.x - m.first_block[e]) * 256 + threadIdx.x) * 8;
    if (i >= n) return;
    const bf16* part = m.part[e];
    const int S = m.S[e];
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
    for (int s = 0; s < S; ++s) {
        union { uint4 q; bf16 e[8]; } v;
        v.q = *reinterpret_cast<const uint4*>(part + (long)s * n + i);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += (float)v.e[j];
    }
    f32x4 a, b;
    a[0] = acc[0]; a[1] = acc[1]; a[2] = acc[2]; a[3] = acc[3];
    b[0] = acc[4]; b[1] = acc[5]; b[2] = acc[6]; b[3] = acc[7];
    float* out = m.out[e];
    *reinterpret_cast<f32x4*>(out + i) = a;
    *reinterpret_cast<f32x4*>(out + i + 4) = b;
}

// sum of squares of a flat fp32 buffer -> out[0] (+= when accumulate): two-stage, deterministic
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, long n, float* __restrict__ part) {
    __shared__ float red[4];
    float a = 0.f;
    const long stride = (long)gridDim.x * 256 * 4;
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
        const f32x4 G = *reinterpret_cast<const f32x4*>(g + i);
        a += G[0] * G[0] + G[1] * G[1] + G[2] * G[2] + G[3] * G[3];
    }
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void sumsq_finish_kernel(const float* __restrict__ part, int nb, float* out) {
    __shared__ float red[256];
    float a = 0.f;
    for (int i = threadIdx.x; i < nb; i += 256) a += part[i];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) out[0] = sqrtf(red[0]);
}

static int grid_for(long n) { long b = (n / 4 + 255) / 256; if (b > 2048) b = 2048; if (b < 1) b = 1; return (int)b; }

static int adamw_launch(long n, float* p, const float* g, float* m, float* v, void* shadow_bf16, float lr, float beta1,
                        float beta2, float eps, float weight_decay, int step, float grad_scale, const float* ctl,
                        void* stream) {
    if (n < 0 || (n % 4) || !p || !g || !m || !v || step < 1) return MMAE_ERR_ARG;
    if (n == 0) return MMAE_OK;
    const long nb = (n / 4 + 255) / 256;
    if (nb > 0x7fffffffL) return MMAE_ERR_ARG;
    MMAE_LAUNCH(adamw_kernel, dim3((unsigned)nb), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p, g, m, v,
                       reinterpret_cast<bf16*>(shadow_bf16), n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, ctl);
    MMAE_CHECK_LAUNCH();
    return MMAE_OK;
}

extern "C" int mmae_adamw_step(long n, float* p, const float* g, float* m, float* v, void* shadow_bf16, float lr,
                               float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                               void* stream) {
    return adamw_launch(n, p, g, m, v, shadow_bf16, lr, beta1, beta2, eps, weight_decay, step, grad_scale, nullptr, stream);
}

extern "C" int mmae_adamw_control(const float* grad_norm, float max_norm, float skip_norm, float grad_scale,
                                  int check_finite, float* ctl8, void* stream) {
    if (!grad_norm || !ctl8 || max_norm < 0.f || skip_norm < 0.f) return MMAE_ERR_ARG;
    MMAE_LAUNCH(adamw_control_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), grad_norm, max_norm,
                       skip_norm, grad_scale, check_finite, ctl8);
    MMAE_CHECK_LAUNCH();
    return MMAE_OK;
}

__global__ void adamw_tick_kernel(float* __restrict__ ctl) {
    if (threadIdx.x == 0 && blockIdx.x == 0) ctl[4] += 1.f;
}

extern "C" int mmae_adamw_tick(float* ctl8, void* stream) {
    if (!ctl8) return MMAE_ERR_ARG;
    MMAE_LAUNCH(adamw_tick_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), ctl8);
    MMAE_CHECK_LAUNCH();
    return MMAE_OK;
}

extern "C" int mmae_adamw_step_ctl(long n, float* p, const float* g, float* m, float* v, void* shadow_bf16, float lr,
                                   float beta1, float beta2, float eps, float weight_decay, int step, const float* ctl8,
                                   void* stream) {
    if (!ctl8) return MMAE_ERR_ARG;
    return adamw_launch(n, p, g, m, v, shadow_bf16, lr, beta1, beta2, eps, weight_decay, step, 1.f, ctl8, stream);
}

extern "C" int mmae_shadow_bf16(long n, const float* p, void* shadow_bf16, void* stream) {
    if (n < 0 || (n % 4) || !p || !shadow_bf16) return MMAE_ERR_ARG;
    if (n == 0) return MMAE_OK;
    MMAE_LAUNCH(shadow_kernel, dim3(grid_for(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p,
                       reinterpret_cast<bf16*>(shadow_bf16), n);
    MMAE_CHECK_LAUNCH();
    return MMAE_OK;
}

extern "C" int mmae_transpose_bf16_batched(const void* src_bf16, void* dst_bf16, const void* tiles, int n_tiles, void* stream) {
    if (n_tiles < 0 || !src_bf16 || !dst_bf16 || (n_tiles && !tiles)) return MMAE_ERR_ARG;
    if (n_tiles == 0) return MMAE_OK;
    MMAE_LAUNCH(transpose_bf16_batched_kernel, dim3(n_tiles), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const bf16*>(src_bf16), reinterpret_cast<bf16*>(dst_bf16),
                       reinterpret_cast<const TrTile*>(tiles));
    MMAE_CHECK_LAUNCH();
    return MMAE_OK;
}

extern "C" int mmae_pad_copy_bf16_batched(const void* tiles, int n_tiles, void* stream) {
    if (n_tiles < 0 || !tiles) return MMAE_ERR_ARG;
    if (n_tiles == 0) return MMAE_OK;
    MMAE_LAUNCH(pad_copy_bf16_batched_kernel, dim3(n_tiles), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const PadTile*>(tiles));
    MMAE_CHECK_LAUNCH();
    return MMAE_OK;
}

extern "C" int mmae_splitk_sum(int S, long n, const void* partials_bf16, float* out, void* stream) {
    if (S < 1 || n < 0 || (n % 8) || !partials_bf16 || !out) return MMAE_ERR_ARG;
    if (n == 0) return MMAE_OK;
    MMAE_LAUNCH(splitk_sum_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const bf16*>(partials_bf16), S, n, out);
    MMAE_CHECK_LAUNCH();
    return MMAE_OK;
}

extern "C" int mmae_splitk_sum_multi(int count, const void* const* partials_bf16, float* const* outs, const int* S, const long* n, void* stream) {
    if (count < 0 || !partials_bf16 || !outs || !S || !n) return MMAE_ERR_ARG;
    for (int i = 0; i < count; ++i)
        if (!partials_bf16[i] || !outs[i] || S[i] < 1 || n[i] <= 0 || (n[i] % 8)) return MMAE_ERR_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    for (int base = 0; base < count; base += 16) {
        SplitkMulti m{};
        m.count = count - base < 16 ? count - base : 16;
        int blocks = 0;
        for (int k = 0; k < m.count; ++k) {
            m.part[k] = reinterpret_cast<const bf16*>(partials_bf16[base + k]); m.out[k] = outs[base + k];
            m.n[k] = n[base + k]; m.S[k] = S[base + k]; m.first_block[k] = blocks;
            blocks += (int)((n[base + k] / 8 + 255) / 256);
        }
        m.first_block[m.count] = blocks;
        MMAE_LAUNCH(splitk_sum_multi_kernel, dim3(blocks), dim3(256), 0, st, m);
        MMAE_CHECK_LAUNCH();
    }
    return MMAE_OK;
}

extern "C" int mmae_grad_norm(long n, const float* g, float* partial_ws_2048, float* out_norm, void* stream) {
    if (n < 0 || (n % 4) || !g || !partial_ws_2048 || !out_norm) return MMAE_ERR_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int nb = grid_for(n);
    MMAE_LAUNCH(sumsq_partial_kernel, dim3(nb), dim3(256), 0, st, g, n, partial_ws_2048);
    MMAE_CHECK_LAUNCH();
    MMAE_LAUNCH(sumsq_finish_kernel, dim3(1), dim3(256), 0, st, partial_ws_2048, nb, out_norm);
    MMAE_CHECK_LAUNCH();
    return MMAE_OK;
}
