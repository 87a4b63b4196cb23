// HBM-bound row kernels of the fusion-token path (gfx950): fused residual-add + (double) LayerNorm, GEGLU, GELU,
// row gather / scatter.  One 64-lane wave owns one token row; all loads/stores are 16 B per lane and coalesced.
//
// Reference compositions replaced (downstream/instance_segmentation/modeling/multimae/zorro_utils.py):
//   Block.forward :238-239  x + attn(norm1(x)), x + mlp(norm2(x)) with Attention.norm (:176) / FeedForward[0] (:124)
//   applied again on top of norm1/norm2 -> LN(LN(x)): both LayerNorms and the preceding residual add are ONE pass here.
//   LayerNorm :103-110 (gamma, zero beta buffer, eps 1e-5); decoder nn.LayerNorm(eps=1e-6) with bias
//   (pretraining/multimae/output_adapters_simple.py:75).  GEGLU :115-118 (exact erf GELU, value = first half, gate = second).
#include "common.hpp"
#include "mmae_hip.h"

template <typename T> __device__ __forceinline__ f32x4 ld4f(const T* p);
template <> __device__ __forceinline__ f32x4 ld4f<float>(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
template <> __device__ __forceinline__ f32x4 ld4f<bf16>(const bf16* p) {
    const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
template <typename T> __device__ __forceinline__ void st4f(T* p, const f32x4& v);
template <> __device__ __forceinline__ void st4f<float>(float* p, const f32x4& v) { *reinterpret_cast<f32x4*>(p) = v; }
template <> __device__ __forceinline__ void st4f<bf16>(bf16* p, const f32x4& v) {
    bf16x4 o; o[0] = (bf16)v[0]; o[1] = (bf16)v[1]; o[2] = (bf16)v[2]; o[3] = (bf16)v[3];
    *reinterpret_cast<bf16x4*>(p) = o;
}

// streaming (nontemporal) forms for the token-row tensors: each is touched once per kernel and is 0.25-0.5 GB
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
template <typename T> __device__ __forceinline__ f32x4 ld4s(const T* p);
template <> __device__ __forceinline__ f32x4 ld4s<float>(const float* p) { return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p)); }
template <> __device__ __forceinline__ f32x4 ld4s<bf16>(const bf16* p) {
    union { u32x2 q; bf16 e[4]; } u;
    u.q = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(p));
    return f32x4{(float)u.e[0], (float)u.e[1], (float)u.e[2], (float)u.e[3]};
}
template <typename T> __device__ __forceinline__ void st4s(T* p, const f32x4& v);
template <> __device__ __forceinline__ void st4s<float>(float* p, const f32x4& v) { __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p)); }
template <> __device__ __forceinline__ void st4s<bf16>(bf16* p, const f32x4& v) {
    union { u32x2 q; bf16 e[4]; } u;
    u.e[0] = (bf16)v[0]; u.e[1] = (bf16)v[1]; u.e[2] = (bf16)v[2]; u.e[3] = (bf16)v[3];
    __builtin_nontemporal_store(u.q, reinterpret_cast<u32x2*>(p));
}

__device__ __forceinline__ float sum4(const f32x4& v) { return (v[0] + v[1]) + (v[2] + v[3]); }

// ------------------------------------------------------------------------------------------ add + LayerNorm(x2) forward
struct AddLnFwd {
    const float* x; const void* delta; float* x_new; void* y;
    const float* g1; const float* b1; const float* g2; const float* b2;
    float eps1, eps2; float* stats; long rows; int D;
    void* y2 = nullptr;          // fast path only: a bf16 copy of an fp32 y, written in the same pass (mmae_add_ln_fwd_cast)
};

// NC = number of 256-column chunks held per lane (4 columns per lane per chunk)
template <typename TD, typename TY, int NC, bool DOUBLE>
__global__ __launch_bounds__(256) void add_ln_fwd_kernel(AddLnFwd p) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.rows) return;
    const int D = p.D;
    const float invD = 1.f / (float)D;
    f32x4 v[NC];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int col = 4 * (lane + 64 * c);
        v[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (col < D) {
            v[c] = ld4s<float>(p.x + row * D + col);
            if (p.delta) v[c] += ld4s<TD>(reinterpret_cast<const TD*>(p.delta) + row * D + col);
            if (p.x_new) st4s<float>(p.x_new + row * D + col, v[c]);
            s += sum4(v[c]);
        }
    }
    const float m1 = wave_sum(s) * invD;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int col = 4 * (lane + 64 * c);
        if (col < D) { const f32x4 d = v[c] - m1; q += sum4(d * d); }
    }
    const float r1 = rsqrtf(wave_sum(q) * invD + p.eps1);
    float s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int col = 4 * (lane + 64 * c);
        if (col < D) {
            f32x4 u = (v[c] - m1) * r1 * ld4f<float>(p.g1 + col);
            if (p.b1) u += ld4f<float>(p.b1 + col);
            v[c] = u;
            s2 += sum4(u);
        }
    }
    float m2 = 0.f, r2 = 1.f;
    if (DOUBLE) {
        m2 = wave_sum(s2) * invD;
        float q2 = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int col = 4 * (lane + 64 * c);
            if (col < D) { const f32x4 d = v[c] - m2; q2 += sum4(d * d); }
        }
        r2 = rsqrtf(wave_sum(q2) * invD + p.eps2);
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int col = 4 * (lane + 64 * c);
        if (col < D) {
            f32x4 o = v[c];
            if (DOUBLE) {
                o = (o - m2) * r2 * ld4f<float>(p.g2 + col);
                if (p.b2) o += ld4f<float>(p.b2 + col);
            }
            st4s<TY>(reinterpret_cast<TY*>(p.y) + row * D + col, o);
        }
    }
    if (lane == 0) *reinterpret_cast<f32x4*>(p.stats + row * 4) = f32x4{m1, r1, m2, r2};
}

// ------------------------------------------------------------------------------------------ add + LayerNorm(x2) backward
struct AddLnBwd {
    const float* x_new; const void* gy; const float* gx_up;
    const float* g1; const float* b1; const float* g2; const float* stats;
    float* gx; void* gdelta; float* ws;   // ws: (nblk, 4, D) partial column sums [dg1, db1, dg2, db2]
    long rows; int D;
};

// HASB: some beta exists (decoder nn.LayerNorm) -> keep the dbeta accumulators; the encoder's bias-less LayerNorms
// instantiate HASB = false and save 36 VGPRs.  gamma/beta vectors live in LDS (staged once per block) instead of
// registers: 170 -> ~100 VGPRs, i.e. 4 resident waves per SIMD instead of 2 for this persistent grid-stride kernel.
template <typename TD, typename TY, int NC, bool DOUBLE, bool HASB>
__global__ __launch_bounds__(256) void add_ln_bwd_kernel(AddLnBwd p) {
    extern __shared__ __attribute__((aligned(16))) float red[];   // [4 waves][D] reduction scratch, then g1, g2, b1 [D] each
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int D = p.D;
    const float invD = 1.f / (float)D;
    float* sg1 = red + 4 * D; float* sg2 = sg1 + D; float* sb1 = sg2 + D;
    for (int c = threadIdx.x; c < D; c += 256) {
        sg1[c] = p.g1[c];
        sg2[c] = DOUBLE ? p.g2[c] : 0.f;
        sb1[c] = (HASB && p.b1) ? p.b1[c] : 0.f;
    }
    __syncthreads();
    f32x4 dg1[NC], dg2[DOUBLE ? NC : 1], db1[HASB ? NC : 1], db2[(HASB && DOUBLE) ? NC : 1];
    const f32x4 z4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        dg1[c] = z4;
        if (DOUBLE) dg2[c] = z4;
        if (HASB) db1[c] = z4;
        if (HASB && DOUBLE) db2[c] = z4;
    }
    for (long row = (long)blockIdx.x * 4 + wave; row < p.rows; row += (long)gridDim.x * 4) {
        const f32x4 st = *reinterpret_cast<const f32x4*>(p.stats + row * 4);
        const float m1 = st[0], r1 = st[1], m2 = st[2], r2 = st[3];
        f32x4 xh[NC], gv[NC];
        float a = 0.f, bsum = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int col = 4 * (lane + 64 * c);
            xh[c] = z4; gv[c] = z4;
            if (col < D) {
                xh[c] = (ld4s<float>(p.x_new + row * D + col) - m1) * r1;
                const f32x4 gy = ld4s<TY>(reinterpret_cast<const TY*>(p.gy) + row * D + col);
                if (DOUBLE) {
                    const f32x4 G1 = *reinterpret_cast<const f32x4*>(sg1 + col);
                    f32x4 u = xh[c] * G1;
                    if (HASB) u += *reinterpret_cast<const f32x4*>(sb1 + col);
                    const f32x4 uh = (u - m2) * r2;
                    dg2[c] += gy * uh;
                    if (HASB) db2[c] += gy;
                    const f32x4 guh = gy * *reinterpret_cast<const f32x4*>(sg2 + col);
                    gv[c] = guh;
                    a += sum4(guh); bsum += sum4(guh * uh);
                } else {
                    gv[c] = gy;
                }
            }
        }
        if (DOUBLE) {
            const float c1 = wave_sum(a) * invD, c2 = wave_sum(bsum) * invD;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int col = 4 * (lane + 64 * c);
                if (col < D) {
                    f32x4 u = xh[c] * *reinterpret_cast<const f32x4*>(sg1 + col);
                    if (HASB) u += *reinterpret_cast<const f32x4*>(sb1 + col);
                    const f32x4 uh = (u - m2) * r2;
                    gv[c] = (gv[c] - c1 - uh * c2) * r2;      // grad wrt u
                }
            }
        }
        float a3 = 0.f, a4 = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int col = 4 * (lane + 64 * c);
            if (col < D) {
                dg1[c] += gv[c] * xh[c];
                if (HASB) db1[c] += gv[c];
                gv[c] = gv[c] * *reinterpret_cast<const f32x4*>(sg1 + col);   // grad wrt xhat
                a3 += sum4(gv[c]); a4 += sum4(gv[c] * xh[c]);
            }
        }
        const float c3 = wave_sum(a3) * invD, c4 = wave_sum(a4) * invD;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int col = 4 * (lane + 64 * c);
            if (col < D) {
                f32x4 gx = (gv[c] - c3 - xh[c] * c4) * r1;
                if (p.gx_up) gx += ld4s<float>(p.gx_up + row * D + col);
                if (p.gx) st4s<float>(p.gx + row * D + col, gx);
                if (p.gdelta) st4s<TD>(reinterpret_cast<TD*>(p.gdelta) + row * D + col, gx);
            }
        }
    }
    // block reduction of the column-sum vectors, one after the other: [dg1, db1, dg2, db2]
#pragma unroll
    for (int qn = 0; qn < 4; ++qn) {
        if (!DOUBLE && qn >= 2) break;
        __syncthreads();
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int col = 4 * (lane + 64 * c);
            if (col < D) {
                f32x4 val = z4;
                if (qn == 0) val = dg1[c];
                else if (qn == 1) { if (HASB) val = db1[c]; }
                else if (qn == 2) { if (DOUBLE) val = dg2[c]; }
                else { if (HASB && DOUBLE) val = db2[c]; }
                *reinterpret_cast<f32x4*>(red + wave * D + col) = val;
            }
        }
        __syncthreads();
        for (int col = threadIdx.x; col < D; col += 256)
            p.ws[((long)blockIdx.x * 4 + qn) * D + col] = red[col] + red[D + col] + red[2 * D + col] + red[3 * D + col];
    }
}

// out_q[col] (+)= sum_b ws[b][q][col].  grid (ceil(D/64), nq); 1024 threads = 16 row groups x 64 columns (coalesced 256 B
// per group), 4 independent loads in flight per thread.
__global__ __launch_bounds__(1024) void colsum_finalize_kernel(const float* __restrict__ ws, int nblk, int D, float* o0,
                                                               float* o1, float* o2, float* o3, int accumulate) {
    __shared__ float red[1024];
    const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + c;
    const int qn = blockIdx.y;
    float* out = qn == 0 ? o0 : qn == 1 ? o1 : qn == 2 ? o2 : o3;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f, s5 = 0.f, s6 = 0.f, s7 = 0.f;
    if (col < D && out) {
        const float* base = ws + (long)qn * D + col;
        const long stride = 4L * D;
        int b = rg;
        for (; b + 112 < nblk; b += 128) {                    // eight independent loads in flight (L2 hits: latency, not bytes)
            s0 += base[(long)b * stride]; s1 += base[(long)(b + 16) * stride];
            s2 += base[(long)(b + 32) * stride]; s3 += base[(long)(b + 48) * stride];
            s4 += base[(long)(b + 64) * stride]; s5 += base[(long)(b + 80) * stride];
            s6 += base[(long)(b + 96) * stride]; s7 += base[(long)(b + 112) * stride];
        }
        for (; b < nblk; b += 16) s0 += base[(long)b * stride];
    }
    red[threadIdx.x] = ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7));
    __syncthreads();
    if (rg == 0 && col < D && out) {
        float s = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) s += red[g * 64 + c];
        out[col] = accumulate ? out[col] + s : s;
    }
}

template <typename TD, typename TY, bool DOUBLE>
static int add_ln_fwd_nc(const AddLnFwd& p, hipStream_t st) {
    const int nc = cdiv(p.D, 256);
    dim3 grid(cdiv(p.rows, 4)), blk(256);
    switch (nc) {
        case 1: MMAE_LAUNCH((add_ln_fwd_kernel<TD, TY, 1, DOUBLE>), grid, blk, 0, st, p); break;
        case 2: MMAE_LAUNCH((add_ln_fwd_kernel<TD, TY, 2, DOUBLE>), grid, blk, 0, st, p); break;
        case 3: MMAE_LAUNCH((add_ln_fwd_kernel<TD, TY, 3, DOUBLE>), grid, blk, 0, st, p); break;
        case 4: MMAE_LAUNCH((add_ln_fwd_kernel<TD, TY, 4, DOUBLE>), grid, blk, 0, st, p); break;
        default: return MMAE_ERR_ARG;
    }
    MMAE_CHECK_LAUNCH();
    return MMAE_OK;
}
template <typename TD, typename TY, bool DOUBLE>
static int add_ln_bwd_nc(const AddLnBwd& p, int nblk, bool hasb, hipStream_t st) {
    const int nc = cdiv(p.D, 256);
    dim3 grid(nblk), blk(256);
    const size_t lds = (size_t)7 * p.D * sizeof(float);
#define GO(NCV)                                                                                           \
    if (hasb) MMAE_LAUNCH((add_ln_bwd_kernel<TD, TY, NCV, DOUBLE, true>), grid, blk, lds, st, p);  \
    else MMAE_LAUNCH((add_ln_bwd_kernel<TD, TY, NCV, DOUBLE, false>), grid, blk, lds, st, p);
    switch (nc) {
        case 1: GO(1) break;
        case 2: GO(2) break;
        case 3: GO(3) break;
        case 4: GO(4) break;
        default: return MMAE_ERR_ARG;
    }
#undef GO
    MMAE_CHECK_LAUNCH();
    return MMAE_OK;
}

// ------------------------------------------------------------------------------------------ dual double-LayerNorm
// The modality rows of a layer are normalised twice from the same residual value: once with Block_Fusion's (norm1,
// attn.norm) for the modality attention's K/V and once with Block's (norm1, attn.norm) for the Zorro attention
// (multimae_crossattn.py:454-470 with DSI-MM/zorro_utils.py:238, :255).  Both first LayerNorms share mean/rstd; one pass
// reads x (+ delta) once and writes both normalised matrices; the backward takes both upstream gradients and touches
// x_new / gx_up / gx once (18 B per element instead of 30).  Bias-less (encoder) LayerNorms only.
struct AddLnFwdDual {
    const float* x; const void* delta; float* x_new; void* ya; void* yb;
    const float* g1a; const float* g2a; const float* g1b; const float* g2b;
    float eps1, eps2; float* stats_a; float* stats_b; long rows; int D;
};
template <typename TD, typename TY, int NC>
__global__ __launch_bounds__(256) void add_ln_fwd_dual_kernel(AddLnFwdDual p) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.rows) return;
    const int D = p.D;
    const float invD = 1.f / (float)D;
    f32x4 v[NC];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int col = 4 * (lane + 64 * c);
        v[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (col < D) {
            v[c] = ld4s<float>(p.x + row * D + col);
            if (p.delta) v[c] += ld4s<TD>(reinterpret_cast<const TD*>(p.delta) + row * D + col);
            if (p.x_new) st4s<float>(p.x_new + row * D + col, v[c]);
            s += sum4(v[c]);
        }
    }
    const float m1 = wave_sum(s) * invD;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int col = 4 * (lane + 64 * c);
        if (col < D) { v[c] = v[c] - m1; q += sum4(v[c] * v[c]); }
    }
    const float r1 = rsqrtf(wave_sum(q) * invD + p.eps1);
#pragma unroll
    for (int c = 0; c < NC; ++c) v[c] = v[c] * r1;                        // xhat, shared by both paths
#pragma unroll
    for (int path = 0; path < 2; ++path) {
        const float* g1 = path ? p.g1b : p.g1a; const float* g2 = path ? p.g2b : p.g2a;
        TY* y = reinterpret_cast<TY*>(path ? p.yb : p.ya);
        f32x4 u[NC];
        float s2 = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int col = 4 * (lane + 64 * c);
            u[c] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (col < D) { u[c] = v[c] * ld4f<float>(g1 + col); s2 += sum4(u[c]); }
        }
        const float m2 = wave_sum(s2) * invD;
        float q2 = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int col = 4 * (lane + 64 * c);
            if (col < D) { const f32x4 d = u[c] - m2; q2 += sum4(d * d); }
        }
        const float r2 = rsqrtf(wave_sum(q2) * invD + p.eps2);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int col = 4 * (lane + 64 * c);
            if (col < D) st4s<TY>(y + row * D + col, (u[c] - m2) * r2 * ld4f<float>(g2 + col));
        }
        if (lane == 0) *reinterpret_cast<f32x4*>((path ? p.stats_b : p.stats_a) + row * 4) = f32x4{m1, r1, m2, r2};
    }
}

struct AddLnBwdDual {
    const float* x_new; const void* gya; const void* gyb; const float* gx_up;
    const float* g1a; const float* g2a; const float* g1b; const float* g2b;
    const float* stats_a; const float* stats_b;
    float* gx; void* gdelta; float* ws;   // ws: (nblk, 4, D) partial column sums [dg1a, dg2a, dg1b, dg2b]
    long rows; int D;
};
template <typename TD, typename TY, int NC>
__global__ __launch_bounds__(256) void add_ln_bwd_dual_kernel(AddLnBwdDual p) {
    extern __shared__ __attribute__((aligned(16))) float red[];   // [4 waves][D] reduction scratch, then g1a, g2a, g1b, g2b [D] each
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int D = p.D;
    const float invD = 1.f / (float)D;
    float* sg = red + 4 * D;                                      // sg + (2 * path + k) * D
    for (int c = threadIdx.x; c < D; c += 256) {
        sg[c] = p.g1a[c]; sg[D + c] = p.g2a[c]; sg[2 * D + c] = p.g1b[c]; sg[3 * D + c] = p.g2b[c];
    }
    __syncthreads();
    const f32x4 z4{0.f, 0.f, 0.f, 0.f};
    f32x4 dg[4][NC];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int c = 0; c < NC; ++c) dg[k][c] = z4;
    for (long row = (long)blockIdx.x * 4 + wave; row < p.rows; row += (long)gridDim.x * 4) {
        const f32x4 sta = *reinterpret_cast<const f32x4*>(p.stats_a + row * 4);
        const f32x4 stb = *reinterpret_cast<const f32x4*>(p.stats_b + row * 4);
        const float m1 = sta[0], r1 = sta[1];
        f32x4 xh[NC], tot[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int col = 4 * (lane + 64 * c);
            xh[c] = z4; tot[c] = z4;
            if (col < D) xh[c] = (ld4s<float>(p.x_new + row * D + col) - m1) * r1;
        }
#pragma unroll
        for (int path = 0; path < 2; ++path) {
            const float m2 = path ? stb[2] : sta[2], r2 = path ? stb[3] : sta[3];
            const float* G1 = sg + 2 * path * D; const float* G2 = G1 + D;
            const TY* gyp = reinterpret_cast<const TY*>(path ? p.gyb : p.gya);
            f32x4 gv[NC];
            float a = 0.f, bsum = 0.f;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int col = 4 * (lane + 64 * c);
                gv[c] = z4;
                if (col < D) {
                    const f32x4 gy = ld4s<TY>(gyp + row * D + col);
                    const f32x4 uh = (xh[c] * *reinterpret_cast<const f32x4*>(G1 + col) - m2) * r2;
                    dg[2 * path + 1][c] += gy * uh;
                    const f32x4 guh = gy * *reinterpret_cast<const f32x4*>(G2 + col);
                    gv[c] = guh;
                    a += sum4(guh); bsum += sum4(guh * uh);
                }
            }
            const float c1 = wave_sum(a) * invD, c2 = wave_sum(bsum) * invD;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int col = 4 * (lane + 64 * c);
                if (col < D) {
                    const f32x4 g1v = *reinterpret_cast<const f32x4*>(G1 + col);
                    const f32x4 uh = (xh[c] * g1v - m2) * r2;
                    const f32x4 gu = (gv[c] - c1 - uh * c2) * r2;             // grad wrt u = xhat * g1
                    dg[2 * path][c] += gu * xh[c];
                    tot[c] += gu * g1v;                                        // grad wrt xhat, both paths summed
                }
            }
        }
        float a3 = 0.f, a4 = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) { a3 += sum4(tot[c]); a4 += sum4(tot[c] * xh[c]); }
        const float c3 = wave_sum(a3) * invD, c4 = wave_sum(a4) * invD;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int col = 4 * (lane + 64 * c);
            if (col < D) {
                f32x4 gx = (tot[c] - c3 - xh[c] * c4) * r1;
                if (p.gx_up) gx += ld4s<float>(p.gx_up + row * D + col);
                if (p.gx) st4s<float>(p.gx + row * D + col, gx);
                if (p.gdelta) st4s<TD>(reinterpret_cast<TD*>(p.gdelta) + row * D + col, gx);
            }
        }
    }
#pragma unroll
    for (int qn = 0; qn < 4; ++qn) {
        __syncthreads();
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int col = 4 * (lane + 64 * c);
            if (col < D) *reinterpret_cast<f32x4*>(red + wave * D + col) = dg[qn][c];
        }
        __syncthreads();
        for (int col = threadIdx.x; col < D; col += 256)
            p.ws[((long)blockIdx.x * 4 + qn) * D + col] = red[col] + red[D + col] + red[2 * D + col] + red[3 * D + col];
    }
}


// ------------------------------------------------------------------------------------------ fast paths (D == 256 * NC, no beta)
// The generic kernels above guard every 256-column chunk with `col < D` and every optional operand with a pointer test.  In
// the generated code each guard is a branch of its own with an s_waitcnt vmcnt(0) behind it: the three chunks of a 768-wide
// row were fetched one after the other, 6-12 dependent trips to memory per row, and the row reductions went through the LDS
// crossbar (ds_bpermute).  The kernels below are the same arithmetic for full rows with the optional operands as template
// flags: one basic block per row, every load of the row in flight at once, reductions on DPP + permlane swaps (VALU only).
// raw (unconverted) 4-element row pieces: the loads of a row are issued as one group and converted where they are used
template <typename T> struct Raw4;
template <> struct Raw4<float> { typedef f32x4 type; };
template <> struct Raw4<bf16> { typedef u32x2 type; };
template <typename T> __device__ __forceinline__ typename Raw4<T>::type ld4raw(const T* p);
template <> __device__ __forceinline__ f32x4 ld4raw<float>(const float* p) { return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p)); }
template <> __device__ __forceinline__ u32x2 ld4raw<bf16>(const bf16* p) { return __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(p)); }
__device__ __forceinline__ f32x4 cvt4(const f32x4& r) { return r; }
__device__ __forceinline__ f32x4 cvt4(const u32x2& r) {
    return f32x4{__builtin_bit_cast(float, r[0] << 16), __builtin_bit_cast(float, r[0] & 0xffff0000u),
                 __builtin_bit_cast(float, r[1] << 16), __builtin_bit_cast(float, r[1] & 0xffff0000u)};
}
template <int CTRL> __device__ __forceinline__ float dpp_get(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_sum_v(float v) {
    v += dpp_get<0xB1>(v);                 // quad_perm [1,0,3,2]
    v += dpp_get<0x4E>(v);                 // quad_perm [2,3,0,1]
    v += dpp_get<0x124>(v);                // row_ror:4
    v += dpp_get<0x128>(v);                // row_ror:8  -> every lane: sum of its 16-lane row
    const unsigned u = __builtin_bit_cast(unsigned, v);
    auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    v = __builtin_bit_cast(float, (unsigned)a[0]) + __builtin_bit_cast(float, (unsigned)a[1]);
    const unsigned w = __builtin_bit_cast(unsigned, v);
    auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    return __builtin_bit_cast(float, (unsigned)b[0]) + __builtin_bit_cast(float, (unsigned)b[1]);
}

// EXPERIMENT knob: the normalised output is the next GEMM's operand -- nontemporal (streaming) or regular stores?
#ifndef MMAE_Y_NT
#define MMAE_Y_NT 1
#endif
#if MMAE_Y_NT
#define STY st4s
#else
#define STY st4f
#endif
template <typename TD, typename TY, int NC, bool DOUBLE, bool HAS_DELTA, bool Y2 = false>
__global__ __launch_bounds__(256) void add_ln_fwd_fast_kernel(AddLnFwd p) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (row >= p.rows) return;
    constexpr int D = 256 * NC;
    constexpr float invD = 1.f / (float)D;
    const long at = row * D + 4 * lane;
    f32x4 v[NC], dl[NC], G1[NC], G2[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) v[c] = ld4s<float>(p.x + at + 256 * c);
    if (HAS_DELTA)
#pragma unroll
        for (int c = 0; c < NC; ++c) dl[c] = ld4s<TD>(reinterpret_cast<const TD*>(p.delta) + at + 256 * c);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        G1[c] = ld4f<float>(p.g1 + 4 * lane + 256 * c);
        if (DOUBLE) G2[c] = ld4f<float>(p.g2 + 4 * lane + 256 * c);
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        if (HAS_DELTA) { v[c] += dl[c]; st4s<float>(p.x_new + at + 256 * c, v[c]); }
        s += sum4(v[c]);
    }
    const float m1 = wave_sum_v(s) * invD;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) { v[c] = v[c] - m1; q += sum4(v[c] * v[c]); }
    const float r1 = rsqrtf(wave_sum_v(q) * invD + p.eps1);
    float s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) { v[c] = v[c] * r1 * G1[c]; s2 += sum4(v[c]); }
    float m2 = 0.f, r2 = 1.f;
    if (DOUBLE) {
        m2 = wave_sum_v(s2) * invD;
        float q2 = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) { v[c] = v[c] - m2; q2 += sum4(v[c] * v[c]); }
        r2 = rsqrtf(wave_sum_v(q2) * invD + p.eps2);
#pragma unroll
        for (int c = 0; c < NC; ++c) v[c] = v[c] * r2 * G2[c];
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        STY<TY>(reinterpret_cast<TY*>(p.y) + at + 256 * c, v[c]);
        if (Y2) st4s<bf16>(reinterpret_cast<bf16*>(p.y2) + at + 256 * c, v[c]);
    }
    if (lane == 0) *reinterpret_cast<f32x4*>(p.stats + row * 4) = f32x4{m1, r1, m2, r2};
}

// persistent: 4 rows per block in flight, gamma vectors in LDS, column sums [dg1, -, dg2, -] in registers
template <typename TD, typename TY, int NC, bool DOUBLE, bool UP, bool GX, bool GD>
__global__ __launch_bounds__(256) void add_ln_bwd_fast_kernel(AddLnBwd p) {
    constexpr int D = 256 * NC;
    constexpr float invD = 1.f / (float)D;
    __shared__ __attribute__((aligned(16))) float red[6 * D];     // [4 waves][D] reduction scratch, then g1, g2
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* sg1 = red + 4 * D; float* sg2 = sg1 + D;
    for (int c = threadIdx.x; c < D; c += 256) { sg1[c] = p.g1[c]; sg2[c] = DOUBLE ? p.g2[c] : 0.f; }
    __syncthreads();
    const f32x4 z4{0.f, 0.f, 0.f, 0.f};
    f32x4 dg1[NC], dg2[DOUBLE ? NC : 1];
#pragma unroll
    for (int c = 0; c < NC; ++c) { dg1[c] = z4; if (DOUBLE) dg2[c] = z4; }
    for (long row = (long)blockIdx.x * 4 + wave; row < p.rows; row += (long)gridDim.x * 4) {
        const long at = row * D + 4 * lane;
        int z = 0;                                                    // opaque zero: the gamma reads below stay LDS reads of this
        asm volatile("" : "+s"(z));                                   // iteration (hoisted out of the loop they cost 24 VGPRs)
        const float* g1l = sg1 + z + 4 * lane; const float* g2l = sg2 + z + 4 * lane;
        f32x4 xh[NC], gv[NC], up[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) xh[c] = ld4s<float>(p.x_new + at + 256 * c);
#pragma unroll
        for (int c = 0; c < NC; ++c) gv[c] = ld4s<TY>(reinterpret_cast<const TY*>(p.gy) + at + 256 * c);
        if (UP)
#pragma unroll
            for (int c = 0; c < NC; ++c) up[c] = ld4s<float>(p.gx_up + at + 256 * c);
        const f32x4 st = *reinterpret_cast<const f32x4*>(p.stats + row * 4);
        const float m1 = st[0], r1 = st[1], m2 = st[2], r2 = st[3];
#pragma unroll
        for (int c = 0; c < NC; ++c) xh[c] = (xh[c] - m1) * r1;
        if (DOUBLE) {
            f32x4 uh[NC];
            float a = 0.f, bsum = 0.f;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                uh[c] = (xh[c] * *reinterpret_cast<const f32x4*>(g1l + 256 * c) - m2) * r2;
                dg2[c] += gv[c] * uh[c];
                gv[c] = gv[c] * *reinterpret_cast<const f32x4*>(g2l + 256 * c);
                a += sum4(gv[c]); bsum += sum4(gv[c] * uh[c]);
            }
            const float c1 = wave_sum_v(a) * invD, c2 = wave_sum_v(bsum) * invD;
#pragma unroll
            for (int c = 0; c < NC; ++c) gv[c] = (gv[c] - c1 - uh[c] * c2) * r2;      // grad wrt u = xhat * g1
        }
        float a3 = 0.f, a4 = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            dg1[c] += gv[c] * xh[c];
            gv[c] = gv[c] * *reinterpret_cast<const f32x4*>(g1l + 256 * c);   // grad wrt xhat
            a3 += sum4(gv[c]); a4 += sum4(gv[c] * xh[c]);
        }
        const float c3 = wave_sum_v(a3) * invD, c4 = wave_sum_v(a4) * invD;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            f32x4 gx = (gv[c] - c3 - xh[c] * c4) * r1;
            if (UP) gx += up[c];
            if (GX) st4s<float>(p.gx + at + 256 * c, gx);
            if (GD) st4s<TD>(reinterpret_cast<TD*>(p.gdelta) + at + 256 * c, gx);
        }
    }
#pragma unroll
    for (int qn = 0; qn < 4; qn += 2) {                               // [dg1, (dbeta1), dg2, (dbeta2)]: no beta on this path
        if (!DOUBLE && qn >= 2) break;
        __syncthreads();
#pragma unroll
        for (int c = 0; c < NC; ++c)
            *reinterpret_cast<f32x4*>(red + wave * D + 4 * lane + 256 * c) = qn == 0 ? dg1[c] : dg2[DOUBLE ? c : 0];
        __syncthreads();
        for (int col = threadIdx.x; col < D; col += 256)
            p.ws[((long)blockIdx.x * 4 + qn) * D + col] = red[col] + red[D + col] + red[2 * D + col] + red[3 * D + col];
    }
}

template <typename TD, typename TY, int NC, bool HAS_DELTA>
__global__ __launch_bounds__(256) void add_ln_fwd_dual_fast_kernel(AddLnFwdDual p) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (row >= p.rows) return;
    constexpr int D = 256 * NC;
    constexpr float invD = 1.f / (float)D;
    const long at = row * D + 4 * lane;
    f32x4 v[NC], dl[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) v[c] = ld4s<float>(p.x + at + 256 * c);
    if (HAS_DELTA)
#pragma unroll
        for (int c = 0; c < NC; ++c) dl[c] = ld4s<TD>(reinterpret_cast<const TD*>(p.delta) + at + 256 * c);
    f32x4 G1[2][NC], G2[2][NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        G1[0][c] = ld4f<float>(p.g1a + 4 * lane + 256 * c); G2[0][c] = ld4f<float>(p.g2a + 4 * lane + 256 * c);
        G1[1][c] = ld4f<float>(p.g1b + 4 * lane + 256 * c); G2[1][c] = ld4f<float>(p.g2b + 4 * lane + 256 * c);
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        if (HAS_DELTA) { v[c] += dl[c]; st4s<float>(p.x_new + at + 256 * c, v[c]); }
        s += sum4(v[c]);
    }
    const float m1 = wave_sum_v(s) * invD;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) { v[c] = v[c] - m1; q += sum4(v[c] * v[c]); }
    const float r1 = rsqrtf(wave_sum_v(q) * invD + p.eps1);
    // both second LayerNorms side by side: their reductions are independent, so the four wave sums overlap pairwise
    f32x4 u[2][NC];
    float s2[2] = {0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        v[c] = v[c] * r1;                                              // xhat, shared by both paths
#pragma unroll
        for (int path = 0; path < 2; ++path) { u[path][c] = v[c] * G1[path][c]; s2[path] += sum4(u[path][c]); }
    }
    const float m2a = wave_sum_v(s2[0]) * invD, m2b = wave_sum_v(s2[1]) * invD;
    float q2[2] = {0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        u[0][c] = u[0][c] - m2a; q2[0] += sum4(u[0][c] * u[0][c]);
        u[1][c] = u[1][c] - m2b; q2[1] += sum4(u[1][c] * u[1][c]);
    }
    const float r2a = rsqrtf(wave_sum_v(q2[0]) * invD + p.eps2), r2b = rsqrtf(wave_sum_v(q2[1]) * invD + p.eps2);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        STY<TY>(reinterpret_cast<TY*>(p.ya) + at + 256 * c, u[0][c] * r2a * G2[0][c]);
        STY<TY>(reinterpret_cast<TY*>(p.yb) + at + 256 * c, u[1][c] * r2b * G2[1][c]);
    }
    if (lane == 0) {
        *reinterpret_cast<f32x4*>(p.stats_a + row * 4) = f32x4{m1, r1, m2a, r2a};
        *reinterpret_cast<f32x4*>(p.stats_b + row * 4) = f32x4{m1, r1, m2b, r2b};
    }
}

template <typename TD, typename TY, int NC, bool UP, bool GX, bool GD>
__global__ __launch_bounds__(256, NC <= 3 ? 3 : 2) void add_ln_bwd_dual_fast_kernel(AddLnBwdDual p) {
    constexpr int D = 256 * NC;
    constexpr float invD = 1.f / (float)D;
    __shared__ __attribute__((aligned(16))) float red[8 * D];     // [4 waves][D] reduction scratch, then g1a, g2a, g1b, g2b
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* sg = red + 4 * D;                                      // sg + (2 * path + k) * D
    for (int c = threadIdx.x; c < D; c += 256) {
        sg[c] = p.g1a[c]; sg[D + c] = p.g2a[c]; sg[2 * D + c] = p.g1b[c]; sg[3 * D + c] = p.g2b[c];
    }
    __syncthreads();
    const f32x4 z4{0.f, 0.f, 0.f, 0.f};
    f32x4 dg[4][NC];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int c = 0; c < NC; ++c) dg[k][c] = z4;
    const int lo = 4 * lane;
    for (long row = (long)blockIdx.x * 4 + wave; row < p.rows; row += (long)gridDim.x * 4) {
        // wave-uniform row bases + a 32-bit lane offset: the loads take the scalar-base form (no 64-bit address pairs in VGPRs)
        const float* xr = p.x_new + row * D;
        const TY* gar = reinterpret_cast<const TY*>(p.gya) + row * D;
        const TY* gbr = reinterpret_cast<const TY*>(p.gyb) + row * D;
        f32x4 xh[NC], up[NC], tot[NC];
        typename Raw4<TY>::type gar_[NC], gbr_[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) xh[c] = ld4s<float>(xr + lo + 256 * c);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            gar_[c] = ld4raw<TY>(gar + lo + 256 * c);
            gbr_[c] = ld4raw<TY>(gbr + lo + 256 * c);
        }
        if (UP) {
            const float* upr = p.gx_up + row * D;
#pragma unroll
            for (int c = 0; c < NC; ++c) up[c] = ld4s<float>(upr + lo + 256 * c);
        }
        const f32x4 sta = *reinterpret_cast<const f32x4*>(p.stats_a + row * 4);
        const f32x4 stb = *reinterpret_cast<const f32x4*>(p.stats_b + row * 4);
        __builtin_amdgcn_sched_barrier(0);                            // every load of the row is in flight before anything is used
        const float m1 = sta[0], r1 = sta[1];
#pragma unroll
        for (int c = 0; c < NC; ++c) xh[c] = (xh[c] - m1) * r1;
        // one path after the other (their live sets do not overlap: 168 VGPRs hold the row, the four column sums and one path)
#pragma unroll
        for (int path = 0; path < 2; ++path) {
            const float m2 = path ? stb[2] : sta[2], r2 = path ? stb[3] : sta[3];
            int z = 0;                                                // opaque zeros: the gamma reads stay LDS reads of this
            asm volatile("" : "+s"(z));                               // iteration, and uh is RE-computed after the reductions
            const float* G1 = sg + 2 * path * D + z + lo; const float* G2 = G1 + D;
            float a = 0.f, bsum = 0.f;
            f32x4 gvv[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                f32x4& gv = gvv[c];
                gv = cvt4(path ? gbr_[c] : gar_[c]);
                const f32x4 uh = (xh[c] * *reinterpret_cast<const f32x4*>(G1 + 256 * c) - m2) * r2;
                dg[2 * path + 1][c] += gv * uh;
                gv = gv * *reinterpret_cast<const f32x4*>(G2 + 256 * c);
                a += sum4(gv); bsum += sum4(gv * uh);
            }
            const float c1 = wave_sum_v(a) * invD, c2 = wave_sum_v(bsum) * invD;
            int z2 = 0;
            asm volatile("" : "+s"(z2));
            const float* G1b = sg + 2 * path * D + z2 + lo;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const f32x4& gv = gvv[c];
                const f32x4 g1v = *reinterpret_cast<const f32x4*>(G1b + 256 * c);
                const f32x4 uh = (xh[c] * g1v - m2) * r2;
                const f32x4 gu = (gv - c1 - uh * c2) * r2;                // grad wrt u = xhat * g1
                dg[2 * path][c] += gu * xh[c];
                tot[c] = path ? tot[c] + gu * g1v : gu * g1v;            // grad wrt xhat, both paths summed
            }
        }
        float a3 = 0.f, a4 = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) { a3 += sum4(tot[c]); a4 += sum4(tot[c] * xh[c]); }
        const float c3 = wave_sum_v(a3) * invD, c4 = wave_sum_v(a4) * invD;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            f32x4 gx = (tot[c] - c3 - xh[c] * c4) * r1;
            if (UP) gx += up[c];
            if (GX) st4s<float>(p.gx + row * D + lo + 256 * c, gx);
            if (GD) st4s<TD>(reinterpret_cast<TD*>(p.gdelta) + row * D + lo + 256 * c, gx);
        }
    }
#pragma unroll
    for (int qn = 0; qn < 4; ++qn) {
        __syncthreads();
#pragma unroll
        for (int c = 0; c < NC; ++c) *reinterpret_cast<f32x4*>(red + wave * D + lo + 256 * c) = dg[qn][c];
        __syncthreads();
        for (int col = threadIdx.x; col < D; col += 256)
            p.ws[((long)blockIdx.x * 4 + qn) * D + col] = red[col] + red[D + col] + red[2 * D + col] + red[3 * D + col];
    }
}

// blocks of the persistent backward kernels: what is resident at once (256 CUs x blocks per CU at the kernel's footprint)
template <typename K> static int resident_blocks(K kernel, int fallback) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0) != hipSuccess || per_cu < 1) return fallback;
    return per_cu * 256;
}


// ---- dispatch of the fast paths: widths 768 (ViT-B) and 1024 (ViT-L); every other width runs the generic kernels
static bool ln_fast_width(int D) { return D == 768 || D == 1024; }

template <typename TD, int NC>
static int add_ln_fwd_cast_t(const AddLnFwd& p, hipStream_t st) {
    dim3 grid(cdiv(p.rows, 4)), blk(256);
    const bool dbl = p.g2 != nullptr;
    if (p.delta) {
        if (dbl) MMAE_LAUNCH((add_ln_fwd_fast_kernel<TD, float, NC, true, true, true>), grid, blk, 0, st, p);
        else MMAE_LAUNCH((add_ln_fwd_fast_kernel<TD, float, NC, false, true, true>), grid, blk, 0, st, p);
    } else {
        if (dbl) MMAE_LAUNCH((add_ln_fwd_fast_kernel<bf16, float, NC, true, false, true>), grid, blk, 0, st, p);
        else MMAE_LAUNCH((add_ln_fwd_fast_kernel<bf16, float, NC, false, false, true>), grid, blk, 0, st, p);
    }
    MMAE_CHECK_LAUNCH();
    return MMAE_OK;
}

template <typename TD, typename TY, int NC>
static int add_ln_fwd_fast_t(const AddLnFwd& p, hipStream_t st) {
    dim3 grid(cdiv(p.rows, 4)), blk(256);
    const bool dbl = p.g2 != nullptr;
    if (p.delta) {
        if (dbl) MMAE_LAUNCH((add_ln_fwd_fast_kernel<TD, TY, NC, true, true>), grid, blk, 0, st, p);
        else MMAE_LAUNCH((add_ln_fwd_fast_kernel<TD, TY, NC, false, true>), grid, blk, 0, st, p);
    } else {
        if (dbl) MMAE_LAUNCH((add_ln_fwd_fast_kernel<bf16, TY, NC, true, false>), grid, blk, 0, st, p);
        else MMAE_LAUNCH((add_ln_fwd_fast_kernel<bf16, TY, NC, false, false>), grid, blk, 0, st, p);
    }
    MMAE_CHECK_LAUNCH();
    return MMAE_OK;
}
#define LN_FAST_TYPES(FN, ...)                                                                                   \
    (dtype_delta == MMAE_BF16                                                                                    \
         ? (dtype_y == MMAE_BF16 ? (nc3 ? FN<bf16, bf16, 3>(__VA_ARGS__) : FN<bf16, bf16, 4>(__VA_ARGS__))       \
                                 : (nc3 ? FN<bf16, float, 3>(__VA_ARGS__) : FN<bf16, float, 4>(__VA_ARGS__)))    \
         : (dtype_y == MMAE_BF16 ? (nc3 ? FN<float, bf16, 3>(__VA_ARGS__) : FN<float, bf16, 4>(__VA_ARGS__))     \
                                 : (nc3 ? FN<float, float, 3>(__VA_ARGS__) : FN<float, float, 4>(__VA_ARGS__))))
static int add_ln_fwd_fast(int dtype_delta, int dtype_y, const AddLnFwd& p, hipStream_t st) {
    const bool nc3 = p.D == 768;
    return LN_FAST_TYPES(add_ln_fwd_fast_t, p, st);
}

// (UP, GX, GD) as template flags; the delta type only matters when gdelta is written
template <typename TD, typename TY, int NC, bool DOUBLE, bool UP, bool GX>
static int add_ln_bwd_fast_f(const AddLnBwd& p, long want, int* used, hipStream_t st) {
#define GO(KERNEL)                                                                              \
    {                                                                                           \
        static const int resident = resident_blocks(KERNEL, 768);                               \
        long nb = want < resident ? want : resident; if (nb > 1024) nb = 1024; if (nb < 1) nb = 1; \
        *used = (int)nb;                                                                        \
        MMAE_LAUNCH(KERNEL, dim3((unsigned)nb), dim3(256), 0, st, p);                           \
    }
    if (p.gdelta) GO((add_ln_bwd_fast_kernel<TD, TY, NC, DOUBLE, UP, GX, true>))
    else GO((add_ln_bwd_fast_kernel<bf16, TY, NC, DOUBLE, UP, GX, false>))
#undef GO
    MMAE_CHECK_LAUNCH();
    return MMAE_OK;
}
template <typename TD, typename TY, int NC>
static int add_ln_bwd_fast_t(const AddLnBwd& p, long want, int* used, hipStream_t st) {
    const bool dbl = p.g2 != nullptr, up = p.gx_up != nullptr, gx = p.gx != nullptr;
#define F(DB, U, G) add_ln_bwd_fast_f<TD, TY, NC, DB, U, G>(p, want, used, st)
    if (dbl) return up ? (gx ? F(true, true, true) : F(true, true, false)) : (gx ? F(true, false, true) : F(true, false, false));
    return up ? (gx ? F(false, true, true) : F(false, true, false)) : (gx ? F(false, false, true) : F(false, false, false));
#undef F
}
static int add_ln_bwd_fast(int dtype_delta, int dtype_y, const AddLnBwd& p, long want, int* used, hipStream_t st) {
    const bool nc3 = p.D == 768;
    return LN_FAST_TYPES(add_ln_bwd_fast_t, p, want, used, st);
}

template <typename TD, typename TY, int NC>
static int add_ln_fwd_dual_fast_t(const AddLnFwdDual& p, hipStream_t st) {
    dim3 grid(cdiv(p.rows, 4)), blk(256);
    if (p.delta) MMAE_LAUNCH((add_ln_fwd_dual_fast_kernel<TD, TY, NC, true>), grid, blk, 0, st, p);
    else MMAE_LAUNCH((add_ln_fwd_dual_fast_kernel<bf16, TY, NC, false>), grid, blk, 0, st, p);
    MMAE_CHECK_LAUNCH();
    return MMAE_OK;
}
static int add_ln_fwd_dual_fast(int dtype_delta, int dtype_y, const AddLnFwdDual& p, hipStream_t st) {
    const bool nc3 = p.D == 768;
    return LN_FAST_TYPES(add_ln_fwd_dual_fast_t, p, st);
}

template <typename TD, typename TY, int NC, bool UP, bool GX>
static int add_ln_bwd_dual_fast_f(const AddLnBwdDual& p, long want, int* used, hipStream_t st) {
#define GO(KERNEL)                                                                              \
    {                                                                                           \
        static const int resident = resident_blocks(KERNEL, 512);                               \
        long nb = want < resident ? want : resident; if (nb > 1024) nb = 1024; if (nb < 1) nb = 1; \
        *used = (int)nb;                                                                        \
        MMAE_LAUNCH(KERNEL, dim3((unsigned)nb), dim3(256), 0, st, p);                           \
    }
    if (p.gdelta) GO((add_ln_bwd_dual_fast_kernel<TD, TY, NC, UP, GX, true>))
    else GO((add_ln_bwd_dual_fast_kernel<bf16, TY, NC, UP, GX, false>))
#undef GO
    MMAE_CHECK_LAUNCH();
    return MMAE_OK;
}
template <typename TD, typename TY, int NC>
static int add_ln_bwd_dual_fast_t(const AddLnBwdDual& p, long want, int* used, hipStream_t st) {
    const bool up = p.gx_up != nullptr, gx = p.gx != nullptr;
    if (up) return gx ? add_ln_bwd_dual_fast_f<TD, TY, NC, true, true>(p, want, used, st) : add_ln_bwd_dual_fast_f<TD, TY, NC, true, false>(p, want, used, st);
    return gx ? add_ln_bwd_dual_fast_f<TD, TY, NC, false, true>(p, want, used, st) : add_ln_bwd_dual_fast_f<TD, TY, NC, false, false>(p, want, used, st);
}
static int add_ln_bwd_dual_fast(int dtype_delta, int dtype_y, const AddLnBwdDual& p, long want, int* used, hipStream_t st) {
    const bool nc3 = p.D == 768;
    return LN_FAST_TYPES(add_ln_bwd_dual_fast_t, p, want, used, st);
}

#define DISPATCH_TD_TY(FN, ...)                                                                         \
    (dtype_delta == MMAE_BF16                                                                           \
         ? (dtype_y == MMAE_BF16 ? (dbl ? FN<bf16, bf16, true>(__VA_ARGS__) : FN<bf16, bf16, false>(__VA_ARGS__))    \
                                 : (dbl ? FN<bf16, float, true>(__VA_ARGS__) : FN<bf16, float, false>(__VA_ARGS__))) \
         : (dtype_y == MMAE_BF16 ? (dbl ? FN<float, bf16, true>(__VA_ARGS__) : FN<float, bf16, false>(__VA_ARGS__))  \
                                 : (dbl ? FN<float, float, true>(__VA_ARGS__) : FN<float, float, false>(__VA_ARGS__))))

static bool ok_dtype(int d) { return d == MMAE_F32 || d == MMAE_BF16; }

extern "C" int mmae_add_ln_fwd(int dtype_delta, int dtype_y, long rows, int D, const float* x, const void* delta,
                               float* x_new, void* y, const float* gamma1, const float* beta1, float eps1,
                               const float* gamma2, const float* beta2, float eps2, float* stats, void* stream) {
    if (!ok_dtype(dtype_delta) || !ok_dtype(dtype_y) || rows < 0 || D <= 0 || (D % 4) || D > 1024) return MMAE_ERR_ARG;
    if (!x || !y || !gamma1 || !stats) return MMAE_ERR_ARG;
    if (rows == 0) return MMAE_OK;
    AddLnFwd p{x, delta, delta ? x_new : nullptr, y, gamma1, beta1, gamma2, beta2, eps1, eps2, stats, rows, D};
    if (delta && !x_new) return MMAE_ERR_ARG;
    const bool dbl = gamma2 != nullptr;
    if (ln_fast_width(D) && !beta1 && !beta2) return add_ln_fwd_fast(dtype_delta, dtype_y, p, reinterpret_cast<hipStream_t>(stream));
    return DISPATCH_TD_TY(add_ln_fwd_nc, p, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int mmae_add_ln_fwd_cast(int dtype_delta, long rows, int D, const float* x, const void* delta, float* x_new, float* y,
                                    void* y_bf16, const float* gamma1, float eps1, const float* gamma2, float eps2, float* stats,
                                    void* stream) {
    if (!ok_dtype(dtype_delta) || rows < 0 || !ln_fast_width(D)) return MMAE_ERR_ARG;
    if (!x || !y || !y_bf16 || !gamma1 || !stats || (delta && !x_new)) return MMAE_ERR_ARG;
    if (rows == 0) return MMAE_OK;
    AddLnFwd p{x, delta, delta ? x_new : nullptr, y, gamma1, nullptr, gamma2, nullptr, eps1, eps2, stats, rows, D};
    p.y2 = y_bf16;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype_delta == MMAE_BF16) return D == 768 ? add_ln_fwd_cast_t<bf16, 3>(p, st) : add_ln_fwd_cast_t<bf16, 4>(p, st);
    return D == 768 ? add_ln_fwd_cast_t<float, 3>(p, st) : add_ln_fwd_cast_t<float, 4>(p, st);
}

extern "C" int mmae_add_ln_bwd_ws_floats(long rows, int D) {
    long nblk = (rows + 3) / 4; if (nblk > 1024) nblk = 1024; if (nblk < 1) nblk = 1;
    return (int)(nblk * 4 * D);
}

extern "C" int mmae_add_ln_bwd(int dtype_delta, int dtype_y, long rows, int D, const float* x_new, const void* gy,
                               const float* gx_up, const float* gamma1, const float* beta1, const float* gamma2,
                               const float* stats, float* gx, void* gdelta, float* dgamma1, float* dbeta1,
                               float* dgamma2, float* dbeta2, float* ws, int accumulate, void* stream) {
    if (!ok_dtype(dtype_delta) || !ok_dtype(dtype_y) || rows < 0 || D <= 0 || (D % 4) || D > 1024) return MMAE_ERR_ARG;
    if (!x_new || !gy || !gamma1 || !stats || !ws) return MMAE_ERR_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    long nblk = (rows + 3) / 4; if (nblk > 1024) nblk = 1024; if (nblk < 1) nblk = 1;
    AddLnBwd p{x_new, gy, gx_up, gamma1, beta1, gamma2, stats, gx, gdelta, ws, rows, D};
    const bool dbl = gamma2 != nullptr;
    // dbeta outputs are only produced (non-zero) when the kernel keeps beta accumulators
    const bool hasb = beta1 != nullptr || dbeta1 != nullptr || dbeta2 != nullptr;
    int rc;
    if (ln_fast_width(D) && !hasb) {
        int fast_blocks = 0;
        rc = add_ln_bwd_fast(dtype_delta, dtype_y, p, nblk, &fast_blocks, st);
        nblk = fast_blocks;
    } else {
        rc = DISPATCH_TD_TY(add_ln_bwd_nc, p, (int)nblk, hasb, st);
    }
    if (rc) return rc;
    MMAE_LAUNCH(colsum_finalize_kernel, dim3(cdiv(D, 64), dbl ? 4 : 2), dim3(1024), 0, st, ws, (int)nblk, D,
                       dgamma1, dbeta1, dgamma2, dbeta2, accumulate);
    MMAE_CHECK_LAUNCH();
    return MMAE_OK;
}

extern "C" int mmae_add_ln_fwd_dual(int dtype_delta, int dtype_y, long rows, int D, const float* x, const void* delta,
                                    float* x_new, void* y_a, void* y_b, const float* gamma1_a, const float* gamma2_a,
                                    const float* gamma1_b, const float* gamma2_b, float eps1, float eps2, float* stats_a,
                                    float* stats_b, void* stream) {
    if (!ok_dtype(dtype_delta) || !ok_dtype(dtype_y) || rows < 0 || D <= 0 || (D % 4) || D > 1024) return MMAE_ERR_ARG;
    if (!x || !y_a || !y_b || !gamma1_a || !gamma2_a || !gamma1_b || !gamma2_b || !stats_a || !stats_b) return MMAE_ERR_ARG;
    if (delta && !x_new) return MMAE_ERR_ARG;
    if (rows == 0) return MMAE_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    AddLnFwdDual p{x, delta, delta ? x_new : nullptr, y_a, y_b, gamma1_a, gamma2_a, gamma1_b, gamma2_b, eps1, eps2, stats_a, stats_b, rows, D};
    dim3 grid(cdiv(rows, 4)), blk(256);
    const int nc = cdiv(D, 256);
    if (ln_fast_width(D)) return add_ln_fwd_dual_fast(dtype_delta, dtype_y, p, st);
#define GO(TD, TY)                                                                                    \
    switch (nc) {                                                                                     \
        case 1: MMAE_LAUNCH((add_ln_fwd_dual_kernel<TD, TY, 1>), grid, blk, 0, st, p); break;         \
        case 2: MMAE_LAUNCH((add_ln_fwd_dual_kernel<TD, TY, 2>), grid, blk, 0, st, p); break;         \
        case 3: MMAE_LAUNCH((add_ln_fwd_dual_kernel<TD, TY, 3>), grid, blk, 0, st, p); break;         \
        default: MMAE_LAUNCH((add_ln_fwd_dual_kernel<TD, TY, 4>), grid, blk, 0, st, p); break;        \
    }
    if (dtype_delta == MMAE_BF16) { if (dtype_y == MMAE_BF16) { GO(bf16, bf16) } else { GO(bf16, float) } }
    else { if (dtype_y == MMAE_BF16) { GO(float, bf16) } else { GO(float, float) } }
#undef GO
    MMAE_CHECK_LAUNCH();
    return MMAE_OK;
}

extern "C" int mmae_add_ln_bwd_dual(int dtype_delta, int dtype_y, long rows, int D, const float* x_new, const void* gy_a,
                                    const void* gy_b, const float* gx_up, const float* gamma1_a, const float* gamma2_a,
                                    const float* gamma1_b, const float* gamma2_b, const float* stats_a, const float* stats_b,
                                    float* gx, void* gdelta, float* dgamma1_a, float* dgamma2_a, float* dgamma1_b,
                                    float* dgamma2_b, float* ws, int accumulate, void* stream) {
    if (!ok_dtype(dtype_delta) || !ok_dtype(dtype_y) || rows < 0 || D <= 0 || (D % 4) || D > 1024) return MMAE_ERR_ARG;
    if (!x_new || !gy_a || !gy_b || !gamma1_a || !gamma2_a || !gamma1_b || !gamma2_b || !stats_a || !stats_b || !ws) return MMAE_ERR_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    // persistent grid = what is resident at once: 168 VGPRs -> 3 waves / SIMD -> 3 blocks / CU x 256 CUs (a 1024-block grid
    // would run as 768 + 256: two rounds)
    long nblk = (rows + 3) / 4; if (nblk > 768) nblk = 768; if (nblk < 1) nblk = 1;
    AddLnBwdDual p{x_new, gy_a, gy_b, gx_up, gamma1_a, gamma2_a, gamma1_b, gamma2_b, stats_a, stats_b, gx, gdelta, ws, rows, D};
    dim3 grid((unsigned)nblk), blk(256);
    const size_t lds = (size_t)8 * D * sizeof(float);
    const int nc = cdiv(D, 256);
    if (ln_fast_width(D)) {
        int fast_blocks = 0;
        const int rc = add_ln_bwd_dual_fast(dtype_delta, dtype_y, p, (rows + 3) / 4, &fast_blocks, st);
        if (rc) return rc;
        MMAE_LAUNCH(colsum_finalize_kernel, dim3(cdiv(D, 64), 4), dim3(1024), 0, st, ws, fast_blocks, D,
                    dgamma1_a, dgamma2_a, dgamma1_b, dgamma2_b, accumulate);
        MMAE_CHECK_LAUNCH();
        return MMAE_OK;
    }
#define GO(TD, TY)                                                                                      \
    switch (nc) {                                                                                       \
        case 1: MMAE_LAUNCH((add_ln_bwd_dual_kernel<TD, TY, 1>), grid, blk, lds, st, p); break;         \
        case 2: MMAE_LAUNCH((add_ln_bwd_dual_kernel<TD, TY, 2>), grid, blk, lds, st, p); break;         \
        case 3: MMAE_LAUNCH((add_ln_bwd_dual_kernel<TD, TY, 3>), grid, blk, lds, st, p); break;         \
        default: MMAE_LAUNCH((add_ln_bwd_dual_kernel<TD, TY, 4>), grid, blk, lds, st, p); break;        \
    }
    if (dtype_delta == MMAE_BF16) { if (dtype_y == MMAE_BF16) { GO(bf16, bf16) } else { GO(bf16, float) } }
    else { if (dtype_y == MMAE_BF16) { GO(float, bf16) } else { GO(float, float) } }
#undef GO
    MMAE_CHECK_LAUNCH();
    MMAE_LAUNCH(colsum_finalize_kernel, dim3(cdiv(D, 64), 4), dim3(1024), 0, st, ws, (int)nblk, D,
                dgamma1_a, dgamma2_a, dgamma1_b, dgamma2_b, accumulate);
    MMAE_CHECK_LAUNCH();
    return MMAE_OK;
}

// ------------------------------------------------------------------------------------------ GEGLU / GELU
// 16 bytes per lane (8 bf16 / 4 fp32) whenever the width allows.  The Gaussian CDF comes from one exp2 + one rcp
// (Abramowitz-Stegun 7.1.26 for erfc, |abs err| <= 1.5e-7, i.e. fp32 rounding level) instead of the branchy library erff:
// these kernels move 12-20 bytes per element at HBM rate, and the library call made them VALU-bound (4.8 of ~5.8 TB/s).
// In the backward the same exponential also gives the density term of GELU'.
struct GeluParts { float cdf, pdf; };
__device__ __forceinline__ GeluParts gelu_parts(float x) {
    const float e = __builtin_amdgcn_exp2f(-0.72134752044448170f * x * x);            // exp(-x^2/2)
    const float t = __builtin_amdgcn_rcpf(fmaf(0.23164189f, fabsf(x), 1.f));          // 1/(1 + 0.3275911*|x|/sqrt2)
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float half_erfc = 0.5f * poly * t * e;                                        // 0.5*erfc(|x|/sqrt2)
    GeluParts r;
    r.cdf = x >= 0.f ? 1.f - half_erfc : half_erfc;
    r.pdf = 0.39894228040143268f * e;
    return r;
}
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// NT: nontemporal (streaming) access -- these tensors are touched once per kernel and are far larger than the caches
template <typename T, int V, bool NT = false> __device__ __forceinline__ void ldv(const T* p, float* o) {
    if constexpr (V == 1) o[0] = to_f(p[0]);
    else {
        union { u32x4 q; float f[4]; bf16 e[8]; } u;
        if constexpr (NT) u.q = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
        else u.q = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
        for (int j = 0; j < V; ++j) { if constexpr (sizeof(T) == 4) o[j] = u.f[j]; else o[j] = (float)u.e[j]; }
    }
}
template <typename T, int V, bool NT = false> __device__ __forceinline__ void stv(T* p, const float* o) {
    if constexpr (V == 1) p[0] = from_f<T>(o[0]);
    else {
        union { u32x4 q; float f[4]; bf16 e[8]; } u;
#pragma unroll
        for (int j = 0; j < V; ++j) { if constexpr (sizeof(T) == 4) u.f[j] = o[j]; else u.e[j] = (bf16)o[j]; }
        if constexpr (NT) __builtin_nontemporal_store(u.q, reinterpret_cast<u32x4*>(p));
        else *reinterpret_cast<u32x4*>(p) = u.q;
    }
}
// Launch shape of the elementwise kernels: ONE pass -- block b owns the EW_U * 256 consecutive vectors starting at
// b * EW_U * 256, no grid-stride loop -- with nontemporal 16-byte accesses.  Measured at rows = 163840, F = 2048
// (tools/probes/geglu_stream_probe.hip): a 4096-block grid-stride loop moves 4.7 TB/s (blocks in flight touch addresses
// 16 MB apart), one pass 5.7-5.9, one pass + nontemporal 6.05-6.16 TB/s (torch's elementwise add on this chip: 6.25).
// V = elements per lane: 16 / sizeof(T) when the row width is a multiple of it, else 1
constexpr int EW_U = 2;
static bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static long ew_blocks(long n) { long b = (n + 256L * EW_U - 1) / (256L * EW_U); return b < 1 ? 1 : b; }
static int log2_exact(long v) { int s = 0; while ((1L << s) < v) ++s; return (1L << s) == v ? s : -1; }
__device__ __forceinline__ void ew_row_col(long i, int per_row, int sh, long& r, int& c) {
    r = sh >= 0 ? (i >> sh) : i / per_row;                 // wave-uniform choice; the bench widths are powers of two
    c = (int)(i - r * per_row);
}
template <typename T, int V>
__global__ __launch_bounds__(256) void geglu_fwd_kernel(const T* __restrict__ h, T* __restrict__ out, long rows, int F, int sh) {
    constexpr bool NT = V > 1;
    const int per_row = F / V;
    const long n = rows * per_row;
    const long base = (long)blockIdx.x * (256 * EW_U) + threadIdx.x;
    float val[EW_U][V], gate[EW_U][V];
    long r[EW_U]; int c[EW_U];
#pragma unroll
    for (int u = 0; u < EW_U; ++u) {
        const long i = base + u * 256;
        if (i < n) {
            ew_row_col(i, per_row, sh, r[u], c[u]);
            const T* hv = h + r[u] * 2 * F + c[u] * V;
            ldv<T, V, NT>(hv, val[u]); ldv<T, V, NT>(hv + F, gate[u]);
        }
    }
#pragma unroll
    for (int u = 0; u < EW_U; ++u) {
        if (base + u * 256 < n) {
            float o[V];
#pragma unroll
            for (int j = 0; j < V; ++j) o[j] = gate[u][j] * gelu_parts(gate[u][j]).cdf * val[u][j];
            stv<T, V, NT>(out + r[u] * F + c[u] * V, o);
        }
    }
}
template <typename T, int V>
__global__ __launch_bounds__(256) void geglu_bwd_kernel(const T* __restrict__ h, const T* __restrict__ g, T* __restrict__ dh, long rows, int F, int sh) {
    constexpr bool NT = V > 1;
    const int per_row = F / V;
    const long n = rows * per_row;
    const long base = (long)blockIdx.x * (256 * EW_U) + threadIdx.x;
    float val[EW_U][V], gate[EW_U][V], gg[EW_U][V];
    long r[EW_U]; int c[EW_U];
#pragma unroll
    for (int u = 0; u < EW_U; ++u) {
        const long i = base + u * 256;
        if (i < n) {
            ew_row_col(i, per_row, sh, r[u], c[u]);
            const T* hv = h + r[u] * 2 * F + c[u] * V;
            ldv<T, V, NT>(hv, val[u]); ldv<T, V, NT>(hv + F, gate[u]); ldv<T, V, NT>(g + r[u] * F + c[u] * V, gg[u]);
        }
    }
#pragma unroll
    for (int u = 0; u < EW_U; ++u) {
        if (base + u * 256 < n) {
            float dval[V], dgate[V];
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const GeluParts gp = gelu_parts(gate[u][j]);
                dval[j] = gg[u][j] * gate[u][j] * gp.cdf;
                dgate[j] = gg[u][j] * val[u][j] * fmaf(gate[u][j], gp.pdf, gp.cdf);
            }
            T* dv = dh + r[u] * 2 * F + c[u] * V;
            stv<T, V, NT>(dv, dval); stv<T, V, NT>(dv + F, dgate);
        }
    }
}
// n = number of elements, a multiple of V
template <typename T, int V>
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, long n) {
    constexpr bool NT = V > 1;
    const long base = ((long)blockIdx.x * (256 * EW_U) + threadIdx.x) * V;
    float v[EW_U][V];
#pragma unroll
    for (int u = 0; u < EW_U; ++u) if (base + (long)u * 256 * V < n) ldv<T, V, NT>(x + base + (long)u * 256 * V, v[u]);
#pragma unroll
    for (int u = 0; u < EW_U; ++u) {
        const long i = base + (long)u * 256 * V;
        if (i < n) {
            float o[V];
#pragma unroll
            for (int j = 0; j < V; ++j) o[j] = v[u][j] * gelu_parts(v[u][j]).cdf;
            stv<T, V, NT>(y + i, o);
        }
    }
}
template <typename T, int V>
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const T* __restrict__ x, const T* __restrict__ g, T* __restrict__ dx, long n) {
    constexpr bool NT = V > 1;
    const long base = ((long)blockIdx.x * (256 * EW_U) + threadIdx.x) * V;
    float v[EW_U][V], gg[EW_U][V];
#pragma unroll
    for (int u = 0; u < EW_U; ++u) {
        const long i = base + (long)u * 256 * V;
        if (i < n) { ldv<T, V, NT>(x + i, v[u]); ldv<T, V, NT>(g + i, gg[u]); }
    }
#pragma unroll
    for (int u = 0; u < EW_U; ++u) {
        const long i = base + (long)u * 256 * V;
        if (i < n) {
            float o[V];
#pragma unroll
            for (int j = 0; j < V; ++j) { const GeluParts gp = gelu_parts(v[u][j]); o[j] = gg[u][j] * fmaf(v[u][j], gp.pdf, gp.cdf); }
            stv<T, V, NT>(dx + i, o);
        }
    }
}

extern "C" int mmae_geglu_fwd(int dtype, long rows, int F, const void* h, void* out, void* stream) {
    if (!ok_dtype(dtype) || rows < 0 || F <= 0 || !h || !out) return MMAE_ERR_ARG;
    if (rows == 0) return MMAE_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int V = dtype == MMAE_BF16 ? 8 : 4;
    const bool vec = (F % V) == 0 && al16(h) && al16(out);
    const long per_row = vec ? F / V : F, n = rows * per_row;
    if (ew_blocks(n) > 0x7fffffffL) return MMAE_ERR_ARG;
    const int sh = log2_exact(per_row);
#define GO(T, V) MMAE_LAUNCH((geglu_fwd_kernel<T, V>), dim3((unsigned)ew_blocks(n)), dim3(256), 0, st, (const T*)h, (T*)out, rows, F, sh)
    if (dtype == MMAE_BF16) { if (vec) GO(bf16, 8); else GO(bf16, 1); } else { if (vec) GO(float, 4); else GO(float, 1); }
#undef GO
    MMAE_CHECK_LAUNCH();
    return MMAE_OK;
}
extern "C" int mmae_geglu_bwd(int dtype, long rows, int F, const void* h, const void* gout, void* dh, void* stream) {
    if (!ok_dtype(dtype) || rows < 0 || F <= 0 || !h || !gout || !dh) return MMAE_ERR_ARG;
    if (rows == 0) return MMAE_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int V = dtype == MMAE_BF16 ? 8 : 4;
    const bool vec = (F % V) == 0 && al16(h) && al16(gout) && al16(dh);
    const long per_row = vec ? F / V : F, n = rows * per_row;
    if (ew_blocks(n) > 0x7fffffffL) return MMAE_ERR_ARG;
    const int sh = log2_exact(per_row);
#define GO(T, V) MMAE_LAUNCH((geglu_bwd_kernel<T, V>), dim3((unsigned)ew_blocks(n)), dim3(256), 0, st, (const T*)h, (const T*)gout, (T*)dh, rows, F, sh)
    if (dtype == MMAE_BF16) { if (vec) GO(bf16, 8); else GO(bf16, 1); } else { if (vec) GO(float, 4); else GO(float, 1); }
#undef GO
    MMAE_CHECK_LAUNCH();
    return MMAE_OK;
}
extern "C" int mmae_gelu_fwd(int dtype, long n, const void* x, void* y, void* stream) {
    if (!ok_dtype(dtype) || n < 0 || !x || !y) return MMAE_ERR_ARG;
    if (n == 0) return MMAE_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const bool vec = (n % (dtype == MMAE_BF16 ? 8 : 4)) == 0 && al16(x) && al16(y);
    if (ew_blocks(n) > 0x7fffffffL) return MMAE_ERR_ARG;
#define GO(T, V) MMAE_LAUNCH((gelu_fwd_kernel<T, V>), dim3((unsigned)ew_blocks(n / V)), dim3(256), 0, st, (const T*)x, (T*)y, n)
    if (dtype == MMAE_BF16) { if (vec) GO(bf16, 8); else GO(bf16, 1); } else { if (vec) GO(float, 4); else GO(float, 1); }
#undef GO
    MMAE_CHECK_LAUNCH();
    return MMAE_OK;
}
extern "C" int mmae_gelu_bwd(int dtype, long n, const void* x, const void* g, void* dx, void* stream) {
    if (!ok_dtype(dtype) || n < 0 || !x || !g || !dx) return MMAE_ERR_ARG;
    if (n == 0) return MMAE_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const bool vec = (n % (dtype == MMAE_BF16 ? 8 : 4)) == 0 && al16(x) && al16(g) && al16(dx);
    if (ew_blocks(n) > 0x7fffffffL) return MMAE_ERR_ARG;
#define GO(T, V) MMAE_LAUNCH((gelu_bwd_kernel<T, V>), dim3((unsigned)ew_blocks(n / V)), dim3(256), 0, st, (const T*)x, (const T*)g, (T*)dx, n)
    if (dtype == MMAE_BF16) { if (vec) GO(bf16, 8); else GO(bf16, 1); } else { if (vec) GO(float, 4); else GO(float, 1); }
#undef GO
    MMAE_CHECK_LAUNCH();
    return MMAE_OK;
}

// ------------------------------------------------------------------------------------------ per-row scale (DropPath)
// out[r, :] = x[r, :] * scale[r]: stochastic depth (DSI-MM/zorro_utils.py:69-84) applied to a residual branch's output in the packed
// row space -- scale[r] = floor(keep + u_sample(r)) / keep.  Runs only when a Block's drop_path rate is > 0 (reference default 0:
// the bench path never launches it); one pass, 16 bytes per lane; its own backward (the same kernel on the gradient).
template <typename T, int V>
__global__ __launch_bounds__(256) void scale_rows_kernel(const T* __restrict__ x, const float* __restrict__ scale, T* __restrict__ out,
                                                         long rows, int W, int sh) {
    constexpr bool NT = V > 1;
    const int per_row = W / V;
    const long n = rows * per_row;
    const long base = (long)blockIdx.x * (256 * EW_U) + threadIdx.x;
#pragma unroll
    for (int u = 0; u < EW_U; ++u) {
        const long i = base + (long)u * 256;
        if (i < n) {
            long r; int c;
            ew_row_col(i, per_row, sh, r, c);
            float v[V];
            ldv<T, V, NT>(x + r * W + (long)c * V, v);
            const float s = scale[r];
#pragma unroll
            for (int j = 0; j < V; ++j) v[j] *= s;
            stv<T, V, NT>(out + r * W + (long)c * V, v);
        }
    }
}
extern "C" int mmae_scale_rows(int dtype, long rows, int W, const void* x, const float* row_scale, void* out, void* stream) {
    if (!ok_dtype(dtype) || rows < 0 || W <= 0 || !x || !row_scale || !out) return MMAE_ERR_ARG;
    if (rows == 0) return MMAE_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int V = dtype == MMAE_BF16 ? 8 : 4;
    const bool vec = (W % V) == 0 && al16(x) && al16(out);
    const long per_row = vec ? W / V : W, n = rows * per_row;
    if (ew_blocks(n) > 0x7fffffffL) return MMAE_ERR_ARG;
    const int sh = log2_exact(per_row);
#define GO(T, V) MMAE_LAUNCH((scale_rows_kernel<T, V>), dim3((unsigned)ew_blocks(n)), dim3(256), 0, st, (const T*)x, row_scale, (T*)out, rows, W, sh)
    if (dtype == MMAE_BF16) { if (vec) GO(bf16, 8); else GO(bf16, 1); } else { if (vec) GO(float, 4); else GO(float, 1); }
#undef GO
    MMAE_CHECK_LAUNCH();
    return MMAE_OK;
}

// ------------------------------------------------------------------------------------------ row gather / scatter
// out[r, :] = src[idx[r], :]   (idx < 0 -> zeros).  W multiple of 4.  One wave per row.
template <typename T>
__global__ __launch_bounds__(256) void gather_rows_kernel(const T* __restrict__ src, const int* __restrict__ idx,
                                                          T* __restrict__ out, long rows, int W, long src_stride, long out_stride) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int s = idx[r];
    for (int c = 4 * lane; c < W; c += 256) {
        f32x4 v{0.f, 0.f, 0.f, 0.f};
        if (s >= 0) v = ld4f<T>(src + (long)s * src_stride + c);
        st4f<T>(out + r * out_stride + c, v);
    }
}
// dst[idx[r], :] (+)= src[r, :]  for idx[r] >= 0 (and filter[r] == filter_value when a filter is given).  Plain
// read-modify-write, no atomics: the indices of the rows that pass the filter must be unique within one call.
template <typename T, bool ACC>
__global__ __launch_bounds__(256) void scatter_rows_kernel(const T* __restrict__ src, const int* __restrict__ idx,
                                                           T* __restrict__ dst, long rows, int W, long src_stride, long dst_stride,
                                                           const int* __restrict__ filter, int filter_value) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int d = idx[r];
    if (d < 0) return;
    if (filter && filter[r] != filter_value) return;
    for (int c = 4 * lane; c < W; c += 256) {
        f32x4 v = ld4f<T>(src + r * src_stride + c);
        if (ACC) v += ld4f<T>(dst + (long)d * dst_stride + c);
        st4f<T>(dst + (long)d * dst_stride + c, v);
    }
}

extern "C" int mmae_gather_rows(int dtype, long rows, int W, const void* src, long src_stride, const int* idx, void* out,
                                long out_stride, void* stream) {
    if (!ok_dtype(dtype) || rows < 0 || W <= 0 || (W % 4) || !src || !idx || !out || (src_stride % 4) || (out_stride % 4)) return MMAE_ERR_ARG;
    if (rows == 0) return MMAE_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MMAE_BF16) MMAE_LAUNCH((gather_rows_kernel<bf16>), dim3(cdiv(rows, 4)), dim3(256), 0, st, (const bf16*)src, idx, (bf16*)out, rows, W, src_stride, out_stride);
    else MMAE_LAUNCH((gather_rows_kernel<float>), dim3(cdiv(rows, 4)), dim3(256), 0, st, (const float*)src, idx, (float*)out, rows, W, src_stride, out_stride);
    MMAE_CHECK_LAUNCH();
    return MMAE_OK;
}
extern "C" int mmae_scatter_rows(int dtype, long rows, int W, const void* src, long src_stride, const int* idx, void* dst,
                                 long dst_stride, int accumulate, const int* filter, int filter_value, void* stream) {
    if (!ok_dtype(dtype) || rows < 0 || W <= 0 || (W % 4) || !src || !idx || !dst || (src_stride % 4) || (dst_stride % 4)) return MMAE_ERR_ARG;
    if (rows == 0) return MMAE_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    dim3 grid(cdiv(rows, 4)), blk(256);
    if (dtype == MMAE_BF16) {
        if (accumulate) MMAE_LAUNCH((scatter_rows_kernel<bf16, true>), grid, blk, 0, st, (const bf16*)src, idx, (bf16*)dst, rows, W, src_stride, dst_stride, filter, filter_value);
        else MMAE_LAUNCH((scatter_rows_kernel<bf16, false>), grid, blk, 0, st, (const bf16*)src, idx, (bf16*)dst, rows, W, src_stride, dst_stride, filter, filter_value);
    } else {
        if (accumulate) MMAE_LAUNCH((scatter_rows_kernel<float, true>), grid, blk, 0, st, (const float*)src, idx, (float*)dst, rows, W, src_stride, dst_stride, filter, filter_value);
        else MMAE_LAUNCH((scatter_rows_kernel<float, false>), grid, blk, 0, st, (const float*)src, idx, (float*)dst, rows, W, src_stride, dst_stride, filter, filter_value);
    }
    MMAE_CHECK_LAUNCH();
    return MMAE_OK;
}


// ------------------------------------------------------------------------------------------ column sums (bias gradients)
// out[c] = sum_r x[r, c] in fp32 (autograd of the `+ bias` of nn.Linear: decoder qkv/proj/fc1/fc2, proj_context, out_proj,
// Mlp -- MM/multimae_utils.py:138-182, MM/output_adapters_simple.py:166-181).  Stage 1: every block sums a strip of rows,
// 8 (bf16) / 4 (fp32) columns per lane, 16-byte loads; stage 2: the partial rows are added in fixed order (deterministic).
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ x, long rows, int cols, long ld,
                                                             int rows_per_block, float* __restrict__ ws) {
    constexpr int V = 16 / sizeof(T);
    __shared__ float red[256 * V];
    const int cg = cols / V;                                  // column groups of 16 bytes (cols % V == 0)
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = min(rows, r0 + rows_per_block);
    auto load_add = [&](long r, int g, float (&acc)[V]) {
        if (sizeof(T) == 2) {
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16*>(x) + r * ld + (long)g * V);
#pragma unroll
            for (int j = 0; j < V; ++j) acc[j] += (float)v[j];
        } else {
            const f32x4 v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(x) + r * ld + (long)g * V);
#pragma unroll
            for (int j = 0; j < V; ++j) acc[j] += v[j];
        }
    };
    if (cg <= 256) {
        // 256 / cg rows in flight per iteration: thread (rr, g) walks rows rr, rr + rpi, ... of the strip in column group g
        const int rpi = 256 / cg, g = threadIdx.x % cg, rr = threadIdx.x / cg;
        float acc[V];
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] = 0.f;
        if (rr < rpi)
            for (long r = r0 + rr; r < r1; r += rpi) load_add(r, g, acc);
#pragma unroll
        for (int j = 0; j < V; ++j) red[threadIdx.x * V + j] = acc[j];
        __syncthreads();
        if (threadIdx.x < cg) {
#pragma unroll
            for (int j = 0; j < V; ++j) {
                float sum = 0.f;
                for (int k = 0; k < rpi; ++k) sum += red[(k * cg + threadIdx.x) * V + j];       // fixed order: deterministic
                ws[(long)blockIdx.x * cols + (long)threadIdx.x * V + j] = sum;
            }
        }
    } else {
        for (int g = threadIdx.x; g < cg; g += 256) {
            float acc[V];
#pragma unroll
            for (int j = 0; j < V; ++j) acc[j] = 0.f;
            for (long r = r0; r < r1; ++r) load_add(r, g, acc);
#pragma unroll
            for (int j = 0; j < V; ++j) ws[(long)blockIdx.x * cols + (long)g * V + j] = acc[j];
        }
    }
}
// out[c] = sum_b ws[b][c]: 16 partial-row groups x 64 columns per block (coalesced 256 B per group), four independent loads in
// flight per thread, combined in fixed order (the 4-group, one-load-at-a-time version spent 16 us on 256 x 1024 floats:
// 64 dependent trips per thread)
__global__ __launch_bounds__(1024) void colsum_finish_kernel(const float* __restrict__ ws, int nblk, int cols, float* __restrict__ out) {
    __shared__ float red[1024];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f, s5 = 0.f, s6 = 0.f, s7 = 0.f;
    if (c < cols) {
        const float* base = ws + c;
        int b = rg;
        for (; b + 112 < nblk; b += 128) {
            s0 += base[(long)b * cols]; s1 += base[(long)(b + 16) * cols];
            s2 += base[(long)(b + 32) * cols]; s3 += base[(long)(b + 48) * cols];
            s4 += base[(long)(b + 64) * cols]; s5 += base[(long)(b + 80) * cols];
            s6 += base[(long)(b + 96) * cols]; s7 += base[(long)(b + 112) * cols];
        }
        for (; b < nblk; b += 16) s0 += base[(long)b * cols];
    }
    red[threadIdx.x] = ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7));
    __syncthreads();
    if (rg == 0 && c < cols) {
        float s = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) s += red[g * 64 + (threadIdx.x & 63)];
        out[c] = s;
    }
}

static int colsum_rows_per_block(long rows, int cols, int V) {
    const int cg = cols / V, rpi = cg <= 256 ? 256 / cg : 1;
    long rpb = (rows + 255) / 256;                                     // at most 256 partial rows
    if (rpb < 16L * rpi) rpb = 16L * rpi;                              // and at least 16 loads per thread
    return (int)rpb;
}
static int colsum_blocks(long rows, int cols, int V) {
    const int rpb = colsum_rows_per_block(rows, cols, V);
    long b = (rows + rpb - 1) / rpb;
    return (int)(b < 1 ? 1 : b);
}
extern "C" long mmae_colsum_ws_floats(long rows, int cols) { return rows < 0 || cols <= 0 ? MMAE_ERR_ARG : 256L * cols; }
extern "C" int mmae_colsum(int dtype, long rows, int cols, const void* x, long ld, float* out, float* ws, void* stream) {
    if (!ok_dtype(dtype) || rows < 0 || cols <= 0 || !x || !out || !ws) return MMAE_ERR_ARG;
    const int V = dtype == MMAE_BF16 ? 8 : 4;
    if ((cols % V) || (ld % V) || ld < cols || (reinterpret_cast<uintptr_t>(x) & 15)) return MMAE_ERR_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int nb = colsum_blocks(rows, cols, V), rpb = colsum_rows_per_block(rows, cols, V);
    if (dtype == MMAE_BF16) MMAE_LAUNCH((colsum_partial_kernel<bf16>), dim3(nb), dim3(256), 0, st, (const bf16*)x, rows, cols, ld, rpb, ws);
    else MMAE_LAUNCH((colsum_partial_kernel<float>), dim3(nb), dim3(256), 0, st, (const float*)x, rows, cols, ld, rpb, ws);
    MMAE_CHECK_LAUNCH();
    MMAE_LAUNCH(colsum_finish_kernel, dim3(cdiv(cols, 64)), dim3(1024), 0, st, ws, nb, cols, out);
    MMAE_CHECK_LAUNCH();
    return MMAE_OK;
}

// ------------------------------------------------------------------------------------------ bench calibration (mmae_internal.h)
// A plain 16-byte-per-lane streaming copy under its own kernel name: bench.py keeps the queue busy with it while it measures what
// an empty HIP-event bracket costs, and a rocprofv3 summary of the bench command shows these launches as `calib_copy_kernel`
// instead of inflating the row of a kernel that belongs to the step.
__global__ __launch_bounds__(256) void calib_copy_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, long n16) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}
extern "C" int mmae_debug_stream_copy(long n_bytes, const void* src, void* dst, void* stream) {
    if (n_bytes <= 0 || (n_bytes % 16) || !src || !dst) return MMAE_ERR_ARG;
    const long n16 = n_bytes / 16;
    MMAE_LAUNCH(calib_copy_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                reinterpret_cast<const f32x4*>(src), reinterpret_cast<f32x4*>(dst), n16);
    MMAE_CHECK_LAUNCH();
    return MMAE_OK;
}
