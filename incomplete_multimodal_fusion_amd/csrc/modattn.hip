// Modality attention of Block_Fusion (gfx950): for every (sample, patch) the fusion token attends the M modality
// slots of that patch plus itself -- sequence length M+1 <= 8.
//
// Reference: multimae_crossattn.py:454-462 builds all_tokens (B,P,M+1,D) by cloning mask_embedding and scattering
// the kept tokens, then Block_Fusion.forward (downstream/.../zorro_utils.py:252-258) runs full attention over the
// M+1 slots and keeps only the fusion slot (:256).  Rows are independent, so only the fusion slot's query is needed and
// K/V of a masked slot depend on the patch only (they come from mask_embedding[p]).  Here K/V live in ONE matrix
// kv[(token rows | P mask-embedding rows), 2*I] and `slot_row[(b,p), s]` names the row each slot reads; the
// (B,P,M+1,D) tensor is never materialised.  HBM-bound: one wave per (b,p), 16 B per lane, heads = groups of DH/8 lanes.
//
// Backward: grid (patch p, sample group); a block's 4 waves stride over the samples of its group; token rows are written
// once each (every token row belongs to exactly one slot); the shared mask-embedding row `shared_base + p` is accumulated
// in registers over the samples, reduced over the 4 waves in LDS into an fp32 partial slab per sample group, and the
// slabs are summed in fixed order by a second small kernel -> deterministic, no atomics.
#include "common.hpp"
#include "mmae_hip.h"

#define MA_MAXS 8

template <typename T> __device__ __forceinline__ void ld8f(const T* p, float (&o)[8]);
template <> __device__ __forceinline__ void ld8f<float>(const float* p, float (&o)[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { o[j] = a[j]; o[4 + j] = b[j]; }
}
template <> __device__ __forceinline__ void ld8f<bf16>(const bf16* p, float (&o)[8]) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (float)v[j];
}
template <typename T> __device__ __forceinline__ void st8f(T* p, const float (&v)[8]);
template <> __device__ __forceinline__ void st8f<float>(float* p, const float (&v)[8]) {
    *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
}
template <> __device__ __forceinline__ void st8f<bf16>(bf16* p, const float (&v)[8]) {
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16)v[j];
    *reinterpret_cast<bf16x8*>(p) = o;
}

template <int DH> __device__ __forceinline__ float head_sum(float v) {
#pragma unroll
    for (int o = 1; o < DH / 8; o <<= 1) v += __shfl_xor(v, o);
    return v;
}

struct ModAttn {
    const void* q; const void* kv; const int* slot_row; void* out;
    const void* dout; void* dq; void* dkv;
    long q_stride, kv_stride, out_stride, do_stride, dq_stride, dkv_stride;
    long rows;            // B*P
    int I, ns, P, B, shared_base;
    float scale;
    float* ws;            // backward: (nsplit, P, 2*I) fp32 partial sums of the shared rows
    int nsplit;
};

template <typename T, int DH>
__global__ __launch_bounds__(256) void modattn_fwd_kernel(ModAttn p) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.rows) return;                 // wave-uniform
    const T* q = reinterpret_cast<const T*>(p.q) + row * p.q_stride;
    const T* kv = reinterpret_cast<const T*>(p.kv);
    T* out = reinterpret_cast<T*>(p.out) + row * p.out_stride;
    int srow[MA_MAXS];
#pragma unroll
    for (int s = 0; s < MA_MAXS; ++s) srow[s] = s < p.ns ? p.slot_row[row * p.ns + s] : 0;
    for (int c0 = 0; c0 < p.I; c0 += 512) {
        const int col = c0 + 8 * lane;
        const bool act = col < p.I;
        float q8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) q8[j] = 0.f;
        if (act) ld8f<T>(q + col, q8);
        float sc[MA_MAXS];
        float mx = -INFINITY;
#pragma unroll
        for (int s = 0; s < MA_MAXS; ++s) {
            sc[s] = -INFINITY;
            if (s < p.ns) {
                float k8[8], d = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) k8[j] = 0.f;
                if (act) ld8f<T>(kv + (long)srow[s] * p.kv_stride + col, k8);
#pragma unroll
                for (int j = 0; j < 8; ++j) d += q8[j] * k8[j];
                sc[s] = head_sum<DH>(d) * p.scale;
                mx = fmaxf(mx, sc[s]);
            }
        }
        float den = 0.f;
#pragma unroll
        for (int s = 0; s < MA_MAXS; ++s) { sc[s] = s < p.ns ? __expf(sc[s] - mx) : 0.f; den += sc[s]; }
        const float inv = 1.f / den;
        float o8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o8[j] = 0.f;
#pragma unroll
        for (int s = 0; s < MA_MAXS; ++s) {
            if (s < p.ns && act) {
                float v8[8];
                ld8f<T>(kv + (long)srow[s] * p.kv_stride + p.I + col, v8);
                const float w = sc[s] * inv;
#pragma unroll
                for (int j = 0; j < 8; ++j) o8[j] += w * v8[j];
            }
        }
        if (act) st8f<T>(out + col, o8);
    }
}

// NCH = ceil(I / 512) register chunks of shared-row accumulators
template <typename T, int DH, int NCH>
__global__ __launch_bounds__(256) void modattn_bwd_kernel(ModAttn p) {
    extern __shared__ __attribute__((aligned(16))) float red[];   // [4][2*I]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pp = blockIdx.x;                                     // patch
    const int sp = blockIdx.y;                                     // sample group: b = sp*4 + wave, step 4*nsplit
    const T* kv = reinterpret_cast<const T*>(p.kv);
    T* dkv = reinterpret_cast<T*>(p.dkv);
    float accK[NCH][8], accV[NCH][8];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
        for (int j = 0; j < 8; ++j) { accK[ch][j] = 0.f; accV[ch][j] = 0.f; }

    for (int b = sp * 4 + wave; b < p.B; b += 4 * p.nsplit) {
        const long row = (long)b * p.P + pp;
        const T* q = reinterpret_cast<const T*>(p.q) + row * p.q_stride;
        const T* dout = reinterpret_cast<const T*>(p.dout) + row * p.do_stride;
        T* dq = reinterpret_cast<T*>(p.dq) + row * p.dq_stride;
        int srow[MA_MAXS];
#pragma unroll
        for (int s = 0; s < MA_MAXS; ++s) srow[s] = s < p.ns ? p.slot_row[row * p.ns + s] : 0;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int col = ch * 512 + 8 * lane;
            const bool act = col < p.I;
            float q8[8], g8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { q8[j] = 0.f; g8[j] = 0.f; }
            if (act) { ld8f<T>(q + col, q8); ld8f<T>(dout + col, g8); }
            float sc[MA_MAXS], dp[MA_MAXS];
            float mx = -INFINITY;
#pragma unroll
            for (int s = 0; s < MA_MAXS; ++s) {
                sc[s] = -INFINITY; dp[s] = 0.f;
                if (s < p.ns) {
                    float k8[8], v8[8], d = 0.f, e = 0.f;
#pragma unroll
                    for (int j = 0; j < 8; ++j) { k8[j] = 0.f; v8[j] = 0.f; }
                    if (act) {
                        ld8f<T>(kv + (long)srow[s] * p.kv_stride + col, k8);
                        ld8f<T>(kv + (long)srow[s] * p.kv_stride + p.I + col, v8);
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) { d += q8[j] * k8[j]; e += g8[j] * v8[j]; }
                    sc[s] = head_sum<DH>(d) * p.scale;
                    dp[s] = head_sum<DH>(e);
                    mx = fmaxf(mx, sc[s]);
                }
            }
            float den = 0.f;
#pragma unroll
            for (int s = 0; s < MA_MAXS; ++s) { sc[s] = s < p.ns ? __expf(sc[s] - mx) : 0.f; den += sc[s]; }
            const float inv = 1.f / den;
            float dot = 0.f;
#pragma unroll
            for (int s = 0; s < MA_MAXS; ++s) { sc[s] *= inv; dot += sc[s] * dp[s]; }
            float dq8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) dq8[j] = 0.f;
#pragma unroll
            for (int s = 0; s < MA_MAXS; ++s) {
                if (s < p.ns && act) {
                    const float ds = sc[s] * (dp[s] - dot) * p.scale;
                    float k8[8], dk8[8], dv8[8];
                    ld8f<T>(kv + (long)srow[s] * p.kv_stride + col, k8);
#pragma unroll
                    for (int j = 0; j < 8; ++j) { dq8[j] += ds * k8[j]; dk8[j] = ds * q8[j]; dv8[j] = sc[s] * g8[j]; }
                    if (srow[s] < p.shared_base) {
                        st8f<T>(dkv + (long)srow[s] * p.dkv_stride + col, dk8);
                        st8f<T>(dkv + (long)srow[s] * p.dkv_stride + p.I + col, dv8);
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) { accK[ch][j] += dk8[j]; accV[ch][j] += dv8[j]; }
                    }
                }
            }
            if (act) st8f<T>(dq + col, dq8);
        }
    }
    // reduce the shared (mask-embedding) row over the 4 waves and store it once
    const int I2 = 2 * p.I;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int col = ch * 512 + 8 * lane;
        if (col < p.I) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { red[wave * I2 + col + j] = accK[ch][j]; red[wave * I2 + p.I + col + j] = accV[ch][j]; }
        }
    }
    __syncthreads();
    float* wrow = p.ws + ((long)sp * p.P + pp) * I2;
    for (int c = threadIdx.x; c < I2; c += 256) wrow[c] = red[c] + red[I2 + c] + red[2 * I2 + c] + red[3 * I2 + c];
}

// shared (mask-embedding) rows: sum the per-sample-group partial slabs in a fixed order -> deterministic
template <typename T>
__global__ __launch_bounds__(256) void modattn_bwd_finish_kernel(ModAttn p) {
    const int I2 = 2 * p.I;
    const long n = (long)p.P * I2;
    T* dkv = reinterpret_cast<T*>(p.dkv);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float s = 0.f;
        for (int k = 0; k < p.nsplit; ++k) s += p.ws[(long)k * n + i];
        const long pp = i / I2, c = i % I2;
        dkv[(p.shared_base + pp) * p.dkv_stride + c] = from_f<T>(s);
    }
}

// ------------------------------------------------------------------------------------------ fast paths (I == 512, NS slots)
// The generic kernels test `s < p.ns` and `col < I` around every access; each test became a branch with an s_waitcnt vmcnt(0)
// behind it, so the 2 * ns + 1 row pieces of a (sample, patch) were fetched one after the other.  Same arithmetic here with
// the slot count as a template parameter and one 512-column chunk: the slot rows come from scalar loads, every K / V piece is
// requested before the first one is used, head sums run on DPP.
template <int CTRL> __device__ __forceinline__ float ma_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int DH> __device__ __forceinline__ float head_sum_v(float v) {
    v += ma_dpp<0xB1>(v);                      // quad_perm [1,0,3,2]
    v += ma_dpp<0x4E>(v);                      // quad_perm [2,3,0,1]
    if (DH == 64) v += ma_dpp<0x141>(v);       // row_half_mirror: lane i <-> 7 - i of its group of 8
    return v;
}
template <typename T> __device__ __forceinline__ void cvt8(const typename Vec8<T>::type& r, float (&o)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = to_f(r[j]);
}

template <typename T, int DH, int NS>
__global__ __launch_bounds__(256) void modattn_fwd_fast_kernel(ModAttn p) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (row >= p.rows) return;
    const int col = 8 * lane;
    const T* kv = reinterpret_cast<const T*>(p.kv);
    int srow[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) srow[s] = __builtin_amdgcn_readfirstlane(p.slot_row[row * NS + s]);
    typename Vec8<T>::type qr, kr[NS], vr[NS];
    qr = ld8<T>(reinterpret_cast<const T*>(p.q) + row * p.q_stride + col);
#pragma unroll
    for (int s = 0; s < NS; ++s) kr[s] = ld8<T>(kv + (long)srow[s] * p.kv_stride + col);
#pragma unroll
    for (int s = 0; s < NS; ++s) vr[s] = ld8<T>(kv + (long)srow[s] * p.kv_stride + 512 + col);
    __builtin_amdgcn_sched_barrier(0);
    float q8[8], sc[NS];
    cvt8<T>(qr, q8);
    float mx = -INFINITY;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        float k8[8], d = 0.f;
        cvt8<T>(kr[s], k8);
#pragma unroll
        for (int j = 0; j < 8; ++j) d += q8[j] * k8[j];
        sc[s] = head_sum_v<DH>(d) * p.scale;
        mx = fmaxf(mx, sc[s]);
    }
    float den = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) { sc[s] = __expf(sc[s] - mx); den += sc[s]; }
    const float inv = 1.f / den;
    float o8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o8[j] = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        float v8[8];
        cvt8<T>(vr[s], v8);
        const float w = sc[s] * inv;
#pragma unroll
        for (int j = 0; j < 8; ++j) o8[j] += w * v8[j];
    }
    st8f<T>(reinterpret_cast<T*>(p.out) + row * p.out_stride + col, o8);
}

template <typename T, int DH, int NS>
__global__ __launch_bounds__(256) void modattn_bwd_fast_kernel(ModAttn p) {
    __shared__ __attribute__((aligned(16))) float red[4 * 1024];  // [4][2 * I]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pp = blockIdx.x;                                     // patch
    const int sp = blockIdx.y;                                     // sample group: b = sp*4 + wave, step 4*nsplit
    const T* kv = reinterpret_cast<const T*>(p.kv);
    T* dkv = reinterpret_cast<T*>(p.dkv);
    const int col = 8 * lane;
    float accK[8], accV[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { accK[j] = 0.f; accV[j] = 0.f; }
    for (int b = sp * 4 + wave; b < p.B; b += 4 * p.nsplit) {
        const long row = (long)b * p.P + pp;
        int srow[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) srow[s] = __builtin_amdgcn_readfirstlane(p.slot_row[row * NS + s]);
        typename Vec8<T>::type qr, gr, kr[NS], vr[NS];
        qr = ld8<T>(reinterpret_cast<const T*>(p.q) + row * p.q_stride + col);
        gr = ld8<T>(reinterpret_cast<const T*>(p.dout) + row * p.do_stride + col);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            kr[s] = ld8<T>(kv + (long)srow[s] * p.kv_stride + col);
            vr[s] = ld8<T>(kv + (long)srow[s] * p.kv_stride + 512 + col);
        }
        __builtin_amdgcn_sched_barrier(0);
        float q8[8], g8[8], sc[NS], dp[NS];
        cvt8<T>(qr, q8); cvt8<T>(gr, g8);
        float mx = -INFINITY;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            float k8[8], v8[8], d = 0.f, e = 0.f;
            cvt8<T>(kr[s], k8); cvt8<T>(vr[s], v8);
#pragma unroll
            for (int j = 0; j < 8; ++j) { d += q8[j] * k8[j]; e += g8[j] * v8[j]; }
            sc[s] = head_sum_v<DH>(d) * p.scale;
            dp[s] = head_sum_v<DH>(e);
            mx = fmaxf(mx, sc[s]);
        }
        float den = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) { sc[s] = __expf(sc[s] - mx); den += sc[s]; }
        const float inv = 1.f / den;
        float dot = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) { sc[s] *= inv; dot += sc[s] * dp[s]; }
        float dq8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) dq8[j] = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const float ds = sc[s] * (dp[s] - dot) * p.scale;
            float k8[8], dk8[8], dv8[8];
            cvt8<T>(kr[s], k8);
#pragma unroll
            for (int j = 0; j < 8; ++j) { dq8[j] += ds * k8[j]; dk8[j] = ds * q8[j]; dv8[j] = sc[s] * g8[j]; }
            if (srow[s] < p.shared_base) {                          // wave-uniform
                st8f<T>(dkv + (long)srow[s] * p.dkv_stride + col, dk8);
                st8f<T>(dkv + (long)srow[s] * p.dkv_stride + 512 + col, dv8);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) { accK[j] += dk8[j]; accV[j] += dv8[j]; }
            }
        }
        st8f<T>(reinterpret_cast<T*>(p.dq) + row * p.dq_stride + col, dq8);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[wave * 1024 + col + j] = accK[j]; red[wave * 1024 + 512 + col + j] = accV[j]; }
    __syncthreads();
    float* wrow = p.ws + ((long)sp * p.P + pp) * 1024;
    for (int c = threadIdx.x; c < 1024; c += 256) wrow[c] = red[c] + red[1024 + c] + red[2048 + c] + red[3072 + c];
}

template <typename T, int DH>
static bool modattn_fwd_fast(const ModAttn& p, dim3 grid, hipStream_t st) {
    switch (p.ns) {
        case 2: MMAE_LAUNCH((modattn_fwd_fast_kernel<T, DH, 2>), grid, dim3(256), 0, st, p); return true;
        case 3: MMAE_LAUNCH((modattn_fwd_fast_kernel<T, DH, 3>), grid, dim3(256), 0, st, p); return true;
        case 4: MMAE_LAUNCH((modattn_fwd_fast_kernel<T, DH, 4>), grid, dim3(256), 0, st, p); return true;
        case 5: MMAE_LAUNCH((modattn_fwd_fast_kernel<T, DH, 5>), grid, dim3(256), 0, st, p); return true;
        default: return false;
    }
}
template <typename T, int DH>
static bool modattn_bwd_fast(const ModAttn& p, dim3 grid, hipStream_t st) {
    switch (p.ns) {
        case 2: MMAE_LAUNCH((modattn_bwd_fast_kernel<T, DH, 2>), grid, dim3(256), 0, st, p); return true;
        case 3: MMAE_LAUNCH((modattn_bwd_fast_kernel<T, DH, 3>), grid, dim3(256), 0, st, p); return true;
        case 4: MMAE_LAUNCH((modattn_bwd_fast_kernel<T, DH, 4>), grid, dim3(256), 0, st, p); return true;
        case 5: MMAE_LAUNCH((modattn_bwd_fast_kernel<T, DH, 5>), grid, dim3(256), 0, st, p); return true;
        default: return false;
    }
}

static bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static int ma_check(int dtype, int head_dim, int B, int P, int ns, int I, long a, long b, long c) {
    if (dtype != MMAE_F32 && dtype != MMAE_BF16) return MMAE_ERR_ARG;
    if (head_dim != 32 && head_dim != 64) return MMAE_ERR_ARG;
    if (B <= 0 || P <= 0 || ns <= 0 || ns > MA_MAXS || I <= 0 || (I % head_dim) || I > 1024) return MMAE_ERR_ARG;
    if ((a % 8) || (b % 8) || (c % 8)) return MMAE_ERR_ARG;
    return MMAE_OK;
}

extern "C" int mmae_modattn_fwd(int dtype, int head_dim, int B, int P, int ns, int inner, const void* q, long q_stride,
                                const void* kv, long kv_stride, const int* slot_row, void* out, long out_stride,
                                float scale, void* stream) {
    int rc = ma_check(dtype, head_dim, B, P, ns, inner, q_stride, kv_stride, out_stride);
    if (rc) return rc;
    if (!q || !kv || !slot_row || !out || !al16(q) || !al16(kv) || !al16(out)) return MMAE_ERR_ARG;
    ModAttn p{};
    p.q = q; p.kv = kv; p.slot_row = slot_row; p.out = out; p.q_stride = q_stride; p.kv_stride = kv_stride;
    p.out_stride = out_stride; p.rows = (long)B * P; p.I = inner; p.ns = ns; p.P = P; p.B = B; p.scale = scale;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    dim3 grid(cdiv(p.rows, 4)), blk(256);
    if (inner == 512 && ns >= 2 && ns <= 5) {
        const bool ok = dtype == MMAE_BF16 ? (head_dim == 64 ? modattn_fwd_fast<bf16, 64>(p, grid, st) : modattn_fwd_fast<bf16, 32>(p, grid, st))
                                           : (head_dim == 64 ? modattn_fwd_fast<float, 64>(p, grid, st) : modattn_fwd_fast<float, 32>(p, grid, st));
        if (ok) { MMAE_CHECK_LAUNCH(); return MMAE_OK; }
    }
    if (dtype == MMAE_BF16) {
        if (head_dim == 64) MMAE_LAUNCH((modattn_fwd_kernel<bf16, 64>), grid, blk, 0, st, p);
        else MMAE_LAUNCH((modattn_fwd_kernel<bf16, 32>), grid, blk, 0, st, p);
    } else {
        if (head_dim == 64) MMAE_LAUNCH((modattn_fwd_kernel<float, 64>), grid, blk, 0, st, p);
        else MMAE_LAUNCH((modattn_fwd_kernel<float, 32>), grid, blk, 0, st, p);
    }
    MMAE_CHECK_LAUNCH();
    return MMAE_OK;
}

// sample groups per patch in the backward (each block's 4 waves stride over the samples of its group)
extern "C" int mmae_modattn_bwd_nsplit(int B) { int n = B / 32; return n < 1 ? 1 : (n > 16 ? 16 : n); }

extern "C" int mmae_modattn_bwd(int dtype, int head_dim, int B, int P, int ns, int inner, const void* q, long q_stride,
                                const void* kv, long kv_stride, const int* slot_row, const void* dout, long do_stride,
                                void* dq, long dq_stride, void* dkv, long dkv_stride, int shared_base, float scale,
                                float* ws, void* stream) {
    int rc = ma_check(dtype, head_dim, B, P, ns, inner, q_stride, kv_stride, do_stride);
    if (rc) return rc;
    if ((dq_stride % 8) || (dkv_stride % 8) || shared_base < 0) return MMAE_ERR_ARG;
    if (!q || !kv || !slot_row || !dout || !dq || !dkv || !ws || !al16(q) || !al16(kv) || !al16(dout) || !al16(dq) || !al16(dkv)) return MMAE_ERR_ARG;
    ModAttn p{};
    p.ws = ws; p.nsplit = mmae_modattn_bwd_nsplit(B);
    p.q = q; p.kv = kv; p.slot_row = slot_row; p.dout = dout; p.dq = dq; p.dkv = dkv;
    p.q_stride = q_stride; p.kv_stride = kv_stride; p.do_stride = do_stride; p.dq_stride = dq_stride; p.dkv_stride = dkv_stride;
    p.rows = (long)B * P; p.I = inner; p.ns = ns; p.P = P; p.B = B; p.shared_base = shared_base; p.scale = scale;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    dim3 grid(P, p.nsplit), blk(256);
    const size_t lds = (size_t)4 * 2 * inner * sizeof(float);
    const bool two = inner > 512;
#define GO(T, DHV, NCHV) MMAE_LAUNCH((modattn_bwd_kernel<T, DHV, NCHV>), grid, blk, lds, st, p)
    bool fast = false;
    if (inner == 512 && ns >= 2 && ns <= 5)
        fast = dtype == MMAE_BF16 ? (head_dim == 64 ? modattn_bwd_fast<bf16, 64>(p, grid, st) : modattn_bwd_fast<bf16, 32>(p, grid, st))
                                  : (head_dim == 64 ? modattn_bwd_fast<float, 64>(p, grid, st) : modattn_bwd_fast<float, 32>(p, grid, st));
    if (fast) {
    } else if (dtype == MMAE_BF16) {
        if (head_dim == 64) { if (two) GO(bf16, 64, 2); else GO(bf16, 64, 1); } else { if (two) GO(bf16, 32, 2); else GO(bf16, 32, 1); }
    } else {
        if (head_dim == 64) { if (two) GO(float, 64, 2); else GO(float, 64, 1); } else { if (two) GO(float, 32, 2); else GO(float, 32, 1); }
    }
#undef GO
    MMAE_CHECK_LAUNCH();
    const long n = (long)P * 2 * inner;
    if (dtype == MMAE_BF16) MMAE_LAUNCH((modattn_bwd_finish_kernel<bf16>), dim3(cdiv(n, 256)), dim3(256), 0, st, p);
    else MMAE_LAUNCH((modattn_bwd_finish_kernel<float>), dim3(cdiv(n, 256)), dim3(256), 0, st, p);
    MMAE_CHECK_LAUNCH();
    return MMAE_OK;
}
