// Segment-masked ("Zorro") multi-head attention, forward and backward, for gfx950.
//
// Replaces the ATen composition in the reference's Attention.forward
// (downstream/instance_segmentation/modeling/multimae/zorro_utils.py:181-193: split heads, q*scale, einsum QK^T,
// masked_fill(~mask, -finfo.max), softmax, einsum PV, merge heads) and the decoder attention core
// (pretraining/multimae/multimae_utils.py:172-179), without materialising the (B,h,S,S) score tensor.
//
// Token layout ("packed token mask"): every sample b owns `nseg` segments of query rows and `nseg` segments of key
// rows; segment s of sample b is the contiguous row range [start[b][s], start[b][s] + len[b][s]) of the q / kv
// matrices.  Segment index == token type (S1, S2, DEM, ..., last = FUSION).  The Zorro rule
// (multimae_crossattn.py:441-447, :489-493) becomes:
//     a query in segment s <  nseg-1 attends the keys of key-segment s only,
//     a query in segment s == nseg-1 attends every key of its sample.
// A query whose key-segment is empty has a fully masked row in the reference: masked_fill with a FINITE value makes
// the softmax uniform over ALL keys of the sample (empty_mode 0).  empty_mode 1 instead returns zeros, which is what
// the reference's un-masked cross attention over an empty context produces (multimae_crossattn.py:530-543 with N_m = 0).
// Lengths may differ per sample (modality dropout); no score outside the allowed segments is ever computed.
//
// Kernels (T = bf16 or f32, DH = 32 or 64; 256 threads = 4 waves; MFMA 16x16 tiles, see common.hpp mma16):
//   mha_fwd   : block = 64 query rows of one segment x one head; wave = 16 query rows.  Computes S^T = K Q^T so that the
//               query sits on the lane: softmax statistics are lane-local + 2 cross-lane steps, and the P^T accumulator
//               is directly the B operand of O^T = V^T P^T (no LDS round trip for P).  K row-major and V transposed are
//               staged in LDS once per 64-key tile and shared by the 4 waves.  Online softmax, fp32 accumulate.
//   mha_bwd_dq: same decomposition; recomputes P from LSE, dS^T = P^T o (dP^T - delta) and dQ^T = K^T dS^T.
//               Also produces delta = rowsum(dO o O) for the second kernel.
//   mha_bwd_dkdv: block = 64 keys of one key segment x one head; wave = 16 keys held on the lane ("key on the lane"),
//               sweeping the query tiles that may attend them; dK^T / dV^T accumulate in registers, no atomics.
#include "common.hpp"
#include "mmae_hip.h"
#include "mmae_internal.h"

#include "mha_common.hpp"

template <typename T> struct Vec4;
template <> struct Vec4<float> { typedef f32x4 type; };
template <> struct Vec4<bf16> { typedef bf16x4 type; };

template <typename T> __device__ __forceinline__ typename Vec8<T>::type lds_ld8(const T* p);
template <> __device__ __forceinline__ bf16x8 lds_ld8<bf16>(const bf16* p) { return *reinterpret_cast<const bf16x8*>(p); }
template <> __device__ __forceinline__ f32x8 lds_ld8<float>(const float* p) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p);
    const f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
    f32x8 r;
    r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3]; r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
    return r;
}
template <typename T> __device__ __forceinline__ void lds_st8(T* p, const typename Vec8<T>::type& v);
template <> __device__ __forceinline__ void lds_st8<bf16>(bf16* p, const bf16x8& v) { *reinterpret_cast<bf16x8*>(p) = v; }
template <> __device__ __forceinline__ void lds_st8<float>(float* p, const f32x8& v) {
    f32x4 a, b;
    a[0] = v[0]; a[1] = v[1]; a[2] = v[2]; a[3] = v[3]; b[0] = v[4]; b[1] = v[5]; b[2] = v[6]; b[3] = v[7];
    *reinterpret_cast<f32x4*>(p) = a;
    *reinterpret_cast<f32x4*>(p + 4) = b;
}
// global 8-element load that only assumes 16-byte alignment
template <typename T> __device__ __forceinline__ typename Vec8<T>::type g_ld8(const T* p) { return lds_ld8<T>(p); }

// Fragment of 8 k-slots taken from a transposed LDS image X^T[row][col]: slots j<4 -> cols c0+j, j>=4 -> cols c0+16+(j-4).
// (matches the accumulator-as-B-operand slot order used below: k-slot (g, j) <-> index 32*ks2 + 16*(j>>2) + 4*g + (j&3))
template <typename T> __device__ __forceinline__ typename Vec8<T>::type lds_ld4x2(const T* row, int c0) {
    typedef typename Vec4<T>::type V4;
    const V4 a = *reinterpret_cast<const V4*>(row + c0);
    const V4 b = *reinterpret_cast<const V4*>(row + c0 + 16);
    typename Vec8<T>::type r;
    r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3]; r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
    return r;
}
template <typename T> __device__ __forceinline__ typename Vec8<T>::type acc_pair_to_frag(const f32x4& lo, const f32x4& hi) {
    typename Vec8<T>::type r;
#pragma unroll
    for (int j = 0; j < 4; ++j) { r[j] = from_f<T>(lo[j]); r[4 + j] = from_f<T>(hi[j]); }
    return r;
}
template <typename T> __device__ __forceinline__ void store4(T* p, const f32x4& v) {
    typename Vec4<T>::type o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = from_f<T>(v[j]);
    *reinterpret_cast<typename Vec4<T>::type*>(p) = o;
}

// Stage a [rows<=64][DH] tile of a strided global matrix into LDS: row-major image RM[64][DH+8] and/or transposed image
// TR[DH][72].  Rows >= nvalid are zero-filled.  All 256 threads take part.
template <typename T, int DH, bool ROWMAJOR, bool TRANSPOSED>
__device__ __forceinline__ void stage_tile(const T* __restrict__ src, long row0, long stride, int col0, int nvalid,
                                           T* RM, T* TR, int tid) {
    constexpr int KP = DH + 8, VP = 72, CPR = DH / 8;
    typedef typename Vec8<T>::type V8;
    for (int c = tid; c < 64 * CPR; c += 256) {
        const int r = c / CPR, dc = c % CPR;
        V8 v = zero8<T>();
        if (r < nvalid) v = g_ld8<T>(src + (row0 + r) * stride + col0 + dc * 8);
        if (ROWMAJOR) lds_st8<T>(RM + r * KP + dc * 8, v);
        if (TRANSPOSED) {
#pragma unroll
            for (int e = 0; e < 8; ++e) TR[(dc * 8 + e) * VP + r] = v[e];
        }
    }
}

// ------------------------------------------------------------------------------------------------------ forward
template <typename T, int DH>
__global__ __launch_bounds__(256) void mha_fwd_kernel(MhaDesc p) {
    constexpr int KS = DH / 32, DT = DH / 16, KP = DH + 8, VP = 72;
    typedef typename Vec8<T>::type V8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* Ks = reinterpret_cast<T*>(smem_raw);
    T* Vt = Ks + 64 * KP;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, g = lane >> 4;
    const BlockSel bs = decode_block(blockIdx.x, p.max_tiles, p.B, p.H);
    if (bs.b < 0) return;
    const int b = bs.b, h = bs.h;
    const int* qlen = p.q_len + b * p.nseg; const int* qst = p.q_start + b * p.nseg;
    const int* klen = p.k_len + b * p.nseg; const int* kst = p.k_start + b * p.nseg;
    const TileSel ts = select_tile(qlen, p.nseg, bs.t);
    if (ts.seg < 0) return;
    const long qrow0 = (long)qst[ts.seg] + ts.t0;

    int ks_begin = 0, ks_end = 0; bool uniform = false;
    if (ts.seg == p.nseg - 1) { ks_begin = 0; ks_end = p.nseg; }
    else if (klen[ts.seg] > 0) { ks_begin = ts.seg; ks_end = ts.seg + 1; }
    else if (p.empty_mode == 0) { ks_begin = 0; ks_end = p.nseg; uniform = true; }

    const int myq = wave * 16 + lr;
    const bool qvalid = myq < ts.n;
    V8 qf[KS];
    {
        const T* qp = reinterpret_cast<const T*>(p.q) + (qrow0 + myq) * p.q_stride + h * DH + 8 * g;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = qvalid ? g_ld8<T>(qp + 32 * ks) : zero8<T>();
    }
    float m = -INFINITY, l = 0.f;
    f32x4 oacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

    const T* kg = reinterpret_cast<const T*>(p.k);
    const T* vg = reinterpret_cast<const T*>(p.v);
    for (int ksg = ks_begin; ksg < ks_end; ++ksg) {
        const int L = klen[ksg];
        const long krow = kst[ksg];
        for (int j0 = 0; j0 < L; j0 += 64) {
            const int kn = min(64, L - j0);
            __syncthreads();
            stage_tile<T, DH, true, false>(kg, krow + j0, p.k_stride, h * DH, kn, Ks, nullptr, tid);
            stage_tile<T, DH, false, true>(vg, krow + j0, p.v_stride, h * DH, kn, nullptr, Vt, tid);
            __syncthreads();
            f32x4 s[4];
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) {
                s[t4] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
                    s[t4] = mma16(lds_ld8<T>(Ks + (16 * t4 + lr) * KP + 32 * ks + 8 * g), qf[ks], s[t4]);
            }
            float mx = -INFINITY;
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = 16 * t4 + 4 * g + r;
                    float val = uniform ? 0.f : s[t4][r] * p.scale;
                    if (key >= kn) val = -INFINITY;
                    s[t4][r] = val;
                    mx = fmaxf(mx, val);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m, mx);           // finite: every tile holds >= 1 valid key
            const float alpha = __expf(m - m_new);      // m = -inf on the first tile -> 0
            float rs = 0.f;
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = __expf(s[t4][r] - m_new);
                    s[t4][r] = pv;
                    rs += pv;
                }
            rs += __shfl_xor(rs, 16);
            rs += __shfl_xor(rs, 32);
            l = l * alpha + rs;
            m = m_new;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) oacc[dt] *= alpha;
#pragma unroll
            for (int ks2 = 0; ks2 < 2; ++ks2) {
                const V8 pb = acc_pair_to_frag<T>(s[2 * ks2], s[2 * ks2 + 1]);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
                    oacc[dt] = mma16(lds_ld4x2<T>(Vt + (16 * dt + lr) * VP, 32 * ks2 + 4 * g), pb, oacc[dt]);
            }
        }
    }
    if (qvalid) {
        const float inv = l > 0.f ? 1.f / l : 0.f;
        T* op = reinterpret_cast<T*>(p.o) + (qrow0 + myq) * p.o_stride + h * DH + 4 * g;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) store4<T>(op + 16 * dt, oacc[dt] * inv);
        if (g == 0) p.lse[(long)h * p.stat_stride + qrow0 + myq] = l > 0.f ? m + __logf(l) : 0.f;
    }
}

// ------------------------------------------------------------------------------------------------------ backward: dQ (+ delta)
template <typename T, int DH>
__global__ __launch_bounds__(256) void mha_bwd_dq_kernel(MhaDesc p) {
    constexpr int KS = DH / 32, DT = DH / 16, KP = DH + 8, VP = 72;
    typedef typename Vec8<T>::type V8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* Ks = reinterpret_cast<T*>(smem_raw);
    T* Vs = Ks + 64 * KP;
    T* Kt = Vs + 64 * KP;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, g = lane >> 4;
    const BlockSel bs = decode_block(blockIdx.x, p.max_tiles, p.B, p.H);
    if (bs.b < 0) return;
    const int b = bs.b, h = bs.h;
    const int* qlen = p.q_len + b * p.nseg; const int* qst = p.q_start + b * p.nseg;
    const int* klen = p.k_len + b * p.nseg; const int* kst = p.k_start + b * p.nseg;
    const TileSel ts = select_tile(qlen, p.nseg, bs.t);
    if (ts.seg < 0) return;
    const long qrow0 = (long)qst[ts.seg] + ts.t0;

    int ks_begin = 0, ks_end = 0; bool uniform = false;
    if (ts.seg == p.nseg - 1) { ks_begin = 0; ks_end = p.nseg; }
    else if (klen[ts.seg] > 0) { ks_begin = ts.seg; ks_end = ts.seg + 1; }
    else if (p.empty_mode == 0) { ks_begin = 0; ks_end = p.nseg; uniform = true; }

    const int myq = wave * 16 + lr;
    const bool qvalid = myq < ts.n;
    V8 qf[KS], dof[KS];
    float dpart = 0.f;
    {
        const T* qp = reinterpret_cast<const T*>(p.q) + (qrow0 + myq) * p.q_stride + h * DH + 8 * g;
        const T* dop = reinterpret_cast<const T*>(p.dout) + (qrow0 + myq) * p.do_stride + h * DH + 8 * g;
        const T* op = reinterpret_cast<const T*>(p.o) + (qrow0 + myq) * p.o_stride + h * DH + 8 * g;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qf[ks] = qvalid ? g_ld8<T>(qp + 32 * ks) : zero8<T>();
            dof[ks] = qvalid ? g_ld8<T>(dop + 32 * ks) : zero8<T>();
            const V8 of = qvalid ? g_ld8<T>(op + 32 * ks) : zero8<T>();
#pragma unroll
            for (int j = 0; j < 8; ++j) dpart += to_f(dof[ks][j]) * to_f(of[j]);
        }
    }
    dpart += __shfl_xor(dpart, 16);
    dpart += __shfl_xor(dpart, 32);
    const float delta = dpart;
    const float lse = qvalid ? p.lse[(long)h * p.stat_stride + qrow0 + myq] : 0.f;
    if (qvalid && g == 0) p.delta[(long)h * p.stat_stride + qrow0 + myq] = delta;

    f32x4 dqacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) dqacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

    const T* kg = reinterpret_cast<const T*>(p.k);
    const T* vg = reinterpret_cast<const T*>(p.v);
    for (int ksg = ks_begin; ksg < ks_end; ++ksg) {
        const int L = klen[ksg];
        const long krow = kst[ksg];
        for (int j0 = 0; j0 < L; j0 += 64) {
            const int kn = min(64, L - j0);
            if (uniform) continue;   // dS == 0 for a uniform (fully masked) row: nothing flows to q (block-uniform)
            __syncthreads();
            stage_tile<T, DH, true, true>(kg, krow + j0, p.k_stride, h * DH, kn, Ks, Kt, tid);
            stage_tile<T, DH, true, false>(vg, krow + j0, p.v_stride, h * DH, kn, Vs, nullptr, tid);
            __syncthreads();
            f32x4 s[4], dp[4];
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) {
                s[t4] = f32x4{0.f, 0.f, 0.f, 0.f};
                dp[t4] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    s[t4] = mma16(lds_ld8<T>(Ks + (16 * t4 + lr) * KP + 32 * ks + 8 * g), qf[ks], s[t4]);
                    dp[t4] = mma16(lds_ld8<T>(Vs + (16 * t4 + lr) * KP + 32 * ks + 8 * g), dof[ks], dp[t4]);
                }
            }
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = 16 * t4 + 4 * g + r;
                    const float pv = key < kn ? __expf(s[t4][r] * p.scale - lse) : 0.f;
                    s[t4][r] = pv * (dp[t4][r] - delta) * p.scale;     // dS^T
                }
#pragma unroll
            for (int ks2 = 0; ks2 < 2; ++ks2) {
                const V8 dsb = acc_pair_to_frag<T>(s[2 * ks2], s[2 * ks2 + 1]);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
                    dqacc[dt] = mma16(lds_ld4x2<T>(Kt + (16 * dt + lr) * VP, 32 * ks2 + 4 * g), dsb, dqacc[dt]);
            }
        }
    }
    if (qvalid) {
        T* dqp = reinterpret_cast<T*>(p.dq) + (qrow0 + myq) * p.dq_stride + h * DH + 4 * g;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) store4<T>(dqp + 16 * dt, dqacc[dt]);
    }
}

// ------------------------------------------------------------------------------------------------------ backward: dK, dV
template <typename T, int DH>
__global__ __launch_bounds__(256) void mha_bwd_dkdv_kernel(MhaDesc p) {
    constexpr int KS = DH / 32, DT = DH / 16, KP = DH + 8, VP = 72;
    typedef typename Vec8<T>::type V8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* Qs = reinterpret_cast<T*>(smem_raw);
    T* dOs = Qs + 64 * KP;
    T* Qt = dOs + 64 * KP;
    T* dOt = Qt + DH * VP;
    float* lse_s = reinterpret_cast<float*>(dOt + DH * VP);
    float* delta_s = lse_s + 64;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, g = lane >> 4;
    const BlockSel bs = decode_block(blockIdx.x, p.max_tiles, p.B, p.H);
    if (bs.b < 0) return;
    const int b = bs.b, h = bs.h;
    const int* qlen = p.q_len + b * p.nseg; const int* qst = p.q_start + b * p.nseg;
    const int* klen = p.k_len + b * p.nseg; const int* kst = p.k_start + b * p.nseg;
    const TileSel ts = select_tile(klen, p.nseg, bs.t);
    if (ts.seg < 0) return;
    const long krow0 = (long)kst[ts.seg] + ts.t0;

    const int mykey = wave * 16 + lr;
    const bool kvalid = mykey < ts.n;
    V8 kf[KS], vf[KS];
    {
        const T* kp = reinterpret_cast<const T*>(p.k) + (krow0 + mykey) * p.k_stride + h * DH + 8 * g;
        const T* vp = reinterpret_cast<const T*>(p.v) + (krow0 + mykey) * p.v_stride + h * DH + 8 * g;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            kf[ks] = kvalid ? g_ld8<T>(kp + 32 * ks) : zero8<T>();
            vf[ks] = kvalid ? g_ld8<T>(vp + 32 * ks) : zero8<T>();
        }
    }
    f32x4 dkacc[DT], dvacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) { dkacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dvacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    const T* qg = reinterpret_cast<const T*>(p.q);
    const T* dog = reinterpret_cast<const T*>(p.dout);
    for (int sq = 0; sq < p.nseg; ++sq) {
        int mode = 0;                                                 // 0 skip, 1 normal, 2 uniform row
        if (sq == p.nseg - 1 || sq == ts.seg) mode = 1;
        else if (klen[sq] == 0 && p.empty_mode == 0) mode = 2;
        if (mode == 0) continue;
        const int QL = qlen[sq];
        const long qrow = qst[sq];
        for (int q0 = 0; q0 < QL; q0 += 64) {
            const int qn = min(64, QL - q0);
            __syncthreads();
            stage_tile<T, DH, true, true>(qg, qrow + q0, p.q_stride, h * DH, qn, Qs, Qt, tid);
            stage_tile<T, DH, true, true>(dog, qrow + q0, p.do_stride, h * DH, qn, dOs, dOt, tid);
            if (tid < 64) {
                const bool v = tid < qn;
                lse_s[tid] = v ? p.lse[(long)h * p.stat_stride + qrow + q0 + tid] : 0.f;
                delta_s[tid] = v ? p.delta[(long)h * p.stat_stride + qrow + q0 + tid] : 0.f;
            }
            __syncthreads();
            f32x4 s[4], dp[4];
#pragma unroll
            for (int qt = 0; qt < 4; ++qt) {
                s[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
                dp[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    s[qt] = mma16(lds_ld8<T>(Qs + (16 * qt + lr) * KP + 32 * ks + 8 * g), kf[ks], s[qt]);
                    dp[qt] = mma16(lds_ld8<T>(dOs + (16 * qt + lr) * KP + 32 * ks + 8 * g), vf[ks], dp[qt]);
                }
            }
#pragma unroll
            for (int qt = 0; qt < 4; ++qt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int qi = 16 * qt + 4 * g + r;
                    const bool valid = kvalid && qi < qn;
                    float pv = 0.f, ds = 0.f;
                    if (valid) {
                        if (mode == 2) { pv = __expf(-lse_s[qi]); }
                        else { pv = __expf(s[qt][r] * p.scale - lse_s[qi]); ds = pv * (dp[qt][r] - delta_s[qi]) * p.scale; }
                    }
                    s[qt][r] = pv;
                    dp[qt][r] = ds;
                }
#pragma unroll
            for (int ks2 = 0; ks2 < 2; ++ks2) {
                const V8 pb = acc_pair_to_frag<T>(s[2 * ks2], s[2 * ks2 + 1]);
                const V8 dsb = acc_pair_to_frag<T>(dp[2 * ks2], dp[2 * ks2 + 1]);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    dvacc[dt] = mma16(lds_ld4x2<T>(dOt + (16 * dt + lr) * VP, 32 * ks2 + 4 * g), pb, dvacc[dt]);
                    dkacc[dt] = mma16(lds_ld4x2<T>(Qt + (16 * dt + lr) * VP, 32 * ks2 + 4 * g), dsb, dkacc[dt]);
                }
            }
        }
    }
    if (kvalid) {
        T* dkp = reinterpret_cast<T*>(p.dk) + (krow0 + mykey) * p.dk_stride + h * DH + 4 * g;
        T* dvp = reinterpret_cast<T*>(p.dv) + (krow0 + mykey) * p.dv_stride + h * DH + 4 * g;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) { store4<T>(dkp + 16 * dt, dkacc[dt]); store4<T>(dvp + 16 * dt, dvacc[dt]); }
    }
}

// ------------------------------------------------------------------------------------------------------ host side
template <typename T, int DH> static size_t fwd_lds() { return (size_t)(64 * (DH + 8) + DH * 72) * sizeof(T); }
template <typename T, int DH> static size_t dq_lds() { return (size_t)(2 * 64 * (DH + 8) + DH * 72) * sizeof(T); }
template <typename T, int DH> static size_t dkdv_lds() { return (size_t)(2 * 64 * (DH + 8) + 2 * DH * 72) * sizeof(T) + 128 * sizeof(float); }

template <typename K> static int set_lds(K kern, size_t bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess
               ? MMAE_OK : MMAE_ERR_LAUNCH;
}

template <typename T, int DH> static int launch_fwd(const MhaDesc& d, hipStream_t st) {
    const size_t lds = fwd_lds<T, DH>();
    if (set_lds(mha_fwd_kernel<T, DH>, lds)) return MMAE_ERR_LAUNCH;
    dim3 grid(xcd_grid(d.B, d.H, d.max_tiles));
    MMAE_LAUNCH((mha_fwd_kernel<T, DH>), grid, dim3(256), lds, st, d);
    MMAE_CHECK_LAUNCH();
    return MMAE_OK;
}
template <typename T, int DH> static int launch_bwd(MhaDesc d, int max_q_tiles, int max_k_tiles, hipStream_t st) {
    size_t lds = dq_lds<T, DH>();
    if (set_lds(mha_bwd_dq_kernel<T, DH>, lds)) return MMAE_ERR_LAUNCH;
    d.max_tiles = max_q_tiles;
    MMAE_LAUNCH((mha_bwd_dq_kernel<T, DH>), dim3(xcd_grid(d.B, d.H, max_q_tiles)), dim3(256), lds, st, d);
    MMAE_CHECK_LAUNCH();
    lds = dkdv_lds<T, DH>();
    if (set_lds(mha_bwd_dkdv_kernel<T, DH>, lds)) return MMAE_ERR_LAUNCH;
    d.max_tiles = max_k_tiles;
    MMAE_LAUNCH((mha_bwd_dkdv_kernel<T, DH>), dim3(xcd_grid(d.B, d.H, max_k_tiles)), dim3(256), lds, st, d);
    MMAE_CHECK_LAUNCH();
    return MMAE_OK;
}

static bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static int check_common(int dtype, int head_dim, int B, int H, int nseg, const void* q, const void* k, const void* v,
                        long qs, long ks, long vs, const int* a, const int* b2, const int* c, const int* d) {
    if (dtype != MMAE_F32 && dtype != MMAE_BF16) return MMAE_ERR_ARG;
    if (head_dim != 32 && head_dim != 64) return MMAE_ERR_ARG;
    if (B <= 0 || H <= 0 || nseg <= 0 || nseg > MAXSEG) return MMAE_ERR_ARG;
    if (!q || !k || !v || !a || !b2 || !c || !d) return MMAE_ERR_ARG;
    if (!al16(q) || !al16(k) || !al16(v)) return MMAE_ERR_ARG;
    if ((qs % 8) || (ks % 8) || (vs % 8)) return MMAE_ERR_ARG;
    return MMAE_OK;
}

extern "C" int mmae_mha_fwd_variant(int dtype, int head_dim, int B, int H, int nseg, const void* q, const void* k, const void* v,
                            void* out, float* lse, long q_stride, long k_stride, long v_stride, long o_stride,
                            long q_rows_total, const int* q_seg_start, const int* q_seg_len, const int* k_seg_start,
                            const int* k_seg_len, int max_q_rows, int max_k_rows, float scale, int empty_mode, int variant,
                            void* stream) {
    int rc = check_common(dtype, head_dim, B, H, nseg, q, k, v, q_stride, k_stride, v_stride, q_seg_start, q_seg_len,
                          k_seg_start, k_seg_len);
    if (rc) return rc;
    if (!out || !lse || !al16(out) || (o_stride % 8) || max_q_rows < 0 || max_k_rows < 0 || q_rows_total <= 0) return MMAE_ERR_ARG;
    MhaDesc d{};
    d.q = q; d.k = k; d.v = v; d.o = out; d.lse = lse;
    d.q_stride = q_stride; d.k_stride = k_stride; d.v_stride = v_stride; d.o_stride = o_stride;
    d.q_start = q_seg_start; d.q_len = q_seg_len; d.k_start = k_seg_start; d.k_len = k_seg_len;
    d.stat_stride = q_rows_total; d.B = B; d.H = H; d.nseg = nseg; d.max_tiles = max_q_rows / 64 + nseg; d.max_q_rows = max_q_rows; d.max_k_rows = max_k_rows;
    d.scale = scale; d.empty_mode = empty_mode;
    if (variant > 0) { d.hpb_req = (variant >> 8) & 15; variant &= 255; }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MMAE_BF16) return variant < 0 ? (head_dim == 64 ? launch_fwd<bf16, 64>(d, st) : launch_fwd<bf16, 32>(d, st)) : mha_bf16_fwd(d, head_dim, variant, st);
    return head_dim == 64 ? launch_fwd<float, 64>(d, st) : launch_fwd<float, 32>(d, st);
}

extern "C" int mmae_mha_fwd(int dtype, int head_dim, int B, int H, int nseg, const void* q, const void* k, const void* v,
                            void* out, float* lse, long q_stride, long k_stride, long v_stride, long o_stride,
                            long q_rows_total, const int* q_seg_start, const int* q_seg_len, const int* k_seg_start,
                            const int* k_seg_len, int max_q_rows, int max_k_rows, float scale, int empty_mode, void* stream) {
    return mmae_mha_fwd_variant(dtype, head_dim, B, H, nseg, q, k, v, out, lse, q_stride, k_stride, v_stride, o_stride,
                                q_rows_total, q_seg_start, q_seg_len, k_seg_start, k_seg_len, max_q_rows, max_k_rows, scale,
                                empty_mode, 0, stream);
}

extern "C" int mmae_mha_fwd_route(int dtype, int head_dim, int nseg, int max_q_rows, int max_k_rows, long k_stride, long v_stride) {
    if ((dtype != MMAE_F32 && dtype != MMAE_BF16) || (head_dim != 32 && head_dim != 64) || nseg <= 0 || nseg > MAXSEG ||
        max_q_rows < 0 || max_k_rows < 0) return MMAE_ERR_ARG;
    if (dtype == MMAE_F32) return 0;
    MhaDesc d{};
    d.nseg = nseg; d.max_q_rows = max_q_rows; d.max_k_rows = max_k_rows; d.k_stride = k_stride; d.v_stride = v_stride;
    return (head_dim == 64 && mha_sh_applicable(d)) ? 2 : 1;   // the same test mha_bf16_fwd() routes the default variant by
}

extern "C" long mmae_mha_bwd_ws_floats(int H, long q_rows_total) {
    return (H <= 0 || q_rows_total <= 0) ? MMAE_ERR_ARG : 3L * H * q_rows_total;
}

// workspace of the fused dQ + dK + dV kernel: the three planes + one fp32 64 x 64 partial per (sample, query tile, head)
extern "C" long mmae_mha_bwd_fused_ws_floats(int B, int H, int nseg, long q_rows_total, int max_q_rows) {
    if (B <= 0 || H <= 0 || nseg <= 0 || q_rows_total <= 0 || max_q_rows < 0) return MMAE_ERR_ARG;
    return 3L * H * q_rows_total + (long)B * (max_q_rows / 64 + nseg) * H * 4096;
}

extern "C" int mmae_mha_bwd_fused_supported(int dtype, int head_dim, int B, int H, int nseg, int max_q_rows, int max_k_rows) {
    if (dtype != MMAE_BF16 || head_dim != 64 || B <= 0 || H <= 0 || nseg <= 0 || max_q_rows < 0 || max_k_rows < 0) return 0;
    MhaDesc d{};
    d.B = B; d.H = H; d.nseg = nseg; d.max_q_rows = max_q_rows; d.max_k_rows = max_k_rows;
    d.max_qt = max_q_rows / 64 + nseg;
    d.dq_ws = reinterpret_cast<float*>(16);          // (any non-null value: the predicate only tests that a workspace was given)
    return mha_sh_fused_supported(d) ? 1 : 0;
}

extern "C" int mmae_mha_bwd_variant(int dtype, int head_dim, int B, int H, int nseg, const void* q, const void* k, const void* v,
                            const void* out, const void* dout, const float* lse, float* delta_ws, void* dq, void* dk,
                            void* dv, long q_stride, long k_stride, long v_stride, long o_stride, long do_stride,
                            long dq_stride, long dk_stride, long dv_stride, long q_rows_total, const int* q_seg_start,
                            const int* q_seg_len, const int* k_seg_start, const int* k_seg_len, int max_q_rows,
                            int max_k_rows, float scale, int empty_mode, int variant, void* stream) {
    int rc = check_common(dtype, head_dim, B, H, nseg, q, k, v, q_stride, k_stride, v_stride, q_seg_start, q_seg_len,
                          k_seg_start, k_seg_len);
    if (rc) return rc;
    if (!out || !dout || !lse || !delta_ws || !dq || !dk || !dv) return MMAE_ERR_ARG;
    if (!al16(out) || !al16(dout) || !al16(dq) || !al16(dk) || !al16(dv)) return MMAE_ERR_ARG;
    if ((o_stride % 8) || (do_stride % 8) || (dq_stride % 8) || (dk_stride % 8) || (dv_stride % 8)) return MMAE_ERR_ARG;
    if (max_q_rows < 0 || max_k_rows < 0 || q_rows_total <= 0) return MMAE_ERR_ARG;
    MhaDesc d{};
    d.q = q; d.k = k; d.v = v; d.o = const_cast<void*>(out); d.dout = dout; d.dq = dq; d.dk = dk; d.dv = dv;
    d.lse = const_cast<float*>(lse); d.delta = delta_ws;
    d.q_stride = q_stride; d.k_stride = k_stride; d.v_stride = v_stride; d.o_stride = o_stride; d.do_stride = do_stride;
    d.dq_stride = dq_stride; d.dk_stride = dk_stride; d.dv_stride = dv_stride;
    d.q_start = q_seg_start; d.q_len = q_seg_len; d.k_start = k_seg_start; d.k_len = k_seg_len;
    d.stat_stride = q_rows_total; d.B = B; d.H = H; d.nseg = nseg; d.max_q_rows = max_q_rows; d.max_k_rows = max_k_rows;
    d.scale = scale; d.empty_mode = empty_mode;
    if (variant > 0) { d.hpb_req = (variant >> 8) & 15; variant &= 255; }
    const int mq = max_q_rows / 64 + nseg, mk = max_k_rows / 64 + nseg;
    if (variant >= 50 && variant <= 55) {          // fused backward (mha_sh_bwd_kernel): the workspace continues behind the three planes -- mmae_mha_bwd_fused_ws_floats
        d.dq_ws = delta_ws + 3L * H * q_rows_total;
        d.max_qt = mq;
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MMAE_BF16) return variant < 0 ? (head_dim == 64 ? launch_bwd<bf16, 64>(d, mq, mk, st) : launch_bwd<bf16, 32>(d, mq, mk, st)) : mha_bf16_bwd(d, head_dim, mq, mk, variant, st);
    return head_dim == 64 ? launch_bwd<float, 64>(d, mq, mk, st) : launch_bwd<float, 32>(d, mq, mk, st);
}

extern "C" int mmae_mha_bwd_fused(int dtype, int head_dim, int B, int H, int nseg, const void* q, const void* k, const void* v,
                                  const void* out, const void* dout, const float* lse, float* delta_ws, void* dq, void* dk,
                                  void* dv, long q_stride, long k_stride, long v_stride, long o_stride, long do_stride,
                                  long dq_stride, long dk_stride, long dv_stride, long q_rows_total, const int* q_seg_start,
                                  const int* q_seg_len, const int* k_seg_start, const int* k_seg_len, int max_q_rows,
                                  int max_k_rows, float scale, int empty_mode, void* stream) {
    if (!mmae_mha_bwd_fused_supported(dtype, head_dim, B, H, nseg, max_q_rows, max_k_rows)) return MMAE_ERR_ARG;
    return mmae_mha_bwd_variant(dtype, head_dim, B, H, nseg, q, k, v, out, dout, lse, delta_ws, dq, dk, dv, q_stride, k_stride,
                                v_stride, o_stride, do_stride, dq_stride, dk_stride, dv_stride, q_rows_total, q_seg_start,
                                q_seg_len, k_seg_start, k_seg_len, max_q_rows, max_k_rows, scale, empty_mode, 50, stream);
}

extern "C" int mmae_mha_bwd(int dtype, int head_dim, int B, int H, int nseg, const void* q, const void* k, const void* v,
                            const void* out, const void* dout, const float* lse, float* delta_ws, void* dq, void* dk,
                            void* dv, long q_stride, long k_stride, long v_stride, long o_stride, long do_stride,
                            long dq_stride, long dk_stride, long dv_stride, long q_rows_total, const int* q_seg_start,
                            const int* q_seg_len, const int* k_seg_start, const int* k_seg_len, int max_q_rows,
                            int max_k_rows, float scale, int empty_mode, void* stream) {
    return mmae_mha_bwd_variant(dtype, head_dim, B, H, nseg, q, k, v, out, dout, lse, delta_ws, dq, dk, dv, q_stride, k_stride,
                                v_stride, o_stride, do_stride, dq_stride, dk_stride, dv_stride, q_rows_total, q_seg_start,
                                q_seg_len, k_seg_start, k_seg_len, max_q_rows, max_k_rows, scale, empty_mode, 0, stream);
}
