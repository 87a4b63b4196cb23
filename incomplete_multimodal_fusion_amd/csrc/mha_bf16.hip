// bf16 fast path of the segment-masked attention (gfx950).  Same decomposition and semantics as mha.hip (read its header
// for the segment rule); what differs is how operands reach the matrix cores:
//   * K, V (forward / dQ) and Q, dO (dK/dV) tiles are staged ROW-MAJOR only, 16 B per lane, into LDS images with a
//     +8-element pitch (b128 row reads of 16 different rows are bank-conflict free);
//   * every operand that must be contracted over the tile's ROW index (V^T for O^T = V^T P^T, K^T for dQ^T = K^T dS^T,
//     dO^T / Q^T for dV^T, dK^T) is fetched with ds_read_b64_tr_b16 -- the hardware transpose read -- from the same
//     row-major image: no transposed copy, no 2-byte scatter writes.  Semantics probed on MI355X
//     (tools/probes/tr_read_probe.hip): in each 16-lane group, lane 4q+p supplies the address of block row q,
//     columns 4p..4p+3, and lane i receives column i of the 4 rows.
//   * the next tile is prefetched into registers while the current one is computed (one LDS buffer, two barriers per tile);
//   * softmax in the exp2 domain with one fma per score, and the key-bound mask only on a segment's ragged last tile.
#include "mha_common.hpp"
#include "mmae_hip.h"

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
#define LDS_AS __attribute__((address_space(3)))
#define MAXT 80   // tiles one block may sweep (keys of a sample / 64 + segments)

#define LOG2E 1.4426950408889634f
#define LN2 0.6931471805599453f

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// 8 k-slots of the column (col0 + lane&15) of a row-major LDS image: slots j<4 -> rows row0 + 4g + j,
// j>=4 -> rows row0 + 16 + 4g + (j-4)   (g = lane>>4).  Matches acc_pair_to_frag()'s slot order.
__device__ __forceinline__ bf16x8 tr_frag(const bf16* img, int pitch, int row0, int col0, int lane) {
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const bf16* a0 = img + (row0 + 4 * g + q) * pitch + col0 + 4 * p;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(a0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(a0 + 16 * pitch));
    s16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3]; r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return __builtin_bit_cast(bf16x8, r);
}
__device__ __forceinline__ bf16x8 ld8(const bf16* p) { return *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ bf16x8 z8() { return __builtin_bit_cast(bf16x8, s16x8{0, 0, 0, 0, 0, 0, 0, 0}); }
__device__ __forceinline__ bf16x8 pack8(const f32x4& lo, const f32x4& hi) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) { r[j] = (bf16)lo[j]; r[4 + j] = (bf16)hi[j]; }
    return r;
}
__device__ __forceinline__ void st4(bf16* p, const f32x4& v) {
    bf16x4 o; o[0] = (bf16)v[0]; o[1] = (bf16)v[1]; o[2] = (bf16)v[2]; o[3] = (bf16)v[3];
    *reinterpret_cast<bf16x4*>(p) = o;
}

template <int DH> struct Geo {
    static constexpr int KS = DH / 32, DT = DH / 16, KP = DH + 8, CPR = DH / 8, NCH = 64 * CPR / 256;
};

// global -> registers: this thread's chunks of a [<=64 rows][DH] tile (rows >= n zero-filled)
template <int DH>
__device__ __forceinline__ void tile_load(const bf16* src, long row0, long stride, int col0, int n, int tid,
                                          bf16x8 (&reg)[Geo<DH>::NCH]) {
#pragma unroll
    for (int i = 0; i < Geo<DH>::NCH; ++i) {
        const int c = tid + 256 * i, r = c / Geo<DH>::CPR, dc = c % Geo<DH>::CPR;
        reg[i] = r < n ? ld8(src + (row0 + r) * stride + col0 + dc * 8) : z8();
    }
}
template <int DH>
__device__ __forceinline__ void tile_store(bf16* img, int tid, const bf16x8 (&reg)[Geo<DH>::NCH]) {
#pragma unroll
    for (int i = 0; i < Geo<DH>::NCH; ++i) {
        const int c = tid + 256 * i, r = c / Geo<DH>::CPR, dc = c % Geo<DH>::CPR;
        *reinterpret_cast<bf16x8*>(img + r * Geo<DH>::KP + dc * 8) = reg[i];
    }
}

// key tiles a query tile sweeps: filled by thread 0, read by everyone after a barrier
struct KeyPlan { int begin, end; bool uniform; };
__device__ __forceinline__ KeyPlan key_plan(int seg, int nseg, const int* klen, int empty_mode) {
    KeyPlan k; k.begin = 0; k.end = 0; k.uniform = false;
    if (seg == nseg - 1) { k.begin = 0; k.end = nseg; }
    else if (klen[seg] > 0) { k.begin = seg; k.end = seg + 1; }
    else if (empty_mode == 0) { k.begin = 0; k.end = nseg; k.uniform = true; }
    return k;
}

// ------------------------------------------------------------------------------------------------------ forward
template <int DH>
__global__ __launch_bounds__(256) void mha_bf16_fwd_kernel(MhaDesc p) {
    typedef Geo<DH> G;
    __shared__ __attribute__((aligned(16))) bf16 Ks[64 * G::KP];
    __shared__ __attribute__((aligned(16))) bf16 Vs[64 * G::KP];
    __shared__ int tl_row[MAXT], tl_n[MAXT], tl_cnt;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, g = lane >> 4;
    const BlockSel bs = decode_block(blockIdx.x, p.max_tiles, p.B, p.H);
    if (bs.b < 0) return;
    const int b = bs.b, h = bs.h;
    const int* qlen = p.q_len + b * p.nseg; const int* qst = p.q_start + b * p.nseg;
    const int* klen = p.k_len + b * p.nseg; const int* kst = p.k_start + b * p.nseg;
    const TileSel ts = select_tile(qlen, p.nseg, bs.t);
    if (ts.seg < 0) return;
    const long qrow0 = (long)qst[ts.seg] + ts.t0;
    const KeyPlan kp = key_plan(ts.seg, p.nseg, klen, p.empty_mode);
    if (tid == 0) {
        int n = 0;
        for (int s = kp.begin; s < kp.end; ++s)
            for (int j0 = 0; j0 < klen[s] && n < MAXT; j0 += 64) { tl_row[n] = kst[s] + j0; tl_n[n] = min(64, klen[s] - j0); ++n; }
        tl_cnt = n;
    }
    const int myq = wave * 16 + lr;
    const bool qvalid = myq < ts.n;
    bf16x8 qf[G::KS];
    {
        const bf16* qp = reinterpret_cast<const bf16*>(p.q) + (qrow0 + myq) * p.q_stride + h * DH + 8 * g;
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) qf[ks] = qvalid ? ld8(qp + 32 * ks) : z8();
    }
    __syncthreads();
    const int ntile = tl_cnt;
    const float c = kp.uniform ? 0.f : p.scale * LOG2E;       // scores enter the exp2 domain through one fma
    float m = -INFINITY, l = 0.f;                             // m is kept in the scaled (log2) domain
    f32x4 oacc[G::DT];
#pragma unroll
    for (int dt = 0; dt < G::DT; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

    const bf16* kg = reinterpret_cast<const bf16*>(p.k);
    const bf16* vg = reinterpret_cast<const bf16*>(p.v);
    bf16x8 kreg[G::NCH], vreg[G::NCH];
    if (ntile > 0) {
        tile_load<DH>(kg, tl_row[0], p.k_stride, h * DH, tl_n[0], tid, kreg);
        tile_load<DH>(vg, tl_row[0], p.v_stride, h * DH, tl_n[0], tid, vreg);
    }
    for (int t = 0; t < ntile; ++t) {
        __syncthreads();
        tile_store<DH>(Ks, tid, kreg);
        tile_store<DH>(Vs, tid, vreg);
        __syncthreads();
        const int kn = tl_n[t];
        if (t + 1 < ntile) {
            tile_load<DH>(kg, tl_row[t + 1], p.k_stride, h * DH, tl_n[t + 1], tid, kreg);
            tile_load<DH>(vg, tl_row[t + 1], p.v_stride, h * DH, tl_n[t + 1], tid, vreg);
        }
        f32x4 s[4];
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
            s[t4] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks)
                s[t4] = mma16(ld8(Ks + (16 * t4 + lr) * G::KP + 32 * ks + 8 * g), qf[ks], s[t4]);
        }
        if (kn < 64) {                                        // ragged last tile of a segment (block-uniform branch)
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (16 * t4 + 4 * g + r >= kn) s[t4][r] = -INFINITY;
        }
        float mx = fmaxf(fmaxf(s[0][0], s[0][1]), fmaxf(s[0][2], s[0][3]));
#pragma unroll
        for (int t4 = 1; t4 < 4; ++t4) mx = fmaxf(mx, fmaxf(fmaxf(s[t4][0], s[t4][1]), fmaxf(s[t4][2], s[t4][3])));
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        // c >= 0: max(c*s) = c*max(s); uniform rows (c == 0): all valid scores are 0 (mx may be -inf only if kn == 0)
        const float m_new = fmaxf(m, kp.uniform ? 0.f : mx * c);
        const float alpha = fast_exp2(m - m_new);
        float rs = 0.f;
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // masked keys hold -inf: fma(-inf, c>0, x) = -inf -> 0; for uniform rows mask explicitly
                float e = fast_exp2(__builtin_fmaf(s[t4][r], c, -m_new));
                if (kp.uniform) e = (16 * t4 + 4 * g + r < kn) ? 1.f : 0.f;
                s[t4][r] = e;
                rs += e;
            }
        rs += __shfl_xor(rs, 16);
        rs += __shfl_xor(rs, 32);
        l = l * alpha + rs;
        m = m_new;
#pragma unroll
        for (int dt = 0; dt < G::DT; ++dt) oacc[dt] *= alpha;
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2) {
            const bf16x8 pb = pack8(s[2 * ks2], s[2 * ks2 + 1]);
#pragma unroll
            for (int dt = 0; dt < G::DT; ++dt)
                oacc[dt] = mma16(tr_frag(Vs, G::KP, 32 * ks2, 16 * dt, lane), pb, oacc[dt]);
        }
    }
    if (qvalid) {
        const float inv = l > 0.f ? 1.f / l : 0.f;
        bf16* op = reinterpret_cast<bf16*>(p.o) + (qrow0 + myq) * p.o_stride + h * DH + 4 * g;
#pragma unroll
        for (int dt = 0; dt < G::DT; ++dt) st4(op + 16 * dt, oacc[dt] * inv);
        if (g == 0) p.lse[(long)h * p.stat_stride + qrow0 + myq] = l > 0.f ? m * LN2 + __logf(l) : 0.f;
    }
}

// ------------------------------------------------------------------------------------------------------ backward: dQ (+ delta)
template <int DH>
__global__ __launch_bounds__(256) void mha_bf16_bwd_dq_kernel(MhaDesc p) {
    typedef Geo<DH> G;
    __shared__ __attribute__((aligned(16))) bf16 Ks[64 * G::KP];
    __shared__ __attribute__((aligned(16))) bf16 Vs[64 * G::KP];
    __shared__ int tl_row[MAXT], tl_n[MAXT], tl_cnt;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, g = lane >> 4;
    const BlockSel bs = decode_block(blockIdx.x, p.max_tiles, p.B, p.H);
    if (bs.b < 0) return;
    const int b = bs.b, h = bs.h;
    const int* qlen = p.q_len + b * p.nseg; const int* qst = p.q_start + b * p.nseg;
    const int* klen = p.k_len + b * p.nseg; const int* kst = p.k_start + b * p.nseg;
    const TileSel ts = select_tile(qlen, p.nseg, bs.t);
    if (ts.seg < 0) return;
    const long qrow0 = (long)qst[ts.seg] + ts.t0;
    const KeyPlan kp = key_plan(ts.seg, p.nseg, klen, p.empty_mode);
    if (tid == 0) {
        int n = 0;
        if (!kp.uniform)          // dS == 0 for a uniform (fully masked) row: nothing flows to q
            for (int s = kp.begin; s < kp.end; ++s)
                for (int j0 = 0; j0 < klen[s] && n < MAXT; j0 += 64) { tl_row[n] = kst[s] + j0; tl_n[n] = min(64, klen[s] - j0); ++n; }
        tl_cnt = n;
    }
    const int myq = wave * 16 + lr;
    const bool qvalid = myq < ts.n;
    bf16x8 qf[G::KS], dof[G::KS];
    float dpart = 0.f;
    {
        const bf16* qp = reinterpret_cast<const bf16*>(p.q) + (qrow0 + myq) * p.q_stride + h * DH + 8 * g;
        const bf16* dop = reinterpret_cast<const bf16*>(p.dout) + (qrow0 + myq) * p.do_stride + h * DH + 8 * g;
        const bf16* op = reinterpret_cast<const bf16*>(p.o) + (qrow0 + myq) * p.o_stride + h * DH + 8 * g;
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) {
            qf[ks] = qvalid ? ld8(qp + 32 * ks) : z8();
            dof[ks] = qvalid ? ld8(dop + 32 * ks) : z8();
            const bf16x8 of = qvalid ? ld8(op + 32 * ks) : z8();
#pragma unroll
            for (int j = 0; j < 8; ++j) dpart += (float)dof[ks][j] * (float)of[j];
        }
    }
    dpart += __shfl_xor(dpart, 16);
    dpart += __shfl_xor(dpart, 32);
    const float delta = dpart;
    const float lse2 = (qvalid ? p.lse[(long)h * p.stat_stride + qrow0 + myq] : 0.f) * LOG2E;
    if (qvalid && g == 0) p.delta[(long)h * p.stat_stride + qrow0 + myq] = delta;
    __syncthreads();
    const int ntile = tl_cnt;
    const float c = p.scale * LOG2E;

    f32x4 dqacc[G::DT];
#pragma unroll
    for (int dt = 0; dt < G::DT; ++dt) dqacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bf16* kg = reinterpret_cast<const bf16*>(p.k);
    const bf16* vg = reinterpret_cast<const bf16*>(p.v);
    bf16x8 kreg[G::NCH], vreg[G::NCH];
    if (ntile > 0) {
        tile_load<DH>(kg, tl_row[0], p.k_stride, h * DH, tl_n[0], tid, kreg);
        tile_load<DH>(vg, tl_row[0], p.v_stride, h * DH, tl_n[0], tid, vreg);
    }
    for (int t = 0; t < ntile; ++t) {
        __syncthreads();
        tile_store<DH>(Ks, tid, kreg);
        tile_store<DH>(Vs, tid, vreg);
        __syncthreads();
        const int kn = tl_n[t];
        if (t + 1 < ntile) {
            tile_load<DH>(kg, tl_row[t + 1], p.k_stride, h * DH, tl_n[t + 1], tid, kreg);
            tile_load<DH>(vg, tl_row[t + 1], p.v_stride, h * DH, tl_n[t + 1], tid, vreg);
        }
        f32x4 s[4], dp[4];
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
            s[t4] = f32x4{0.f, 0.f, 0.f, 0.f};
            dp[t4] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks) {
                s[t4] = mma16(ld8(Ks + (16 * t4 + lr) * G::KP + 32 * ks + 8 * g), qf[ks], s[t4]);
                dp[t4] = mma16(ld8(Vs + (16 * t4 + lr) * G::KP + 32 * ks + 8 * g), dof[ks], dp[t4]);
            }
        }
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float pv = fast_exp2(__builtin_fmaf(s[t4][r], c, -lse2));
                if (kn < 64 && 16 * t4 + 4 * g + r >= kn) pv = 0.f;
                s[t4][r] = pv * (dp[t4][r] - delta) * p.scale;       // dS^T
            }
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2) {
            const bf16x8 dsb = pack8(s[2 * ks2], s[2 * ks2 + 1]);
#pragma unroll
            for (int dt = 0; dt < G::DT; ++dt)
                dqacc[dt] = mma16(tr_frag(Ks, G::KP, 32 * ks2, 16 * dt, lane), dsb, dqacc[dt]);
        }
    }
    if (qvalid) {
        bf16* dqp = reinterpret_cast<bf16*>(p.dq) + (qrow0 + myq) * p.dq_stride + h * DH + 4 * g;
#pragma unroll
        for (int dt = 0; dt < G::DT; ++dt) st4(dqp + 16 * dt, dqacc[dt]);
    }
}

// ------------------------------------------------------------------------------------------------------ backward: dK, dV
template <int DH>
__global__ __launch_bounds__(256) void mha_bf16_bwd_dkdv_kernel(MhaDesc p) {
    typedef Geo<DH> G;
    __shared__ __attribute__((aligned(16))) bf16 Qs[64 * G::KP];
    __shared__ __attribute__((aligned(16))) bf16 dOs[64 * G::KP];
    __shared__ float lse_s[64], delta_s[64];
    __shared__ int tl_row[MAXT], tl_n[MAXT], tl_mode[MAXT], tl_cnt;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, g = lane >> 4;
    const BlockSel bs = decode_block(blockIdx.x, p.max_tiles, p.B, p.H);
    if (bs.b < 0) return;
    const int b = bs.b, h = bs.h;
    const int* qlen = p.q_len + b * p.nseg; const int* qst = p.q_start + b * p.nseg;
    const int* klen = p.k_len + b * p.nseg; const int* kst = p.k_start + b * p.nseg;
    const TileSel ts = select_tile(klen, p.nseg, bs.t);
    if (ts.seg < 0) return;
    const long krow0 = (long)kst[ts.seg] + ts.t0;
    if (tid == 0) {
        int n = 0;
        for (int sq = 0; sq < p.nseg; ++sq) {
            int mode = 0;                                          // 0 skip, 1 normal, 2 uniform row
            if (sq == p.nseg - 1 || sq == ts.seg) mode = 1;
            else if (klen[sq] == 0 && p.empty_mode == 0) mode = 2;
            if (mode == 0) continue;
            for (int q0 = 0; q0 < qlen[sq] && n < MAXT; q0 += 64) {
                tl_row[n] = qst[sq] + q0; tl_n[n] = min(64, qlen[sq] - q0); tl_mode[n] = mode; ++n;
            }
        }
        tl_cnt = n;
    }
    const int mykey = wave * 16 + lr;
    const bool kvalid = mykey < ts.n;
    bf16x8 kf[G::KS], vf[G::KS];
    {
        const bf16* kp_ = reinterpret_cast<const bf16*>(p.k) + (krow0 + mykey) * p.k_stride + h * DH + 8 * g;
        const bf16* vp_ = reinterpret_cast<const bf16*>(p.v) + (krow0 + mykey) * p.v_stride + h * DH + 8 * g;
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) {
            kf[ks] = kvalid ? ld8(kp_ + 32 * ks) : z8();
            vf[ks] = kvalid ? ld8(vp_ + 32 * ks) : z8();
        }
    }
    f32x4 dkacc[G::DT], dvacc[G::DT];
#pragma unroll
    for (int dt = 0; dt < G::DT; ++dt) { dkacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dvacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    __syncthreads();
    const int ntile = tl_cnt;
    const float c = p.scale * LOG2E;
    const bf16* qg = reinterpret_cast<const bf16*>(p.q);
    const bf16* dog = reinterpret_cast<const bf16*>(p.dout);
    bf16x8 qreg[G::NCH], doreg[G::NCH];
    float lreg = 0.f, dreg = 0.f;
    auto fetch = [&](int t) {
        const int row = tl_row[t], n = tl_n[t];
        tile_load<DH>(qg, row, p.q_stride, h * DH, n, tid, qreg);
        tile_load<DH>(dog, row, p.do_stride, h * DH, n, tid, doreg);
        if (tid < 64) {
            const bool v = tid < n;
            lreg = v ? p.lse[(long)h * p.stat_stride + row + tid] * LOG2E : 0.f;
            dreg = v ? p.delta[(long)h * p.stat_stride + row + tid] : 0.f;
        }
    };
    if (ntile > 0) fetch(0);
    for (int t = 0; t < ntile; ++t) {
        __syncthreads();
        tile_store<DH>(Qs, tid, qreg);
        tile_store<DH>(dOs, tid, doreg);
        if (tid < 64) { lse_s[tid] = lreg; delta_s[tid] = dreg; }
        __syncthreads();
        const int qn = tl_n[t], mode = tl_mode[t];
        if (t + 1 < ntile) fetch(t + 1);
        f32x4 s[4], dp[4];
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
            s[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
            dp[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks) {
                s[qt] = mma16(ld8(Qs + (16 * qt + lr) * G::KP + 32 * ks + 8 * g), kf[ks], s[qt]);
                dp[qt] = mma16(ld8(dOs + (16 * qt + lr) * G::KP + 32 * ks + 8 * g), vf[ks], dp[qt]);
            }
        }
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
            const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_s + 16 * qt + 4 * g);
            const f32x4 d4 = *reinterpret_cast<const f32x4*>(delta_s + 16 * qt + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qi = 16 * qt + 4 * g + r;
                float pv, ds;
                if (mode == 2) { pv = fast_exp2(-l4[r]); ds = 0.f; }
                else { pv = fast_exp2(__builtin_fmaf(s[qt][r], c, -l4[r])); ds = pv * (dp[qt][r] - d4[r]) * p.scale; }
                const bool valid = kvalid && qi < qn;
                s[qt][r] = valid ? pv : 0.f;
                dp[qt][r] = valid ? ds : 0.f;
            }
        }
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2) {
            const bf16x8 pb = pack8(s[2 * ks2], s[2 * ks2 + 1]);
            const bf16x8 dsb = pack8(dp[2 * ks2], dp[2 * ks2 + 1]);
#pragma unroll
            for (int dt = 0; dt < G::DT; ++dt) {
                dvacc[dt] = mma16(tr_frag(dOs, G::KP, 32 * ks2, 16 * dt, lane), pb, dvacc[dt]);
                dkacc[dt] = mma16(tr_frag(Qs, G::KP, 32 * ks2, 16 * dt, lane), dsb, dkacc[dt]);
            }
        }
    }
    if (kvalid) {
        bf16* dkp = reinterpret_cast<bf16*>(p.dk) + (krow0 + mykey) * p.dk_stride + h * DH + 4 * g;
        bf16* dvp = reinterpret_cast<bf16*>(p.dv) + (krow0 + mykey) * p.dv_stride + h * DH + 4 * g;
#pragma unroll
        for (int dt = 0; dt < G::DT; ++dt) { st4(dkp + 16 * dt, dkacc[dt]); st4(dvp + 16 * dt, dvacc[dt]); }
    }
}

// ------------------------------------------------------------------------------------------------------ host side
int mha_bf16_fwd(const MhaDesc& d, int head_dim, hipStream_t st) {
    if (d.max_tiles > MAXT) return MMAE_ERR_ARG;
    dim3 grid(xcd_grid(d.B, d.H, d.max_tiles));
    if (head_dim == 64) hipLaunchKernelGGL((mha_bf16_fwd_kernel<64>), grid, dim3(256), 0, st, d);
    else hipLaunchKernelGGL((mha_bf16_fwd_kernel<32>), grid, dim3(256), 0, st, d);
    MMAE_CHECK_LAUNCH();
    return MMAE_OK;
}

int mha_bf16_bwd(MhaDesc d, int head_dim, int max_q_tiles, int max_k_tiles, hipStream_t st) {
    if (max_q_tiles > MAXT || max_k_tiles > MAXT) return MMAE_ERR_ARG;
    d.max_tiles = max_q_tiles;
    if (head_dim == 64) hipLaunchKernelGGL((mha_bf16_bwd_dq_kernel<64>), dim3(xcd_grid(d.B, d.H, max_q_tiles)), dim3(256), 0, st, d);
    else hipLaunchKernelGGL((mha_bf16_bwd_dq_kernel<32>), dim3(xcd_grid(d.B, d.H, max_q_tiles)), dim3(256), 0, st, d);
    MMAE_CHECK_LAUNCH();
    d.max_tiles = max_k_tiles;
    if (head_dim == 64) hipLaunchKernelGGL((mha_bf16_bwd_dkdv_kernel<64>), dim3(xcd_grid(d.B, d.H, max_k_tiles)), dim3(256), 0, st, d);
    else hipLaunchKernelGGL((mha_bf16_bwd_dkdv_kernel<32>), dim3(xcd_grid(d.B, d.H, max_k_tiles)), dim3(256), 0, st, d);
    MMAE_CHECK_LAUNCH();
    return MMAE_OK;
}
