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
// Tried and rejected (tools/bench_attn.py, one process): prefetch distance 2 (-4 %), 128-row query tiles / 8 waves (+-0),
// one persistent block per (sample, head) pair (-17 %: many short blocks overlap better than few long ones).
#include <type_traits>
#include "mha_common.hpp"
#include "mmae_hip.h"

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
#define LDS_AS __attribute__((address_space(3)))
#define MAXT 80   // tiles one block may sweep (keys of a sample / 64 + segments)

#define LOG2E 1.4426950408889634f
#define LN2 0.6931471805599453f

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// Cross-lane reductions over the four 16-lane rows of a wave (lanes l, l^16, l^32, l^48) with VALU-only lane swaps
// (v_permlane16_swap / v_permlane32_swap) instead of __shfl_xor, which lowers to ds_bpermute through the LDS crossbar.
// permlane32_swap(a, b): lanes 32-63 of a <-> lanes 0-31 of b; with a == b == v the two results hold, for every lane,
// v of its own half and v of the other half.  permlane16_swap does the same for odd/even 16-lane rows.
__device__ __forceinline__ float rows_max(float v) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    v = fmaxf(__builtin_bit_cast(float, (unsigned)a[0]), __builtin_bit_cast(float, (unsigned)a[1]));
    const unsigned w = __builtin_bit_cast(unsigned, v);
    auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    return fmaxf(__builtin_bit_cast(float, (unsigned)b[0]), __builtin_bit_cast(float, (unsigned)b[1]));
}
__device__ __forceinline__ float rows_sum(float v) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    v = __builtin_bit_cast(float, (unsigned)a[0]) + __builtin_bit_cast(float, (unsigned)a[1]);
    const unsigned w = __builtin_bit_cast(unsigned, v);
    auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    return __builtin_bit_cast(float, (unsigned)b[0]) + __builtin_bit_cast(float, (unsigned)b[1]);
}

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

template <int DH, int NT = 256> struct Geo {
    // LDS row pitch.  DH 64: 160 B.  With 16-byte slots s = (10 * row + chunk) mod 16 the ds_read_b128 lane groups
    // ({0-3, 12-15, 20-27}: rows 0-3 / 12-15 of chunk g, rows 4-11 of chunk g+1) hit 16 distinct slots, and the 8 rows x 32 B of
    // one transposed read (ds_read_b64_tr_b16, 32-lane half) start at banks 40 * row mod 64 = {0,40,16,56,32,8,48,24}: both
    // conflict-free.  The former 144 B pitch (DH + 8) made both kinds of read 2-way: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
    // was 36-41 % in all three kernels, with the LDS array 54-57 % busy in the backward ones (now 0 % / 33-37 %; the kernel
    // times did not move: the LDS array was not what they wait for, see the forward kernel's notes on block start-up latency).
    // DH 32 (the decoders' 8 heads x 32): 96 B.  Slots s = (6 * row + chunk) mod 16 are distinct over a b128 lane group (rows 0-3 /
    // 12-15 of chunk g: {0,6,12,2,8,14,4,10}, rows 4-11 of chunk g+1: {9,15,5,11,1,7,13,3}), and the 8 rows x 32 B of a
    // transposed read start at banks 24 * row mod 64 = {0,24,48,8,32,56,16,40}.  The 80 B pitch (DH + 8) left 40-48 % of the LDS
    // cycles of the three DH-32 kernels conflicted (profiles/r02_sq_step.md).
    static constexpr int KS = DH / 32, DT = DH / 16, KP = DH == 64 ? 80 : 48, CPR = DH / 8, NCH = 64 * CPR / NT;
};

// global -> registers: this thread's chunks of a [<=64 rows][DH] tile (rows >= n zero-filled)
template <int DH, int NT = 256>
__device__ __forceinline__ void tile_load(const bf16* src, long row0, long stride, int col0, int n, int tid,
                                          bf16x8 (&reg)[Geo<DH, NT>::NCH]) {
#pragma unroll
    for (int i = 0; i < Geo<DH, NT>::NCH; ++i) {
        const int c = tid + NT * i, r = c / Geo<DH>::CPR, dc = c % Geo<DH>::CPR;
        reg[i] = r < n ? ld8(src + (row0 + r) * stride + col0 + dc * 8) : z8();
    }
}
template <int DH, int NT = 256>
__device__ __forceinline__ void tile_store(bf16* img, int tid, const bf16x8 (&reg)[Geo<DH, NT>::NCH]) {
#pragma unroll
    for (int i = 0; i < Geo<DH, NT>::NCH; ++i) {
        const int c = tid + NT * i, r = c / Geo<DH>::CPR, dc = c % Geo<DH>::CPR;
        *reinterpret_cast<bf16x8*>(img + r * Geo<DH>::KP + dc * 8) = reg[i];
    }
}

// Segment table of one sample, fetched in ONE parallel round trip: lane s (< nseg) holds entry s of the four arrays;
// get() broadcasts a (wave-uniform) entry with a lane read.  Replaces ~5 dependent global reads per block prologue.
struct SegTab {
    int qlen, qst, klen, kst;
    __device__ __forceinline__ void load(const MhaDesc& p, int b, int lane) {
        const int i = b * p.nseg + (lane < p.nseg ? lane : 0);
        qlen = p.q_len[i]; qst = p.q_start[i]; klen = p.k_len[i]; kst = p.k_start[i];
    }
    // v_readlane_b32 (the segment index is wave-uniform): a VALU -> SGPR move, not a ds_bpermute trip through the LDS crossbar
    __device__ __forceinline__ int ql(int s) const { return __builtin_amdgcn_readlane(qlen, __builtin_amdgcn_readfirstlane(s)); }
    __device__ __forceinline__ int qs(int s) const { return __builtin_amdgcn_readlane(qst, __builtin_amdgcn_readfirstlane(s)); }
    __device__ __forceinline__ int kl(int s) const { return __builtin_amdgcn_readlane(klen, __builtin_amdgcn_readfirstlane(s)); }
    __device__ __forceinline__ int ks(int s) const { return __builtin_amdgcn_readlane(kst, __builtin_amdgcn_readfirstlane(s)); }
};
// (segment, 64-row tile) of linear tile index t over the lengths len(s); seg = -1: none
template <int BM = 64, typename F> __device__ __forceinline__ TileSel pick_tile(F len, int nseg, int t) {
    TileSel r; r.seg = -1; r.t0 = 0; r.n = 0;
    for (int s = 0; s < nseg; ++s) {
        const int L = len(s);
        const int nt = (L + BM - 1) / BM;
        if (r.seg < 0 && t < nt) { r.seg = s; r.t0 = t * BM; r.n = min(BM, L - t * BM); }
        t -= nt;
    }
    return r;
}

// key tiles a query tile sweeps: filled by thread 0, read by everyone after a barrier
struct KeyPlan { int begin, end; bool uniform; };
__device__ __forceinline__ KeyPlan key_plan(int seg, int nseg, int klen_seg, int empty_mode) {
    KeyPlan k; k.begin = 0; k.end = 0; k.uniform = false;
    if (seg == nseg - 1) { k.begin = 0; k.end = nseg; }
    else if (klen_seg > 0) { k.begin = seg; k.end = seg + 1; }
    else if (empty_mode == 0) { k.begin = 0; k.end = nseg; k.uniform = true; }
    return k;
}

// ------------------------------------------------------------------------------------------------------ forward
// Per-thread staging geometry: chunk i of this thread covers row (tid + NT*i) / CPR, 8 columns at ((tid + NT*i) % CPR)*8.
// The element offset inside a tile is loop invariant; only the tile's first row (a wave-uniform scalar) changes.
template <int DH, int NT> struct StageIdx {
    int boff[Geo<DH, NT>::NCH];      // (row * stride + col) * 2: byte offset relative to the tile's first row
    int loff[Geo<DH, NT>::NCH];      // LDS element offset
    int row_bytes;                   // stride * 2
    __device__ __forceinline__ void init(int tid, long stride, int col0) {
        row_bytes = (int)stride * 2;
#pragma unroll
        for (int i = 0; i < Geo<DH, NT>::NCH; ++i) {
            const int c = tid + NT * i, r = c / Geo<DH>::CPR, dc = c % Geo<DH>::CPR;
            boff[i] = (r * (int)stride + col0 + dc * 8) * 2; loff[i] = r * Geo<DH>::KP + dc * 8;
        }
    }
};
// Rows >= n of the tile read as ZERO through the buffer descriptor's range check (raw buffer, num_records = n rows):
// no per-row compare / exec mask / branch around the loads.  `base` (first row of the tile) and n are wave-uniform.
typedef int i32x4v __attribute__((ext_vector_type(4)));
template <int DH, int NT>
__device__ __forceinline__ void tile_load2(const bf16* base, const StageIdx<DH, NT>& ix, int n,
                                           bf16x8 (&reg)[Geo<DH, NT>::NCH]) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(base), 0, n * ix.row_bytes, 0x00020000);
#pragma unroll
    for (int i = 0; i < Geo<DH, NT>::NCH; ++i)
        reg[i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, ix.boff[i], 0, 0));
}
template <int DH, int NT>
__device__ __forceinline__ void tile_store2(bf16* img, const StageIdx<DH, NT>& ix, const bf16x8 (&reg)[Geo<DH, NT>::NCH]) {
#pragma unroll
    for (int i = 0; i < Geo<DH, NT>::NCH; ++i) *reinterpret_cast<bf16x8*>(img + ix.loff[i]) = reg[i];
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// __launch_bounds__(.., 2): at most 256 registers per lane, so the compiler selects the VGPR form of the MFMAs -- with the
// default budget it parks accumulators in AGPRs and pays ~80 v_accvgpr_read/write per tile around the softmax.
// QB = 16-query blocks per wave: with QB = 2 a block covers 128 queries with the same 4 waves, so every K/V tile that is
// fetched (L2/HBM) and every K/V fragment read from LDS feeds twice the MFMAs -- the kernel is bound by exactly those two
// streams (K/V tiles are re-read once per query tile: 1.7 GB per launch at the bench shape), not by issue slots.
template <int DH, int NW, int QB = 1>
__global__ __launch_bounds__(NW * 64, 2) void mha_bf16_fwd_kernel(MhaDesc p) {
    typedef Geo<DH, NW * 64> G;
    constexpr int NT = NW * 64, BM = NW * 16 * QB;
    __shared__ __attribute__((aligned(16))) bf16 Ks[64 * G::KP];
    __shared__ __attribute__((aligned(16))) bf16 Vs[64 * G::KP];
    __shared__ int tl_row[MAXT], tl_n[MAXT], tl_cnt;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, g = lane >> 4;
    const BlockSel bs = decode_block(blockIdx.x, p.max_tiles, p.B, p.H);
    if (bs.b < 0) return;
    const int b = bs.b, h = bs.h;
    SegTab st; st.load(p, b, lane);
    const TileSel ts = pick_tile<BM>([&](int s) { return st.ql(s); }, p.nseg, bs.t);
    if (ts.seg < 0) return;
    const long qrow0 = (long)st.qs(ts.seg) + ts.t0;
    const KeyPlan kp = key_plan(ts.seg, p.nseg, st.kl(ts.seg), p.empty_mode);
    if (wave == 0) {                                           // whole wave takes part in the lane broadcasts
        int n = 0;
        for (int s = kp.begin; s < kp.end; ++s) {
            const int L = st.kl(s), r0 = st.ks(s);
            for (int j0 = 0; j0 < L && n < MAXT; j0 += 64) { if (lane == 0) { tl_row[n] = r0 + j0; tl_n[n] = min(64, L - j0); } ++n; }
        }
        if (lane == 0) tl_cnt = n;
    }
    int myq[QB]; bool qvalid[QB];
    bf16x8 qf[QB][G::KS];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        myq[qb] = (wave * QB + qb) * 16 + lr;
        qvalid[qb] = myq[qb] < ts.n;
        const bf16* qp = reinterpret_cast<const bf16*>(p.q) + (qrow0 + myq[qb]) * p.q_stride + h * DH + 8 * g;
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) qf[qb][ks] = qvalid[qb] ? ld8(qp + 32 * ks) : z8();
    }
    __syncthreads();
    const int ntile = uni(tl_cnt);
    float m[QB], l[QB];                                       // m is kept in the scaled (log2) domain
    f32x4 oacc[QB][G::DT];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        m[qb] = -INFINITY; l[qb] = 0.f;
#pragma unroll
        for (int dt = 0; dt < G::DT; ++dt) oacc[qb][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    const bf16* kg = reinterpret_cast<const bf16*>(p.k);
    const bf16* vg = reinterpret_cast<const bf16*>(p.v);
    StageIdx<DH, NT> ixk, ixv;
    ixk.init(tid, p.k_stride, h * DH);
    ixv.init(tid, p.v_stride, h * DH);
    // LDS fragment addresses (loop invariant)
    const bf16* kfrag = Ks + lr * G::KP + 8 * g;
    bf16x8 kreg[G::NCH], vreg[G::NCH];
    if (ntile > 0) {
        const long r0 = uni(tl_row[0]); const int n0 = uni(tl_n[0]);
        tile_load2<DH, NT>(kg + r0 * p.k_stride, ixk, n0, kreg);
        tile_load2<DH, NT>(vg + r0 * p.v_stride, ixv, n0, vreg);
    }
    if (!kp.uniform) {
        const float c = p.scale * LOG2E;                      // scores enter the exp2 domain through one fma
        for (int t = 0; t < ntile; ++t) {
            __syncthreads();
            tile_store2<DH, NT>(Ks, ixk, kreg);
            tile_store2<DH, NT>(Vs, ixv, vreg);
            __syncthreads();
            const int kn = uni(tl_n[t]);
            if (t + 1 < ntile) {
                const long r1 = uni(tl_row[t + 1]); const int n1 = uni(tl_n[t + 1]);
                tile_load2<DH, NT>(kg + r1 * p.k_stride, ixk, n1, kreg);
                tile_load2<DH, NT>(vg + r1 * p.v_stride, ixv, n1, vreg);
            }
            f32x4 s[QB][4];
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) {
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) s[qb][t4] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < G::KS; ++ks) {
                    const bf16x8 kfr = ld8(kfrag + 16 * t4 * G::KP + 32 * ks);        // one LDS read, QB products
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) s[qb][t4] = mma16(kfr, qf[qb][ks], s[qb][t4]);
                }
            }
            bf16x8 pb[QB][2];
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                f32x4 (&sq)[4] = s[qb];
                if (kn < 64) {                                // ragged last tile of a segment: scalar branch
#pragma unroll
                    for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (16 * t4 + 4 * g + r >= kn) sq[t4][r] = -INFINITY;
                }
                float mx = fmaxf(fmaxf(sq[0][0], sq[0][1]), fmaxf(sq[0][2], sq[0][3]));
#pragma unroll
                for (int t4 = 1; t4 < 4; ++t4) mx = fmaxf(mx, fmaxf(fmaxf(sq[t4][0], sq[t4][1]), fmaxf(sq[t4][2], sq[t4][3])));
                mx = rows_max(mx);
                const float m_new = fmaxf(m[qb], mx * c);     // c > 0: max(c*s) = c*max(s); every tile has a valid key
                const float alpha = fast_exp2(m[qb] - m_new);
                float rs = 0.f;
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float e = fast_exp2(__builtin_fmaf(sq[t4][r], c, -m_new));   // -inf -> 0
                        sq[t4][r] = e;
                        rs += e;
                    }
                l[qb] = l[qb] * alpha + rs;                   // per-lane partial; the four key rows are summed after the loop
                m[qb] = m_new;
#pragma unroll
                for (int dt = 0; dt < G::DT; ++dt) oacc[qb][dt] *= alpha;
                pb[qb][0] = pack8(sq[0], sq[1]);
                pb[qb][1] = pack8(sq[2], sq[3]);
            }
#pragma unroll
            for (int ks2 = 0; ks2 < 2; ++ks2)
#pragma unroll
                for (int dt = 0; dt < G::DT; ++dt) {
                    const bf16x8 vfr = tr_frag(Vs, G::KP, 32 * ks2, 16 * dt, lane);    // one transposed LDS read, QB products
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) oacc[qb][dt] = mma16(vfr, pb[qb][ks2], oacc[qb][dt]);
                }
        }
    } else {
        // fully masked rows (finite masked_fill in the reference): uniform attention over every key -> column sums of V
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) m[qb] = 0.f;
        for (int t = 0; t < ntile; ++t) {
            __syncthreads();
            tile_store2<DH, NT>(Vs, ixv, vreg);
            __syncthreads();
            const int kn = uni(tl_n[t]);
            if (t + 1 < ntile) {
                const long r1 = uni(tl_row[t + 1]); const int n1 = uni(tl_n[t + 1]);
                tile_load2<DH, NT>(vg + r1 * p.v_stride, ixv, n1, vreg);
            }
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) l[qb] += 0.25f * (float)kn;   // (the four lane rows are summed after the loop)
            f32x4 one4[4];
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
                for (int r = 0; r < 4; ++r) one4[t4][r] = (16 * t4 + 4 * g + r < kn) ? 1.f : 0.f;
#pragma unroll
            for (int ks2 = 0; ks2 < 2; ++ks2) {
                const bf16x8 pb = pack8(one4[2 * ks2], one4[2 * ks2 + 1]);
#pragma unroll
                for (int dt = 0; dt < G::DT; ++dt) {
                    const bf16x8 vfr = tr_frag(Vs, G::KP, 32 * ks2, 16 * dt, lane);
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) oacc[qb][dt] = mma16(vfr, pb, oacc[qb][dt]);
                }
            }
        }
    }
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const float lq = rows_sum(l[qb]);
        if (qvalid[qb]) {
            const float inv = lq > 0.f ? 1.f / lq : 0.f;
            bf16* op = reinterpret_cast<bf16*>(p.o) + (qrow0 + myq[qb]) * p.o_stride + h * DH + 4 * g;
#pragma unroll
            for (int dt = 0; dt < G::DT; ++dt) st4(op + 16 * dt, oacc[qb][dt] * inv);
            if (g == 0) p.lse[(long)h * p.stat_stride + qrow0 + myq[qb]] = lq > 0.f ? m[qb] * LN2 + __logf(lq) : 0.f;
        }
    }
}

// __launch_bounds__(.., 2): at most 256 registers per lane, so the compiler selects the VGPR form of the MFMAs -- with the
// default budget it parks accumulators in AGPRs and pays ~80 v_accvgpr_read/write per tile around the softmax.
// ------------------------------------------------------------------------------------------------------ backward: dQ (+ delta)
// No per-element masks: rows of a ragged tile beyond its length are ZERO in the LDS images, so a padded key has K = V = 0
// and contributes K^T dS = 0 to dQ whatever its (finite) dS is.
// QB = 16-query blocks per wave (see the forward kernel): 2 -> 128-query tiles, K/V tile fetches and fragment reads halve.
template <int DH, int QB = 1>
__global__ __launch_bounds__(256, 2) void mha_bf16_bwd_dq_kernel(MhaDesc p) {
    typedef Geo<DH> G;
    constexpr int NT = 256, BM = 64 * QB;
    __shared__ __attribute__((aligned(16))) bf16 Ks[64 * G::KP];
    __shared__ __attribute__((aligned(16))) bf16 Vs[64 * G::KP];
    __shared__ int tl_row[MAXT], tl_n[MAXT], tl_cnt;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, g = lane >> 4;
    const BlockSel bs = decode_block(blockIdx.x, p.max_tiles, p.B, p.H);
    if (bs.b < 0) return;
    const int b = bs.b, h = bs.h;
    SegTab st; st.load(p, b, lane);
    const TileSel ts = pick_tile<BM>([&](int s) { return st.ql(s); }, p.nseg, bs.t);
    if (ts.seg < 0) return;
    const long qrow0 = (long)st.qs(ts.seg) + ts.t0;
    const KeyPlan kp = key_plan(ts.seg, p.nseg, st.kl(ts.seg), p.empty_mode);
    if (wave == 0) {
        int n = 0;
        if (!kp.uniform)          // dS == 0 for a uniform (fully masked) row: nothing flows to q
            for (int s = kp.begin; s < kp.end; ++s) {
                const int L = st.kl(s), r0 = st.ks(s);
                for (int j0 = 0; j0 < L && n < MAXT; j0 += 64) { if (lane == 0) { tl_row[n] = r0 + j0; tl_n[n] = min(64, L - j0); } ++n; }
            }
        if (lane == 0) tl_cnt = n;
    }
    int myq[QB]; bool qvalid[QB];
    bf16x8 qf[QB][G::KS], dof[QB][G::KS];
    f32x4 neg_lse4[QB], neg_delta4[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        myq[qb] = (wave * QB + qb) * 16 + lr;
        qvalid[qb] = myq[qb] < ts.n;
        float dpart = 0.f;
        const bf16* qp = reinterpret_cast<const bf16*>(p.q) + (qrow0 + myq[qb]) * p.q_stride + h * DH + 8 * g;
        const bf16* dop = reinterpret_cast<const bf16*>(p.dout) + (qrow0 + myq[qb]) * p.do_stride + h * DH + 8 * g;
        const bf16* op = reinterpret_cast<const bf16*>(p.o) + (qrow0 + myq[qb]) * p.o_stride + h * DH + 8 * g;
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) {
            qf[qb][ks] = qvalid[qb] ? ld8(qp + 32 * ks) : z8();
            dof[qb][ks] = qvalid[qb] ? ld8(dop + 32 * ks) : z8();
            const bf16x8 of = qvalid[qb] ? ld8(op + 32 * ks) : z8();
#pragma unroll
            for (int j = 0; j < 8; ++j) dpart += (float)dof[qb][ks][j] * (float)of[j];
        }
        const float delta = rows_sum(dpart);
        const float lse2 = (qvalid[qb] ? p.lse[(long)h * p.stat_stride + qrow0 + myq[qb]] : 0.f) * LOG2E;
        if (qvalid[qb] && g == 0) {
            // plane 0: delta; planes 1 / 2: the row constants as the key-stationary kernels seed them (-lse in the log2 domain, -delta)
            const long at = (long)h * p.stat_stride + qrow0 + myq[qb], plane = (long)p.H * p.stat_stride;
            p.delta[at] = delta; p.delta[at + plane] = -lse2; p.delta[at + 2 * plane] = -delta;
        }
        neg_lse4[qb] = f32x4{-lse2, -lse2, -lse2, -lse2};
        neg_delta4[qb] = f32x4{-delta, -delta, -delta, -delta};
    }
    __syncthreads();
    const int ntile = uni(tl_cnt);
    // Row constants as the INITIAL accumulators (the query, hence lse and delta, is fixed per lane): with Q pre-scaled by
    // scale*log2(e) the score chain ends as S' = log2-domain score - lse2 and the dP chain as dP - delta, so per score only
    // exp2 and one multiply remain (was fma, exp2, sub, mul, mul); the softmax scale is applied to dQ once at the end.
    {
        const float c = p.scale * LOG2E;
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) qf[qb][ks][j] = (bf16)((float)qf[qb][ks][j] * c);
    }

    f32x4 dqacc[QB][G::DT];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int dt = 0; dt < G::DT; ++dt) dqacc[qb][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bf16* kg = reinterpret_cast<const bf16*>(p.k);
    const bf16* vg = reinterpret_cast<const bf16*>(p.v);
    StageIdx<DH, NT> ixk, ixv;
    ixk.init(tid, p.k_stride, h * DH);
    ixv.init(tid, p.v_stride, h * DH);
    const bf16* kfrag = Ks + lr * G::KP + 8 * g;
    const bf16* vfrag = Vs + lr * G::KP + 8 * g;
    bf16x8 kreg[G::NCH], vreg[G::NCH];
    if (ntile > 0) {
        const long r0 = uni(tl_row[0]); const int n0 = uni(tl_n[0]);
        tile_load2<DH, NT>(kg + r0 * p.k_stride, ixk, n0, kreg);
        tile_load2<DH, NT>(vg + r0 * p.v_stride, ixv, n0, vreg);
    }
    for (int t = 0; t < ntile; ++t) {
        __syncthreads();
        tile_store2<DH, NT>(Ks, ixk, kreg);
        tile_store2<DH, NT>(Vs, ixv, vreg);
        __syncthreads();
        if (t + 1 < ntile) {
            const long r1 = uni(tl_row[t + 1]); const int n1 = uni(tl_n[t + 1]);
            tile_load2<DH, NT>(kg + r1 * p.k_stride, ixk, n1, kreg);
            tile_load2<DH, NT>(vg + r1 * p.v_stride, ixv, n1, vreg);
        }
        f32x4 s[QB][4], dp[QB][4];
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) { s[qb][t4] = neg_lse4[qb]; dp[qb][t4] = neg_delta4[qb]; }
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks) {
                const bf16x8 kfr = ld8(kfrag + 16 * t4 * G::KP + 32 * ks);
                const bf16x8 vfr = ld8(vfrag + 16 * t4 * G::KP + 32 * ks);
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) {
                    s[qb][t4] = mma16(kfr, qf[qb][ks], s[qb][t4]);
                    dp[qb][t4] = mma16(vfr, dof[qb][ks], dp[qb][t4]);
                }
            }
        }
        bf16x8 dsb[QB][2];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    s[qb][t4][r] = fast_exp2(s[qb][t4][r]) * dp[qb][t4][r];   // dS^T / scale (finite also for padded keys)
            dsb[qb][0] = pack8(s[qb][0], s[qb][1]);
            dsb[qb][1] = pack8(s[qb][2], s[qb][3]);
        }
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2)
#pragma unroll
            for (int dt = 0; dt < G::DT; ++dt) {
                const bf16x8 kt = tr_frag(Ks, G::KP, 32 * ks2, 16 * dt, lane);
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) dqacc[qb][dt] = mma16(kt, dsb[qb][ks2], dqacc[qb][dt]);
            }
    }
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
        if (qvalid[qb]) {
            bf16* dqp = reinterpret_cast<bf16*>(p.dq) + (qrow0 + myq[qb]) * p.dq_stride + h * DH + 4 * g;
#pragma unroll
            for (int dt = 0; dt < G::DT; ++dt) st4(dqp + 16 * dt, dqacc[qb][dt] * p.scale);
        }
}


// Head-looping form of the dQ kernel (variant 23): one block per (sample, 64-query tile) walks the H heads.  The segment table,
// the tile choice and the key-tile list are the same for every head of a sample, so they are built once; the next head's Q / dO /
// O rows, its lse and its first K/V tile are requested while the current head's last key tile is being processed -- the two
// dependent trips to memory that open every block of the tile-per-(head) kernel (43 % of a block's life at S = 640) are paid once
// per 8 heads and otherwise hidden.
template <int DH>
__global__ __launch_bounds__(256, 2) void mha_bf16_bwd_dq_heads_kernel(MhaDesc p) {
    typedef Geo<DH> G;
    constexpr int NT = 256;
    __shared__ __attribute__((aligned(16))) bf16 Ks[64 * G::KP];
    __shared__ __attribute__((aligned(16))) bf16 Vs[64 * G::KP];
    __shared__ int tl_row[MAXT], tl_n[MAXT], tl_cnt;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, g = lane >> 4;
    const BlockSel bs = decode_block(blockIdx.x, p.max_tiles, p.B, 1);          // (sample, tile): the head is the loop below
    if (bs.b < 0) return;
    const int b = bs.b;
    SegTab st; st.load(p, b, lane);
    const TileSel ts = pick_tile<64>([&](int s) { return st.ql(s); }, p.nseg, bs.t);
    if (ts.seg < 0) return;
    const long qrow0 = (long)st.qs(ts.seg) + ts.t0;
    const KeyPlan kp = key_plan(ts.seg, p.nseg, st.kl(ts.seg), p.empty_mode);
    if (wave == 0) {
        int n = 0;
        if (!kp.uniform)
            for (int s = kp.begin; s < kp.end; ++s) {
                const int L = st.kl(s), r0 = st.ks(s);
                for (int j0 = 0; j0 < L && n < MAXT; j0 += 64) { if (lane == 0) { tl_row[n] = r0 + j0; tl_n[n] = min(64, L - j0); } ++n; }
            }
        if (lane == 0) tl_cnt = n;
    }
    const int myq = wave * 16 + lr;
    const bool qvalid = myq < ts.n;
    const long qr = qrow0 + myq;
    const bf16* qbase = reinterpret_cast<const bf16*>(p.q) + qr * p.q_stride + 8 * g;
    const bf16* dobase = reinterpret_cast<const bf16*>(p.dout) + qr * p.do_stride + 8 * g;
    const bf16* obase = reinterpret_cast<const bf16*>(p.o) + qr * p.o_stride + 8 * g;
    // prefetch registers: the rows of the NEXT head
    bf16x8 qn[G::KS], don[G::KS], on[G::KS];
    float lsen = 0.f;
    auto load_rows = [&](int h) {
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) {
            qn[ks] = qvalid ? ld8(qbase + h * DH + 32 * ks) : z8();
            don[ks] = qvalid ? ld8(dobase + h * DH + 32 * ks) : z8();
            on[ks] = qvalid ? ld8(obase + h * DH + 32 * ks) : z8();
        }
        lsen = qvalid ? p.lse[(long)h * p.stat_stride + qr] : 0.f;
    };
    load_rows(0);
    __syncthreads();
    const int ntile = uni(tl_cnt);
    const bf16* kg = reinterpret_cast<const bf16*>(p.k);
    const bf16* vg = reinterpret_cast<const bf16*>(p.v);
    StageIdx<DH, NT> ixk, ixv;
    ixk.init(tid, p.k_stride, 0);
    ixv.init(tid, p.v_stride, 0);
    const bf16* kfrag = Ks + lr * G::KP + 8 * g;
    const bf16* vfrag = Vs + lr * G::KP + 8 * g;
    bf16x8 kreg[G::NCH], vreg[G::NCH];
    if (ntile > 0) {
        const long r0 = uni(tl_row[0]); const int n0 = uni(tl_n[0]);
        tile_load2<DH, NT>(kg + r0 * p.k_stride, ixk, n0, kreg);
        tile_load2<DH, NT>(vg + r0 * p.v_stride, ixv, n0, vreg);
    }
    const float c = p.scale * LOG2E;
    const long plane = (long)p.H * p.stat_stride;
    for (int h = 0; h < p.H; ++h) {
        bf16x8 qf[G::KS], dof[G::KS];
        float dpart = 0.f;
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) {
            dof[ks] = don[ks];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                dpart += (float)don[ks][j] * (float)on[ks][j];
                qf[ks][j] = (bf16)((float)qn[ks][j] * c);
            }
        }
        const float delta = rows_sum(dpart);
        const float lse2 = lsen * LOG2E;
        if (qvalid && g == 0) {
            const long at = (long)h * p.stat_stride + qr;
            p.delta[at] = delta; p.delta[at + plane] = -lse2; p.delta[at + 2 * plane] = -delta;
        }
        const f32x4 neg_lse4 = f32x4{-lse2, -lse2, -lse2, -lse2}, neg_delta4 = f32x4{-delta, -delta, -delta, -delta};
        f32x4 dqacc[G::DT];
#pragma unroll
        for (int dt = 0; dt < G::DT; ++dt) dqacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (ntile == 0 && h + 1 < p.H) load_rows(h + 1);
        for (int t = 0; t < ntile; ++t) {
            __syncthreads();
            tile_store2<DH, NT>(Ks, ixk, kreg);
            tile_store2<DH, NT>(Vs, ixv, vreg);
            __syncthreads();
            if (t + 1 < ntile) {
                const long r1 = uni(tl_row[t + 1]); const int n1 = uni(tl_n[t + 1]);
                tile_load2<DH, NT>(kg + r1 * p.k_stride + h * DH, ixk, n1, kreg);
                tile_load2<DH, NT>(vg + r1 * p.v_stride + h * DH, ixv, n1, vreg);
            } else if (h + 1 < p.H) {                       // last key tile of this head: the next head's rows and first tile
                const long r1 = uni(tl_row[0]); const int n1 = uni(tl_n[0]);
                tile_load2<DH, NT>(kg + r1 * p.k_stride + (h + 1) * DH, ixk, n1, kreg);
                tile_load2<DH, NT>(vg + r1 * p.v_stride + (h + 1) * DH, ixv, n1, vreg);
                load_rows(h + 1);
            }
            f32x4 s[4], dp[4];
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) {
                s[t4] = neg_lse4; dp[t4] = neg_delta4;
#pragma unroll
                for (int ks = 0; ks < G::KS; ++ks) {
                    const bf16x8 kfr = ld8(kfrag + 16 * t4 * G::KP + 32 * ks);
                    const bf16x8 vfr = ld8(vfrag + 16 * t4 * G::KP + 32 * ks);
                    s[t4] = mma16(kfr, qf[ks], s[t4]);
                    dp[t4] = mma16(vfr, dof[ks], dp[t4]);
                }
            }
            bf16x8 dsb[2];
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
                for (int r = 0; r < 4; ++r) s[t4][r] = fast_exp2(s[t4][r]) * dp[t4][r];
            dsb[0] = pack8(s[0], s[1]);
            dsb[1] = pack8(s[2], s[3]);
#pragma unroll
            for (int ks2 = 0; ks2 < 2; ++ks2)
#pragma unroll
                for (int dt = 0; dt < G::DT; ++dt) dqacc[dt] = mma16(tr_frag(Ks, G::KP, 32 * ks2, 16 * dt, lane), dsb[ks2], dqacc[dt]);
        }
        if (qvalid) {
            bf16* dqp = reinterpret_cast<bf16*>(p.dq) + qr * p.dq_stride + h * DH + 4 * g;
#pragma unroll
            for (int dt = 0; dt < G::DT; ++dt) st4(dqp + 16 * dt, dqacc[dt] * p.scale);
        }
    }
}

// ------------------------------------------------------------------------------------------------------ backward: dK, dV
// No per-element masks either: a padded query row has Q = dO = 0 (zero-filled image) and lse = delta = 0, so P is finite
// and both dO^T P and Q^T dS get exactly 0 from it; a padded key column only pollutes its own (never stored) column.
template <int DH>
__global__ __launch_bounds__(256, 2) void mha_bf16_bwd_dkdv_kernel(MhaDesc p) {
    typedef Geo<DH> G;
    constexpr int NT = 256;
    __shared__ __attribute__((aligned(16))) bf16 Qs[64 * G::KP];
    __shared__ __attribute__((aligned(16))) bf16 dOs[64 * G::KP];
    __shared__ __attribute__((aligned(16))) float lse_s[64];
    __shared__ __attribute__((aligned(16))) float delta_s[64];
    __shared__ int tl_row[MAXT], tl_n[MAXT], tl_mode[MAXT], tl_cnt;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, g = lane >> 4;
    const BlockSel bs = decode_block(blockIdx.x, p.max_tiles, p.B, p.H);
    if (bs.b < 0) return;
    const int b = bs.b, h = bs.h;
    SegTab st; st.load(p, b, lane);
    const TileSel ts = pick_tile([&](int s) { return st.kl(s); }, p.nseg, bs.t);
    if (ts.seg < 0) return;
    const long krow0 = (long)st.ks(ts.seg) + ts.t0;
    if (wave == 0) {
        int n = 0;
        for (int sq = 0; sq < p.nseg; ++sq) {
            const int QL = st.ql(sq), r0 = st.qs(sq), KL = st.kl(sq);
            int mode = 0;                                          // 0 skip, 1 normal, 2 uniform row
            if (sq == p.nseg - 1 || sq == ts.seg) mode = 1;
            else if (KL == 0 && p.empty_mode == 0) mode = 2;
            if (mode == 0) continue;
            for (int q0 = 0; q0 < QL && n < MAXT; q0 += 64) {
                if (lane == 0) { tl_row[n] = r0 + q0; tl_n[n] = min(64, QL - q0); tl_mode[n] = mode; }
                ++n;
            }
        }
        if (lane == 0) tl_cnt = n;
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
        // K pre-scaled by scale*log2(e) (it only feeds the score product here); -lse2 / -delta of the tile's query rows are
        // the initial accumulators of the S and dP chains (see the dQ kernel); the softmax scale goes onto dK at the end
        const float c = p.scale * LOG2E;
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) kf[ks][j] = (bf16)((float)kf[ks][j] * c);
    }
    f32x4 dkacc[G::DT], dvacc[G::DT];
#pragma unroll
    for (int dt = 0; dt < G::DT; ++dt) { dkacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dvacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    __syncthreads();
    const int ntile = uni(tl_cnt);
    const bf16* qg = reinterpret_cast<const bf16*>(p.q);
    const bf16* dog = reinterpret_cast<const bf16*>(p.dout);
    StageIdx<DH, NT> ixq, ixo;
    ixq.init(tid, p.q_stride, h * DH);
    ixo.init(tid, p.do_stride, h * DH);
    const bf16* qfrag = Qs + lr * G::KP + 8 * g;
    const bf16* dofrag = dOs + lr * G::KP + 8 * g;
    const float* lse_g = p.lse + (long)h * p.stat_stride;
    const float* delta_g = p.delta + (long)h * p.stat_stride;
    bf16x8 qreg[G::NCH], doreg[G::NCH];
    float lreg = 0.f, dreg = 0.f;
    auto fetch = [&](int t) {
        const long row = uni(tl_row[t]); const int n = uni(tl_n[t]);
        tile_load2<DH, NT>(qg + row * p.q_stride, ixq, n, qreg);
        tile_load2<DH, NT>(dog + row * p.do_stride, ixo, n, doreg);
        if (tid < 64) {
            const bool v = tid < n;
            lreg = v ? -lse_g[row + tid] * LOG2E : 0.f;            // stored negated: they are accumulator seeds
            dreg = v ? -delta_g[row + tid] : 0.f;
        }
    };
    if (ntile > 0) fetch(0);
    for (int t = 0; t < ntile; ++t) {
        __syncthreads();
        tile_store2<DH, NT>(Qs, ixq, qreg);
        tile_store2<DH, NT>(dOs, ixo, doreg);
        if (tid < 64) { lse_s[tid] = lreg; delta_s[tid] = dreg; }
        __syncthreads();
        const int mode = uni(tl_mode[t]);
        if (t + 1 < ntile) fetch(t + 1);
        f32x4 s[4], dp[4];
        if (mode == 1) {
#pragma unroll
            for (int qt = 0; qt < 4; ++qt) {
                s[qt] = *reinterpret_cast<const f32x4*>(lse_s + 16 * qt + 4 * g);        // -lse2 of the four query rows
                dp[qt] = *reinterpret_cast<const f32x4*>(delta_s + 16 * qt + 4 * g);     // -delta
#pragma unroll
                for (int ks = 0; ks < G::KS; ++ks) {
                    s[qt] = mma16(ld8(qfrag + 16 * qt * G::KP + 32 * ks), kf[ks], s[qt]);
                    dp[qt] = mma16(ld8(dofrag + 16 * qt * G::KP + 32 * ks), vf[ks], dp[qt]);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = fast_exp2(s[qt][r]);
                    s[qt][r] = pv;
                    dp[qt][r] = pv * dp[qt][r];                                           // dS / scale
                }
            }
        } else {                                                  // uniform (fully masked) query rows: P = 1/K, dS = 0
#pragma unroll
            for (int qt = 0; qt < 4; ++qt) {
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_s + 16 * qt + 4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r) { s[qt][r] = fast_exp2(l4[r]); dp[qt][r] = 0.f; }
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
        for (int dt = 0; dt < G::DT; ++dt) { st4(dkp + 16 * dt, dkacc[dt] * p.scale); st4(dvp + 16 * dt, dvacc[dt]); }
    }
}

// ------------------------------------------------------------------------------------------------------ forward, 32x32x16
// Second-generation forward for head_dim 64 (what mmae_mha_fwd runs; the 16x16x32 kernel above stays as variant 1 / 8 and
// serves head_dim 32).  What changes against mha_bf16_fwd_kernel:
//   * v_mfma_f32_32x32x16_bf16 with the score tile computed transposed (S^T = K Q^T): a lane owns ONE query and 16 keys of
//     each 32-key block, so the row maximum is an in-lane max3 chain plus one v_permlane32_swap, and the exponentiated
//     accumulator registers 8s..8s+7 ARE the B operand of O^T += V^T P^T for k-step s (k-slot permutation absorbed by the
//     transposed V read) -- half the MFMA instructions per FLOP, 24 instead of 8 free VALU issue cycles under each MFMA;
//   * the running reference -m enters as the C operand of the first k-step (`negm`, a register splat that changes only on a
//     rescale), Q is pre-scaled by scale*log2(e): a score costs one v_exp_f32 and one add -- no fma, no subtract; the O / l
//     rescale happens only when a row's new scores exceed the reference by 2^6 (wave-uniform branch; exact algebra, only
//     the bf16 rounding point of P moves);
//   * padded keys of a segment's ragged last tile leave the softmax through the matrix core: one extra k-step whose Q side is
//     e_0 and whose K side holds -1e30 in slot 0 of every padded key (two MFMAs behind a scalar branch; written as
//     compare/select pairs hipcc if-converts them into EVERY iteration: 60 VALU per tile);
//   * fully masked rows (empty key segment, finite fill in the reference) run the same loop with a zero query;
//   * double-buffered K / V images, one barrier per key tile; the next tile's registers are written to the idle buffer between
//     the softmax and the PV products, so the ds_write latency hides under the PV MFMAs;
//   * K image pitch 144 B (the 32x32x16 A-operand b128 row reads are conflict-free), V image pitch 192 B (the 4 rows x 2
//     column blocks of one transposed read land on the four 64-byte quarters of the 256-byte bank row): SQ_LDS_BANK_CONFLICT 0.
// Measured (tools/bench_attn.py, one process, B 256): 233-248 us against 240-262 us of the 16x16x32 kernel on the same box
// (+3...+10 %).  Per-instruction counts fell by 2x (116 VALU + 16 MFMA per 32 x 64 score tile against 192 + 32), time did
// not follow: the stamped build (variant 9, tools/probes/attn_stamps.py) shows a wave spending 10-11k cycles between its first
// instruction and its first key tile -- ~2.8k until the segment table is in registers, ~6k until its (HBM-cold) Q rows and
// first K/V tile have arrived, ~2k for the LDS staging and two barriers -- against ~2.8k cycles per key-tile iteration and 5
// iterations per wave on average: block start-up latency, not the inner loop, bounds all three attention kernels at this
// sequence length.  Tried on top and measured no gain (removed again): -m through an extra k-step instead of the splat (16
// VGPRs fewer), 256-query tiles (QB 2: 255 VGPRs, -45 %), launch bounds for 3 / 4 waves per SIMD (spills), several query tiles
// per block with an exact grid (no empty blocks, one segment-table round trip per block), samples in reverse order (Infinity
// Cache), the segment table through scalar loads (slower: 32 serial K$ misses).
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ f32x16 mma32(const bf16x8& a, const bf16x8& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float swap32_max(float v) {
    const unsigned w = __builtin_bit_cast(unsigned, v);
    auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    return fmaxf(__builtin_bit_cast(float, (unsigned)b[0]), __builtin_bit_cast(float, (unsigned)b[1]));
}
__device__ __forceinline__ float swap32_sum(float v) {
    const unsigned w = __builtin_bit_cast(unsigned, v);
    auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    return __builtin_bit_cast(float, (unsigned)b[0]) + __builtin_bit_cast(float, (unsigned)b[1]);
}
// A operand of O^T[dh, query] += V^T[dh, key] P^T[key, query] for the 16 keys base..base+15 and dh block dhb (32 rows):
// element j of lane (r = lane & 31, hh = lane >> 5) is V[base + 8*(j >> 2) + 4*hh + (j & 3)][32*dhb + r] -- the k-slot order in
// which the 32x32 score accumulator hands out its registers (cdna_hip_programming.md, "An accumulator tile as the next
// MFMA's operand").  Two transposed reads of 4 keys x 16 columns per 16-lane group.
__device__ __forceinline__ bf16x8 tr32_frag(const bf16* img, int pitch, int base, int dhb, int lane) {
    const int g16 = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const bf16* a0 = img + (base + 4 * (g16 >> 1) + q) * pitch + 32 * dhb + 16 * (g16 & 1) + 4 * pp;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(a0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(a0 + 8 * pitch));
    s16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3]; r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return __builtin_bit_cast(bf16x8, r);
}
// staging of one [<= 64 rows][64] bf16 tile by 256 threads: 2 chunks of 16 B per thread, LDS pitch PITCH elements
template <int PITCH> struct Stage64 {
    int boff[2], loff[2], row_bytes;
    __device__ __forceinline__ void init(int tid, long stride, int col0) {
        row_bytes = (int)stride * 2;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid + 256 * i, rr = c >> 3, dc = c & 7;
            boff[i] = (rr * (int)stride + col0 + dc * 8) * 2; loff[i] = rr * PITCH + dc * 8;
        }
    }
    __device__ __forceinline__ void load(const bf16* base, int n, bf16x8 (&reg)[2]) const {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(base), 0, n * row_bytes, 0x00020000);
#pragma unroll
        for (int i = 0; i < 2; ++i) reg[i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, boff[i], 0, 0));
    }
    __device__ __forceinline__ void store(bf16* img, const bf16x8 (&reg)[2]) const {
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<bf16x8*>(img + loff[i]) = reg[i];
    }
};

#define FWD32_THR 6.0f    // log2 domain: P <= 2^6 between rescales

// Diagnostic build (STAMP; variant 9 of mmae_internal.h, never the product path): per-wave s_memtime sums of where the time
// goes -- [0] whole key loop, [1] inside the end-of-tile barrier, [2] in the register -> LDS staging block (its vmcnt wait),
// [3] iterations, [4] kernel entry -> first iteration, [5] 1 per wave with work, [6] entry -> segment table in registers,
// [7] -> Q rows and first K/V tile arrived.  One row per wave, plain stores (same-address atomics of 41k waves serialise for
// milliseconds and distort everything they time); mmae_debug_mha_stamps() sums and clears them.
#ifndef MMAE_DIAG
#define MMAE_DIAG 1
#endif
#define STAMP_WAVES (MMAE_DIAG ? 131072 : 1)
__device__ unsigned long long g_mha_stamps[STAMP_WAVES][8];
__device__ __forceinline__ unsigned long long stamp_now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}
extern "C" int mmae_debug_mha_stamps(unsigned long long* host8) {
    if (!host8 || !MMAE_DIAG) return MMAE_ERR_ARG;
    static unsigned long long* h = nullptr;
    if (!h) h = new unsigned long long[(size_t)STAMP_WAVES * 8];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_mha_stamps), sizeof(unsigned long long) * STAMP_WAVES * 8) != hipSuccess) return MMAE_ERR_LAUNCH;
    for (int i = 0; i < 8; ++i) host8[i] = 0;
    for (size_t w = 0; w < STAMP_WAVES; ++w)
        for (int i = 0; i < 8; ++i) host8[i] += h[w * 8 + i];
    void* dptr = nullptr;
    if (hipGetSymbolAddress(&dptr, HIP_SYMBOL(g_mha_stamps)) != hipSuccess) return MMAE_ERR_LAUNCH;
    if (hipMemset(dptr, 0, sizeof(unsigned long long) * STAMP_WAVES * 8) != hipSuccess) return MMAE_ERR_LAUNCH;
    return MMAE_OK;
}

template <int QB, bool STAMP = false>
__global__ __launch_bounds__(256, 2) void mha_bf16_fwd32_kernel(MhaDesc p) {
    unsigned long long t_entry = 0, t_seg = 0, t_q = 0, t_loop = 0, t_bar = 0, t_stage = 0;
    if (STAMP) t_entry = stamp_now();
    constexpr int DH = 64, BM = 128 * QB, KP = 72, VP = 96;
    __shared__ __attribute__((aligned(16))) bf16 Ks[2][64 * KP];
    __shared__ __attribute__((aligned(16))) bf16 Vs[2][64 * VP];
    __shared__ int tl_row[MAXT], tl_n[MAXT], tl_cnt;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
    const BlockSel bs = decode_block(blockIdx.x, p.max_tiles, p.B, p.H);
    if (bs.b < 0) return;
    const int b = bs.b, h = bs.h;
    SegTab st; st.load(p, b, lane);
    const TileSel ts = pick_tile<BM>([&](int s) { return st.ql(s); }, p.nseg, bs.t);
    if (ts.seg < 0) return;
    const long qrow0 = (long)st.qs(ts.seg) + ts.t0;
    if (STAMP) { __builtin_amdgcn_sched_barrier(0); t_seg = stamp_now(); __builtin_amdgcn_sched_barrier(0); }
    const KeyPlan kp = key_plan(ts.seg, p.nseg, st.kl(ts.seg), p.empty_mode);
    int first_row = 0, first_n = 0;                               // tile 0, known to every wave without the shared list
    if (wave == 0) {
        int n = 0;
        for (int s = kp.begin; s < kp.end; ++s) {
            const int L = st.kl(s), r0 = st.ks(s);
            for (int j0 = 0; j0 < L && n < MAXT; j0 += 64) { if (lane == 0) { tl_row[n] = r0 + j0; tl_n[n] = min(64, L - j0); } ++n; }
        }
        if (lane == 0) tl_cnt = n;
    }
    for (int s = kp.end - 1; s >= kp.begin; --s) {
        const int L = st.kl(s);
        if (L > 0) { first_row = st.ks(s); first_n = min(64, L); }
    }
    const bf16* kg = reinterpret_cast<const bf16*>(p.k);
    const bf16* vg = reinterpret_cast<const bf16*>(p.v);
    Stage64<KP> ixk; Stage64<VP> ixv;
    ixk.init(tid, p.k_stride, h * DH);
    ixv.init(tid, p.v_stride, h * DH);
    bf16x8 kreg[2], vreg[2];
    first_row = uni(first_row); first_n = uni(first_n);
    if (first_n > 0) {
        ixk.load(kg + (long)first_row * p.k_stride, first_n, kreg);
        ixv.load(vg + (long)first_row * p.v_stride, first_n, vreg);
    }
    // Q^T fragments (B operand of S^T = K Q^T), pre-scaled into the exp2 domain
    int myq[QB]; bool qvalid[QB];
    bf16x8 qf[QB][4];
    {
        const float c = p.scale * LOG2E;
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            myq[qb] = (wave * QB + qb) * 32 + r;
            qvalid[qb] = myq[qb] < ts.n;
            const bf16* qp = reinterpret_cast<const bf16*>(p.q) + (qrow0 + myq[qb]) * p.q_stride + h * DH + 8 * hh;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                // a fully masked row attends every key uniformly: a zero query gives every swept key the score 0
                qf[qb][ks] = (qvalid[qb] && !kp.uniform) ? ld8(qp + 16 * ks) : z8();
#pragma unroll
                for (int j = 0; j < 8; ++j) qf[qb][ks][j] = (bf16)((float)qf[qb][ks][j] * c);
            }
        }
    }
    float mref[QB], lsum[QB];
    f32x16 oacc[QB][2], negm[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        mref[qb] = 0.f; lsum[qb] = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) negm[qb][i] = 0.f;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int i = 0; i < 16; ++i) oacc[qb][d][i] = 0.f;
    }
    if (STAMP) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); t_q = stamp_now(); __builtin_amdgcn_sched_barrier(0); }
    __syncthreads();                                               // tile list visible
    const int ntile = uni(tl_cnt);
    if (ntile > 0) {
        ixk.store(Ks[0], kreg);
        ixv.store(Vs[0], vreg);
        if (ntile > 1) {
            const long r1 = uni(tl_row[1]); const int n1 = uni(tl_n[1]);
            ixk.load(kg + r1 * p.k_stride, n1, kreg);
            ixv.load(vg + r1 * p.v_stride, n1, vreg);
        }
    }
    __syncthreads();
    const bf16* kbase = Ks[0] + r * KP + 8 * hh;
    bf16x8 qone = z8();
    if (hh == 0) qone[0] = (bf16)1.0f;
    if (STAMP) t_loop = stamp_now();
    for (int t = 0; t < ntile; ++t) {
        const int cur = t & 1;
        const int kn = uni(tl_n[t]);
        const bf16* Kc = kbase + cur * (64 * KP);
        const bf16* Vc = Vs[cur];
        // ---- S'^T = K Q~^T - m : 2 key blocks x 4 k-steps, the first one seeded with -m
        f32x16 sacc[QB][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 kfr = ld8(Kc + 32 * kb * KP + 16 * ks);
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) sacc[qb][kb] = mma32(kfr, qf[qb][ks], ks == 0 ? negm[qb] : sacc[qb][kb]);
            }
        if (kn < 64) {                                             // ragged last tile of a segment (scalar branch)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                bf16x8 kmask = z8();
                if (hh == 0 && 32 * kb + r >= kn) kmask[0] = (bf16)(-1.0e30f);
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) sacc[qb][kb] = mma32(kmask, qone, sacc[qb][kb]);
            }
        }
        // ---- softmax in registers
        bf16x8 pb[QB][2][2];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            float mx = fmaxf(sacc[qb][0][0], sacc[qb][1][0]);
#pragma unroll
            for (int i = 1; i < 16; ++i) mx = fmaxf(fmaxf(mx, sacc[qb][0][i]), sacc[qb][1][i]);
            mx = swap32_max(mx);
            const bool need = (t == 0) | (mx > FWD32_THR);
            if (__builtin_amdgcn_ballot_w64(need) != 0) {          // wave-uniform: first tile, or some row outgrew the reference
                const float delta = t == 0 ? mx : fmaxf(mx, 0.f);
                if (t > 0) {
                    const float alpha = fast_exp2(-delta);
                    lsum[qb] *= alpha;
#pragma unroll
                    for (int d = 0; d < 2; ++d)
#pragma unroll
                        for (int i = 0; i < 16; ++i) oacc[qb][d][i] *= alpha;
                }
                mref[qb] += delta;
#pragma unroll
                for (int i = 0; i < 16; ++i) negm[qb][i] = -mref[qb];
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int i = 0; i < 16; ++i) sacc[qb][kb][i] -= delta;
            }
            float rs = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) { const float e = fast_exp2(sacc[qb][kb][i]); sacc[qb][kb][i] = e; rs += e; }
            lsum[qb] += rs;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) pb[qb][kb][s2][j] = (bf16)sacc[qb][kb][8 * s2 + j];
                }
        }
        // ---- the next tile's registers go to the idle buffer; the write latency hides under the PV products
        if (t + 1 < ntile) {
            unsigned long long a = 0;
            if (STAMP) { __builtin_amdgcn_sched_barrier(0); a = stamp_now(); __builtin_amdgcn_sched_barrier(0); }
            ixk.store(Ks[cur ^ 1], kreg);
            ixv.store(Vs[cur ^ 1], vreg);
            if (STAMP) { __builtin_amdgcn_sched_barrier(0); t_stage += stamp_now() - a; __builtin_amdgcn_sched_barrier(0); }
        }
        // ---- O^T += V^T P^T : 2 key blocks x 2 k-steps x 2 dh blocks
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const bf16x8 vfr = tr32_frag(Vc, VP, 32 * kb + 16 * s2, d, lane);
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) oacc[qb][d] = mma32(vfr, pb[qb][kb][s2], oacc[qb][d]);
                }
        if (STAMP) {
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long a = stamp_now();
            __syncthreads();
            t_bar += stamp_now() - a;
            __builtin_amdgcn_sched_barrier(0);
        } else {
            __syncthreads();
        }
        if (t + 2 < ntile) {
            const long r2 = uni(tl_row[t + 2]); const int n2 = uni(tl_n[t + 2]);
            ixk.load(kg + r2 * p.k_stride, n2, kreg);
            ixv.load(vg + r2 * p.v_stride, n2, vreg);
        }
    }
    if (STAMP) {
        const unsigned long long t_end = stamp_now();
        const unsigned w = blockIdx.x * 4 + wave;
        if (lane == 0 && w < STAMP_WAVES) {
            g_mha_stamps[w][0] = t_end - t_loop; g_mha_stamps[w][1] = t_bar; g_mha_stamps[w][2] = t_stage;
            g_mha_stamps[w][3] = (unsigned long long)ntile; g_mha_stamps[w][4] = t_loop - t_entry; g_mha_stamps[w][5] = 1ull;
            g_mha_stamps[w][6] = t_seg - t_entry; g_mha_stamps[w][7] = t_q - t_seg;
        }
    }
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const float lq = swap32_sum(lsum[qb]);
        if (qvalid[qb]) {
            const float inv = lq > 0.f ? 1.f / lq : 0.f;
            bf16* op = reinterpret_cast<bf16*>(p.o) + (qrow0 + myq[qb]) * p.o_stride + h * DH + 4 * hh;
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    f32x4 v;
                    v[0] = oacc[qb][d][4 * i] * inv; v[1] = oacc[qb][d][4 * i + 1] * inv;
                    v[2] = oacc[qb][d][4 * i + 2] * inv; v[3] = oacc[qb][d][4 * i + 3] * inv;
                    st4(op + 32 * d + 8 * i, v);
                }
            if (hh == 0) p.lse[(long)h * p.stat_stride + qrow0 + myq[qb]] = lq > 0.f ? mref[qb] * LN2 + __logf(lq) : 0.f;
        }
    }
}

// ------------------------------------------------------------------------------------------------------ host side
// `variant` (per call; mmae_internal.h): forward tiling 0 default, 1, 8; backward 2 -- tools/bench_attn.py A/B material.
int mha_bf16_fwd(const MhaDesc& d, int head_dim, int g_variant, hipStream_t st) {
    if (g_variant == 23 || (g_variant >= 50 && g_variant <= 55)) g_variant = 0;   // backward-only variants (head-looping dQ, fused backward): default forward
    if (d.max_tiles > MAXT) return MMAE_ERR_ARG;
    if (head_dim == 64 && g_variant >= 30 && g_variant <= 32) return mha_sh_fwd(d, g_variant - 20, st);   // two tiles per barrier (+ its diagnostics)
    if (head_dim == 64 && g_variant == 35) return mha_sh_fwd(d, 10, st);
    if (head_dim == 64 && g_variant == 45) return mha_sh_fwd(d, 4, st);   // diagnostic: sixteen global waves (wrong modality rows)
    if (head_dim == 64 && (g_variant == 40 || g_variant == 41)) return mha_sh_fwd(d, g_variant - 20, st);   // four loader waves (+ stream only)
    if (head_dim == 64 && g_variant >= 5 && g_variant <= 8) return mha_sh_fwd(d, g_variant - 5, st);   // sample-head forward (mha_sh.hip); 6 / 7 / 8: diagnostics
    // default: the sample-head kernel wherever a sample has enough query rows to fill its 16 waves (encoder blocks); the
    // tile-per-block kernel below for the few-query calls (attention pooling, contrastive pools: 1-8 queries per sample)
    if (head_dim == 64 && g_variant == 0 && mha_sh_applicable(d)) return mha_sh_fwd(d, 0, st);
    if (head_dim == 64 && (g_variant == 0 || g_variant == 3 || g_variant == 4 || g_variant == 9)) {
        // default (0 == 3): 32x32x16 forward, 4 waves x 32 queries = 128-query tiles; 4: 256-query tiles; 9: stamped diagnostic
        MhaDesc e = d;
        e.max_tiles = g_variant == 4 ? (d.max_tiles + 3) / 4 + d.nseg : (d.max_tiles + 1) / 2 + d.nseg;
        const dim3 grid(xcd_grid(e.B, e.H, e.max_tiles)), blk(256);
        if (g_variant == 4) MMAE_LAUNCH((mha_bf16_fwd32_kernel<2>), grid, blk, 0, st, e);
        else if (g_variant == 9) {
#if MMAE_DIAG
            MMAE_LAUNCH((mha_bf16_fwd32_kernel<1, true>), grid, blk, 0, st, e);
#else
            return MMAE_ERR_ARG;
#endif
        }
        else MMAE_LAUNCH((mha_bf16_fwd32_kernel<1>), grid, blk, 0, st, e);
    } else if (head_dim == 64 && g_variant == 2) {           // round-1 kernel: 16x16x32, 4 waves x 32 queries = 128-query tiles
        MhaDesc e = d; e.max_tiles = (d.max_tiles + 1) / 2 + d.nseg;
        MMAE_LAUNCH((mha_bf16_fwd_kernel<64, 4, 2>), dim3(xcd_grid(e.B, e.H, e.max_tiles)), dim3(256), 0, st, e);
    } else if (head_dim == 64 && g_variant == 18) {          // 8 waves x 16 queries (measured: no gain over 4 x 16)
        MhaDesc e = d; e.max_tiles = (d.max_tiles + 1) / 2 + d.nseg;
        MMAE_LAUNCH((mha_bf16_fwd_kernel<64, 8>), dim3(xcd_grid(e.B, e.H, e.max_tiles)), dim3(512), 0, st, e);
    } else {                                                 // variant 1 (dh 64) / dh 32: 16x16x32, 4 waves x 16 queries
        dim3 grid(xcd_grid(d.B, d.H, d.max_tiles));
        if (head_dim == 64) MMAE_LAUNCH((mha_bf16_fwd_kernel<64, 4>), grid, dim3(256), 0, st, d);
        else MMAE_LAUNCH((mha_bf16_fwd_kernel<32, 4>), grid, dim3(256), 0, st, d);
    }
    MMAE_CHECK_LAUNCH();
    return MMAE_OK;
}

int mha_bf16_bwd(MhaDesc d, int head_dim, int max_q_tiles, int max_k_tiles, int g_variant, hipStream_t st) {
    if (max_q_tiles > MAXT || max_k_tiles > MAXT) return MMAE_ERR_ARG;
    if ((g_variant >= 30 && g_variant <= 32) || g_variant == 40 || g_variant == 41 || g_variant == 45) g_variant = 0;   // forward-only variants: default backward
    if (g_variant == 35) g_variant = 5;                      // two-tile forward + sample-head dQ
    d.max_tiles = max_q_tiles;
    if (g_variant >= 50 && g_variant <= 55) {                // dQ + dK + dV in ONE key-stationary kernel (mha_sh.hip: mha_sh_bwd_kernel); 51..55: its diagnostics
        if (head_dim == 64 && mha_sh_fused_supported(d)) return mha_sh_bwd_fused(d, g_variant - 50, st);
        g_variant = 0;                                       // shapes beyond its schedule tables: the product pair
    }
    // dQ: the tile-per-block kernel stays the default.  The query-stationary sample-head kernel (mha_sh.hip: mha_sh_dq_kernel) runs
    // on request only (variant 5: tests, tools/bench_attn.py): its 8 KB of operand staging per wave cap it at 12 waves, i.e. 128 local
    // query rows per pass, and the usual Dirichlet splits keep more than 128 tokens of some modality -- a second pass.  Measured at
    // the bench shape: 349 against 378 us with equal splits, but 30 us SLOWER than this kernel at 201 / 64 / 119.
    const bool sh_dq = head_dim == 64 && mha_sh_dq_supported(d) && (g_variant == 5 || g_variant == 6 || g_variant == 7);
    if (sh_dq) {
        const int rc = mha_sh_dq(d, (g_variant == 6 || g_variant == 7) ? g_variant - 5 : 0, st);
        if (rc != MMAE_OK) return rc;
    } else
    if (head_dim == 64 && g_variant == 23) {                 // head-looping blocks: (sample, tile) grid, prefetch across heads
        MMAE_LAUNCH((mha_bf16_bwd_dq_heads_kernel<64>), dim3(xcd_grid(d.B, 1, max_q_tiles)), dim3(256), 0, st, d);
    } else
    if (head_dim == 64 && g_variant == 22) {                 // 128-query tiles: measured +-1 % (207 VGPRs, 2 waves/SIMD) -> not default
        d.max_tiles = (max_q_tiles + 1) / 2 + d.nseg;
        MMAE_LAUNCH((mha_bf16_bwd_dq_kernel<64, 2>), dim3(xcd_grid(d.B, d.H, d.max_tiles)), dim3(256), 0, st, d);
    } else if (head_dim == 64) MMAE_LAUNCH((mha_bf16_bwd_dq_kernel<64>), dim3(xcd_grid(d.B, d.H, max_q_tiles)), dim3(256), 0, st, d);
    else MMAE_LAUNCH((mha_bf16_bwd_dq_kernel<32>), dim3(xcd_grid(d.B, d.H, max_q_tiles)), dim3(256), 0, st, d);
    MMAE_CHECK_LAUNCH();
    d.max_tiles = max_k_tiles;
    // dK / dV: the key-stationary sample-head kernel (mha_sh.hip) where a sample has enough keys to fill it (encoder blocks) or
    // when asked for (variant 5: tests); variant 3 = the tile-per-block kernels of round 2 throughout
    if (head_dim == 64 && mha_sh_dkdv_supported(d) && (g_variant == 5 || ((g_variant == 0 || g_variant == 23) && d.max_k_rows >= 128))) return mha_sh_dkdv(d, 0, st);
    if (head_dim == 64 && mha_sh_dkdv_supported(d) && (g_variant == 6 || g_variant == 7)) return mha_sh_dkdv(d, g_variant - 5, st);   // diagnostics: stream only / no ring DMA
    if (head_dim == 64) MMAE_LAUNCH((mha_bf16_bwd_dkdv_kernel<64>), dim3(xcd_grid(d.B, d.H, max_k_tiles)), dim3(256), 0, st, d);
    else MMAE_LAUNCH((mha_bf16_bwd_dkdv_kernel<32>), dim3(xcd_grid(d.B, d.H, max_k_tiles)), dim3(256), 0, st, d);
    MMAE_CHECK_LAUNCH();
    return MMAE_OK;
}
