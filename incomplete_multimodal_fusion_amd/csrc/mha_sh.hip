// "Sample-head" kernels of the segment-masked attention, bf16, head_dim 64 (gfx950).  Semantics: mha.hip's header.
//
// What bounds the tile-per-block kernels of mha_bf16.hip at the pretraining shapes (S = 640 rows per sample, 8 heads x 64)
// is not the matrix core but start-up latency: a block lives for ~5 key tiles and spends 43 % of that waiting for its
// segment table, its Q rows and its first K/V tile (profiles/r02_attention.md), and every K/V tile is staged once per
// 128-query tile.  The whole forward moves 671 MB for 112 GFLOP -- at the chip's ridge it is an HBM-streaming problem.
// Here the Zorro mask's structure does the work instead:
//   * ONE workgroup (8 waves, one per CU) owns a sample and walks its heads; per (sample, head) every K/V tile is fetched
//     and staged exactly ONCE and serves both kinds of query that may see it -- the fusion queries (they see every key:
//     "global" slot of a wave, 32 queries) and the queries of the tile's own modality ("local" slot, 32 queries);
//   * K/V tiles arrive by LDS-DMA (buffer_load ... lds, 1 KiB per wave-instruction, no VGPRs) into a 4-stage ring, issued
//     three tiles ahead of their use and across head boundaries: the stream never drains between (sample, head) items, one
//     s_barrier per tile, counted vmcnt waits;
//   * Q rows are prefetched the same way into a wave-private staging area one segment (local slot) / one pass (global slot)
//     ahead -- no block barrier, the wave waits for its own DMA -- so a slot switch costs four LDS reads;
//   * images are plain 128-byte rows (the DMA destination is lane-linear); bank conflicts are removed by a swizzle applied
//     on the SOURCE side: 16-byte chunk c of row r sits at chunk c ^ f(r), f(r) = r1<<2 | r2<<1 | r3 (bits of r), which makes
//     both the b128 row reads and the ds_read_b64_tr_b16 transposed reads of the 32x32x16 operands conflict free
//     (tools/probes/lds_swizzle_check.py);
//   * the running reference -m and the mask of a ragged tile's padded keys enter the scores through ONE extra k-step
//     (K side [1, pad ? -1e30 : 0], Q side [-m, 1]): no accumulator splat, no per-element compare; m is kept
//     bf16-representable so the seed is exact.
// Segments longer than 256 queries take further passes over the tile list (chunk c = queries [256c, 256c + 256) of every
// segment); query segments whose key segment is empty attend uniformly (empty_mode 0: extra passes with a zero query on the
// global slot) or produce zeros (empty_mode 1).
#include "mha_common.hpp"
#include "mmae_hip.h"

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
#define LDS_AS __attribute__((address_space(3)))

#ifndef SH_SKEW
#define SH_SKEW 0
#endif
#ifndef SH_SLEEP_CLASS
#define SH_SLEEP_CLASS (-1)
#endif
#define SH_NS 4                 // ring stages
#define SH_D 3                  // tiles in flight ahead of the consumer
#define SH_MAXT 64              // key tiles per sample (one lane each)
#define SH_MAXP 64              // passes per sample (one lane each)
#define SH_LOG2E 1.4426950408889634f
#define SH_LN2 0.6931471805599453f
#define SH_THR 6.0f             // log2 domain: P <= 2^6 between rescales

__device__ __forceinline__ int sh_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int sh_f(int r) { return (((r >> 1) & 1) << 2) | (((r >> 2) & 1) << 1) | ((r >> 3) & 1); }
__device__ __forceinline__ float sh_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ f32x16 sh_mma(const bf16x8& a, const bf16x8& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float sh_swap_max(float v) {
    const unsigned w = __builtin_bit_cast(unsigned, v);
    auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    return fmaxf(__builtin_bit_cast(float, (unsigned)b[0]), __builtin_bit_cast(float, (unsigned)b[1]));
}
__device__ __forceinline__ float sh_swap_sum(float v) {
    const unsigned w = __builtin_bit_cast(unsigned, v);
    auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    return __builtin_bit_cast(float, (unsigned)b[0]) + __builtin_bit_cast(float, (unsigned)b[1]);
}
// s_waitcnt vmcnt(n) for a wave-uniform RUN-TIME n (the instruction only takes an immediate): a computed jump into a table of
// 32 { s_waitcnt vmcnt(k); s_branch end } pairs -- 8 scalar instructions whatever n is (as a C switch hipcc emitted a tree of a
// dozen compare / branch pairs per call, a few hundred cycles at the top of every tile).  n = how many YOUNGER vector-memory
// instructions may stay in flight; rounding n down only waits for more, so counts above 31 use 31.
#define SH_W1(k) "s_waitcnt vmcnt(" #k ")\n\ts_branch .Lshw_end_%=\n\t"
#define SH_W4(a, b, c, d) SH_W1(a) SH_W1(b) SH_W1(c) SH_W1(d)
__device__ __forceinline__ void sh_wait_vm(int n) {
    int k = sh_uni(n < 0 ? 0 : (n > 31 ? 31 : n));
    asm volatile(
        "s_getpc_b64 vcc\n\t"                 // address of the next instruction; the table starts 20 bytes further on
        "s_lshl_b32 %0, %0, 3\n\t"
        "s_add_u32 %0, %0, 20\n\t"
        "s_add_u32 vcc_lo, vcc_lo, %0\n\t"
        "s_addc_u32 vcc_hi, vcc_hi, 0\n\t"
        "s_setpc_b64 vcc\n\t"
        SH_W4(0, 1, 2, 3) SH_W4(4, 5, 6, 7) SH_W4(8, 9, 10, 11) SH_W4(12, 13, 14, 15)
        SH_W4(16, 17, 18, 19) SH_W4(20, 21, 22, 23) SH_W4(24, 25, 26, 27) SH_W4(28, 29, 30, 31)
        ".Lshw_end_%=:"
        : "+s"(k) : : "vcc", "scc", "memory");
}
__device__ __forceinline__ float sh_bf16_round(float x) { return (float)(bf16)x; }
__device__ __forceinline__ unsigned sh_pack2(float lo, float hi) {
    const unsigned a = __builtin_bit_cast(unsigned short, (bf16)lo), b = __builtin_bit_cast(unsigned short, (bf16)hi);
    return a | (b << 16);
}
__device__ __forceinline__ bf16x8 sh_ext(unsigned w0) { return __builtin_bit_cast(bf16x8, u32x4{w0, 0u, 0u, 0u}); }

// segment table of one sample: lane s holds entry s (one parallel round trip); wave-uniform reads through v_readlane
struct ShSeg {
    int qlen, qst, klen, kst;
    __device__ __forceinline__ void load(const MhaDesc& p, int b, int lane) {
        const int i = b * p.nseg + (lane < p.nseg ? lane : 0);
        qlen = p.q_len[i]; qst = p.q_start[i]; klen = p.k_len[i]; kst = p.k_start[i];
    }
    __device__ __forceinline__ int ql(int s) const { return __builtin_amdgcn_readlane(qlen, sh_uni(s)); }
    __device__ __forceinline__ int qs(int s) const { return __builtin_amdgcn_readlane(qst, sh_uni(s)); }
    __device__ __forceinline__ int kl(int s) const { return __builtin_amdgcn_readlane(klen, sh_uni(s)); }
    __device__ __forceinline__ int ks(int s) const { return __builtin_amdgcn_readlane(kst, sh_uni(s)); }
};

// one 1-KiB LDS-DMA piece: 8 rows x 128 B of a [rows][row_bytes] matrix; lane l fetches the 16-byte chunk that belongs at
// byte 16 l of the piece under the swizzle (voff, loop invariant per lane); rows >= n fail the range check
// The descriptor inputs go through v_readfirstlane: they ARE wave-uniform, but unless that is provable hipcc wraps every
// buffer instruction in a "waterfall" loop (4 readfirstlane + 2 compares + exec save / restore per DMA: the stamped build showed
// ~500 cycles per LDS-DMA instruction; cdna_hip_programming.md T20).
__device__ __forceinline__ void sh_dma(const bf16* base, int n_rows, int row_bytes, int voff, int soff, bf16* lds_piece) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    bf16* ub = reinterpret_cast<bf16*>(((unsigned long long)hi << 32) | lo);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(ub, 0, sh_uni(n_rows * row_bytes), 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_AS void*)lds_piece, 16, voff, sh_uni(soff), 0, 0);
}

// 64 floats (row constants), 4 bytes per lane; entries >= n read as 0
__device__ __forceinline__ void sh_dma4(const float* base, int n, float* lds, int lane) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    float* ub = reinterpret_cast<float*>(((unsigned long long)hi << 32) | lo);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(ub, 0, sh_uni(n * 4), 0x00020000);
    int z = 0;
    asm volatile("" : "+s"(z));
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_AS void*)lds, 4, (lane + z) * 4, 0, 0, 0);
}

// Plain buffer loads as inline assembly: the results arrive asynchronously and are waited for with the kernel's own counted
// s_waitcnt (sh_wait_vm + sh_pin) -- a load the compiler tracks would get ITS wait, which on a loader wave covers the ring DMA issued
// after it (the compiler merges the waves' different instruction streams to the strictest count).  Entries beyond `bytes` read as 0.
// (s_nop 4: the descriptor's SGPRs may come straight from v_readfirstlane -- five wait states before a VMEM instruction reads them, which
// the compiler's hazard recognizer provides for its own instructions but not inside an asm statement.)
__device__ __forceinline__ u32x4 sh_rsrc_words(const void* base, int bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    return u32x4{(unsigned)__builtin_amdgcn_readfirstlane((unsigned)a), (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) & 0xffffu,
                 (unsigned)sh_uni(bytes), 0x00020000u};
}
__device__ __forceinline__ u32x4 sh_ld128_async(const void* base, int bytes, int voff) {
    u32x4 v;
    asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, 0 offen" : "=&v"(v) : "v"(voff), "s"(sh_rsrc_words(base, bytes)) : "memory");
    return v;
}
__device__ __forceinline__ unsigned sh_ld32_async(const void* base, int bytes, int voff) {
    unsigned v;
    asm volatile("s_nop 4\n\tbuffer_load_dword %0, %1, %2, 0 offen" : "=&v"(v) : "v"(voff), "s"(sh_rsrc_words(base, bytes)) : "memory");
    return v;
}
// after sh_wait_vm: no use of the loaded registers may be scheduled above the wait
__device__ __forceinline__ void sh_pin(u32x4& a, u32x4& b, unsigned& c) { asm volatile("" : "+v"(a), "+v"(b), "+v"(c)); }
template <int CTRL>
__device__ __forceinline__ float sh_add_dpp(float x) {
    return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}

// transposed operand of the 32x32x16 products: element j of lane (r = lane & 31, hh = lane >> 5) is
// X[base + 8 (j >> 2) + 4 hh + (j & 3)][32 dhb + r] of the swizzled image (mha_bf16.hip: tr32_frag)
__device__ __forceinline__ bf16x8 sh_tr(const bf16* img, int off0, int off1) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(img + off0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(img + off1));
    s16x8 x;
    x[0] = lo[0]; x[1] = lo[1]; x[2] = lo[2]; x[3] = lo[3]; x[4] = hi[0]; x[5] = hi[1]; x[6] = hi[2]; x[7] = hi[3];
    return __builtin_bit_cast(bf16x8, x);
}
__device__ __forceinline__ bf16x8 sh_ld8(const bf16* p) { return *reinterpret_cast<const bf16x8*>(p); }

// Lane-invariant LDS element offsets of the fragment reads, swizzle folded in.  Only TWO registers: the other offsets differ
// from these by XOR with a constant (the swizzle is an XOR on address bits 3-5 of a 128-byte row, k-step / dh-block / row-group
// indices are XORed in the same way), and the kernels XOR them onto an address that already carries the ring stage -- a
// loop-variant value, so the compiler cannot hoist eight address registers out of the tile loop (at a 128-VGPR budget they
// were what spilled, and every scratch reload drains the DMA queue with its vmcnt(0)).
//   row read,  row (lane & 31) of a 32-row block, k-step ks :  krow0 ^ (16 ks)
//   transposed read, dh block d, row group e (rows +8 e)     :  tr00 ^ (32 d + 520 e)
struct ShAddr {
    int krow0, tr00;
    __device__ __forceinline__ void init(int lane) {
        const int r = lane & 31, hh = lane >> 5;
        krow0 = r * 64 + 8 * (hh ^ sh_f(r));
        const int g16 = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
        const int row = 4 * (g16 >> 1) + q, ch = 2 * (g16 & 1) + (pp >> 1);
        tr00 = row * 64 + 8 * (ch ^ sh_f(row)) + 4 * (pp & 1);
    }
};

// a wave's query slot: 32 queries (one per lane pair r / r + 32), online-softmax state in registers
struct ShSlot {
    bf16x8 q[4];
    f32x16 o[2];
    float mref, lsum;
    unsigned ext;         // Q side of the seed k-step: (bf16(-mref), 1.0) on lanes < 32, 0 above
    int row;              // this lane's query row in the q / out matrices
    int nq;               // valid queries of the slot (wave-uniform): lane pair r is live iff r < nq
};

// Diagnostic build (MODE 3; never the product path): per-wave s_memtime sums of where an iteration goes.  One row per wave,
// plain stores; mmae_debug_sh_stamps() sums the rows of the two roles separately and clears them.
//   [0] vmcnt wait at the top  [1] barrier  [2] ring issue + tile record  [3] slot switch  [4] QK^T  [5] softmax  [6] PV
//   [7] finish (stores)  [8] whole loop  [9] units (slot-steps)  [10] iterations  [11] waves
//   slot switch in detail: [12] wait for the staged Q  [13] LDS reads + conversion  [14] next-target scan  (the rest of [3] is the Q DMA issue)
#ifndef MMAE_DIAG
#define MMAE_DIAG 1
#endif
#define SH_STAMP_WAVES (MMAE_DIAG ? 8192 : 1)
__device__ unsigned long long g_sh_stamps[SH_STAMP_WAVES][16];
__device__ __forceinline__ unsigned long long sh_now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}
extern "C" int mmae_debug_sh_stamps(unsigned long long* host32) {
    if (!host32 || !MMAE_DIAG) return MMAE_ERR_ARG;
    static unsigned long long* h = nullptr;
    if (!h) h = new unsigned long long[(size_t)SH_STAMP_WAVES * 16];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_sh_stamps), sizeof(unsigned long long) * SH_STAMP_WAVES * 16) != hipSuccess) return MMAE_ERR_LAUNCH;
    for (int i = 0; i < 32; ++i) host32[i] = 0;
    for (size_t w = 0; w < SH_STAMP_WAVES; ++w)
        for (int i = 0; i < 16; ++i) host32[16 * ((w & 15) >= 8) + i] += h[w * 16 + i];      // [0..15] global waves, [16..31] local waves
    void* dptr = nullptr;
    if (hipGetSymbolAddress(&dptr, HIP_SYMBOL(g_sh_stamps)) != hipSuccess) return MMAE_ERR_LAUNCH;
    if (hipMemset(dptr, 0, sizeof(unsigned long long) * SH_STAMP_WAVES * 16) != hipSuccess) return MMAE_ERR_LAUNCH;
    return MMAE_OK;
}
#define SH_T(i) do { if (ST) { __builtin_amdgcn_sched_barrier(0); const unsigned long long n__ = sh_now(); tt[i] += n__ - tl; tl = n__; __builtin_amdgcn_sched_barrier(0); } } while (0)

// scores of 64 keys x 32 queries, online softmax, O^T += V^T P^T
template <bool ST>
__device__ __forceinline__ void sh_attend(ShSlot& s, bool fresh, const bf16* Kst, const bf16* Vst, int stage_el, const ShAddr& ad,
                                          const unsigned (&kext)[2], int hh, unsigned long long (&tt)[16], unsigned long long& tl) {
    f32x16 sacc[2];
    const bf16x8 qe = sh_ext(s.ext);
    const int kbase = ad.krow0 + stage_el, vbase = ad.tr00 + stage_el;      // element offsets incl. the ring stage (see ShAddr)
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        bf16x8 kf[4];                                             // the block's four K fragments are requested together
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) kf[ks] = sh_ld8(Kst + 2048 * kb + (kbase ^ (16 * ks)));
        sacc[kb] = sh_mma(sh_ext(kext[kb]), qe, z);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) sacc[kb] = sh_mma(kf[ks], s.q[ks], sacc[kb]);
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);        // emitted order: 4 LDS reads, then the 5 products
        __builtin_amdgcn_sched_group_barrier(0x008, 5, 0);
    }
    if (ST) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");    // let the MFMA chain drain into [4]
    SH_T(4);
    float mx0 = fmaxf(sacc[0][0], sacc[0][1]), mx1 = fmaxf(sacc[1][0], sacc[1][1]);      // two independent max3 chains
#pragma unroll
    for (int i = 2; i < 16; i += 2) { mx0 = fmaxf(fmaxf(mx0, sacc[0][i]), sacc[0][i + 1]); mx1 = fmaxf(fmaxf(mx1, sacc[1][i]), sacc[1][i + 1]); }
    const float mx = sh_swap_max(fmaxf(mx0, mx1));
    const bool need = fresh | (mx > SH_THR);
    if (__builtin_amdgcn_ballot_w64(need) != 0) {              // wave-uniform: first tile, or some row outgrew its reference
        const float want = s.mref + (fresh ? mx : fmaxf(mx, 0.f));
        const float mnew = sh_bf16_round(want);
        const float delta = mnew - s.mref;                       // exact: both are bf16 values
        if (!fresh) {
            const float alpha = sh_exp2(-delta);
            s.lsum *= alpha;
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int i = 0; i < 16; ++i) s.o[d][i] *= alpha;
        }
        s.mref = mnew;
        s.ext = hh == 0 ? sh_pack2(-mnew, 1.0f) : 0u;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) sacc[kb][i] -= delta;
    }
    // the row sum is taken over the bf16-ROUNDED probabilities (v_dot2c_f32_bf16 with a pair of ones: one instruction per two
    // keys): O = sum(P_r V) / sum(P_r) is then an exact weighted mean -- a row with a single key returns V itself, which the
    // backward's dP - delta cancellation relies on -- whatever the (bf16-valued, deferred) reference happens to be
    bf16x8 pb[2][2];
    const bf16x2 ones = {(bf16)1.0f, (bf16)1.0f};
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
        for (int i = 0; i < 16; ++i) sacc[kb][i] = sh_exp2(sacc[kb][i]);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
            for (int j = 0; j < 8; ++j) pb[kb][s2][j] = (bf16)sacc[kb][8 * s2 + j];
            const u32x4 w = __builtin_bit_cast(u32x4, pb[kb][s2]);
#pragma unroll
            for (int j = 0; j < 4; ++j) s.lsum = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, (unsigned)w[j]), ones, s.lsum, false);
        }
    }
    SH_T(5);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const bf16* vb = Vst + 64 * (32 * kb + 16 * s2);
            if (kb == 0 && s2 == 0 && fresh) {                    // a fresh slot's accumulators start from the C = 0 form: no zero fill
#pragma unroll
                for (int d = 0; d < 2; ++d) s.o[d] = sh_mma(sh_tr(vb, vbase ^ (32 * d), vbase ^ (32 * d + 520)), pb[kb][s2], z);
            } else {
#pragma unroll
                for (int d = 0; d < 2; ++d) s.o[d] = sh_mma(sh_tr(vb, vbase ^ (32 * d), vbase ^ (32 * d + 520)), pb[kb][s2], s.o[d]);
            }
        }
        if (kb == 0) __builtin_amdgcn_sched_barrier(0);           // at most 8 transposed fragments in flight (VGPR budget: 128)
    }
    if (ST) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    SH_T(6);
}

// Normalise and store a slot's 32 x 64 output tile.  A lane pair (r, r + 32) holds one query row: lane half hh owns the
// columns 32 d + 8 i + 4 hh + (0..3) of accumulator register group (d, i).  v_permlane32_swap on neighbouring groups (i, i + 1)
// gives each lane 16 contiguous bytes (lower half: columns 8 i .. 8 i + 7, upper half: 8 i + 8 .. 8 i + 15), so the tile leaves
// in 4 dwordx4 stores per lane instead of 8 dwordx2 -- the epilogue of a row-per-lane tile is store-ISSUE bound
// (cdna_hip_programming.md T21; the stamped build showed 6.6k cycles per finish with the narrow stores).
__device__ __forceinline__ void sh_finish(const ShSlot& s, const MhaDesc& p, int h, int r, int hh) {
    const float lq = sh_swap_sum(s.lsum);
    const float inv = lq > 0.f ? 1.f / lq : 0.f;
    const bool valid = r < s.nq;
    bf16* op = reinterpret_cast<bf16*>(p.o) + (long)s.row * p.o_stride + h * 64 + 8 * hh;
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int i = 0; i < 4; i += 2) {
            unsigned a0 = sh_pack2(s.o[d][4 * i] * inv, s.o[d][4 * i + 1] * inv), a1 = sh_pack2(s.o[d][4 * i + 2] * inv, s.o[d][4 * i + 3] * inv);
            unsigned b0 = sh_pack2(s.o[d][4 * i + 4] * inv, s.o[d][4 * i + 5] * inv), b1 = sh_pack2(s.o[d][4 * i + 6] * inv, s.o[d][4 * i + 7] * inv);
            auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false); a0 = r0[0]; b0 = r0[1];
            auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false); a1 = r1[0]; b1 = r1[1];
            if (valid) *reinterpret_cast<u32x4*>(op + 32 * d + 8 * i) = u32x4{a0, a1, b0, b1};
        }
    if (valid && hh == 0) p.lse[(long)h * p.stat_stride + s.row] = lq > 0.f ? s.mref * SH_LN2 + __logf(lq) : 0.f;
}

// zero output rows (empty key context under empty_mode 1, or a sample without keys): every thread of the block takes part
__device__ __forceinline__ void sh_zero_rows(const MhaDesc& p, long row0, int n, int h, int tid) {
    for (int i = tid; i < n * 8; i += 1024) {
        const int rr = i >> 3, c = i & 7;
        *reinterpret_cast<u32x4*>(reinterpret_cast<bf16*>(p.o) + (row0 + rr) * p.o_stride + h * 64 + 8 * c) = u32x4{0u, 0u, 0u, 0u};
        if (c == 0) p.lse[(long)h * p.stat_stride + row0 + rr] = 0.f;
    }
}

// MODE (diagnostic builds, never the product path): 1 = stream only (no scores), 2 = compute only (no K/V DMA: stale tiles),
// 3 = stamped (g_sh_stamps)
//
// 16 waves, four per SIMD, ONE query slot per wave: waves 0-7 hold the "global" queries of the pass (the fusion chunk, or a
// fully masked segment's chunk attending uniformly), waves 8-15 the "local" queries of the segment whose key tiles are being
// swept.  Each SIMD then runs two global and two local waves; with four resident waves the LDS / MFMA latencies of one wave
// are covered by the others without hand-interleaving (the first version of this kernel -- 8 waves with both slots in every
// wave, 212 VGPRs, two waves per SIMD running the same code in lock-step -- measured 230-240 us at the bench shape with
// 41 % of wave cycles parked and 22 % issue-stalled: no faster than the tile-per-block kernel).
// VAR 1: two key tiles per workgroup barrier -- the loop body runs twice behind one s_barrier: half the barriers, half the points
// at which 16 waves wait for the slowest one (a slot switch, a finish, a loader still issuing); the ring then holds the pair
// being consumed and the pair in flight (4 stages, tiles issued one pair-iteration ahead of their use).
// VAR 2: the 16 ring pieces of a tile are issued by FOUR loader waves (12-15: K pieces 0-3 / 4-7, V pieces 0-3 / 4-7), every step,
// instead of one rotating wave issuing all 16 every fourth step: an LDS-DMA instruction costs its issuer ~230 cycles, so the lone
// loader needs ~3.7 k cycles for a tile -- most of a step -- and when it also holds queries (rows 128-255 of a segment: every
// ragged split) its step runs long and the whole workgroup waits for it at the next barrier.
template <int MODE, int VAR>
__global__ __launch_bounds__(1024) void mha_sh_fwd_kernel(MhaDesc p, int hpb) {
    __shared__ __attribute__((aligned(1024))) bf16 ringK[SH_NS][4096];
    __shared__ __attribute__((aligned(1024))) bf16 ringV[SH_NS][4096];
    __shared__ __attribute__((aligned(1024))) bf16 qst[16][32 * 64];       // wave-private Q staging: 32 rows x 64
    const int tid = threadIdx.x, lane = tid & 63, wave = sh_uni(tid >> 6), r = lane & 31, hh = lane >> 5;
    const bool local = MODE == 4 ? false : wave >= 8;               // role of this wave (MODE 4, diagnostic: sixteen global waves --
                                                                    // waves 8-15 repeat the fusion queries of 0-7: what would twice the global work per step cost?)
    const int qw = wave & 7;                                        // its 32-query block inside a 256-query chunk
    const int hgroups = p.H / hpb;
    const int b = blockIdx.x / hgroups, h0 = (blockIdx.x % hgroups) * hpb;
    const int nseg = p.nseg, fus = nseg - 1;
    ShSeg st; st.load(p, b, lane);
    // Schedule of this sample, held in registers by every wave (lane i = entry i, read with v_readlane: no LDS round trip in the
    // loop).  Tiles: first key row and n | seg << 8 | (first-of-segment | last-of-segment << 1) << 16.
    // Passes: (global segment + 1) | global chunk << 8 | (local chunk + 1) << 16.
    int tile_row = 0, tile_info = 0, pass_info = 0, ntile = 0, npass = 0;
    {
        int acc = 0;
        for (int s = 0; s < nseg; ++s) {
            const int L = st.kl(s), nt = (L + 63) >> 6, j = lane - acc;
            if (j >= 0 && j < nt) { tile_row = st.ks(s) + 64 * j; tile_info = min(64, L - 64 * j) | (s << 8) | (((j == 0 ? 1 : 0) | (j == nt - 1 ? 2 : 0)) << 16); }
            acc += nt;
        }
        ntile = min(acc, SH_MAXT);
        int nch = 1;
        for (int s = 0; s < nseg; ++s) nch = max(nch, (st.ql(s) + 255) >> 8);
        if (lane < nch) pass_info = ((256 * lane < st.ql(fus) ? fus : -1) + 1) | (lane << 8) | ((lane + 1) << 16);
        acc = nch;
        if (p.empty_mode == 0)                                     // fully masked rows: uniform attention over every key
            for (int s = 0; s < fus; ++s)
                if (st.kl(s) == 0 && st.ql(s) > 0) {
                    const int nc = (st.ql(s) + 255) >> 8, j = lane - acc;
                    if (j >= 0 && j < nc) pass_info = (s + 1) | (j << 8);
                    acc += nc;
                }
        npass = min(acc, SH_MAXP);
    }
    auto t_row = [&](int t) { return __builtin_amdgcn_readlane(tile_row, t); };
    auto t_info = [&](int t) { return __builtin_amdgcn_readlane(tile_info, t); };
    auto p_info = [&](int pi) { return __builtin_amdgcn_readlane(pass_info, pi); };
    // rows that see no key at all
    for (int hi = 0; hi < hpb; ++hi)
        for (int s = 0; s < nseg; ++s) {
            const int QL = st.ql(s);
            if (QL > 0 && (ntile == 0 || (s < fus && st.kl(s) == 0 && p.empty_mode == 1))) sh_zero_rows(p, st.qs(s), QL, h0 + hi, tid);
        }
    if (ntile == 0) return;

    const bf16* qg = reinterpret_cast<const bf16*>(p.q);
    const bf16* kg = reinterpret_cast<const bf16*>(p.k);
    const bf16* vg = reinterpret_cast<const bf16*>(p.v);
    const int qsb = (int)p.q_stride * 2, kvsb = (int)p.k_stride * 2;          // (host: k_stride == v_stride)
    // DMA source offsets: lane l of a piece fills bytes [16 l, 16 l + 16) = row l >> 3, chunk position l & 7.
    // (row 8 j + prow of a block: f(row) = f(prow) ^ (j & 1), so ONE register serves every piece of a matrix: the row part 8 j
    // goes into the scalar offset, odd pieces flip bit 4)
    const int prow = lane >> 3;
    const int kvoff = prow * kvsb + 16 * ((lane & 7) ^ sh_f(prow));
    const int qvoff = prow * qsb + 16 * ((lane & 7) ^ sh_f(prow));
    ShAddr ad; ad.init(lane);
    const float cq = p.scale * SH_LOG2E;

    const int G = hpb * npass * ntile;
    // vmcnt bookkeeping: `vm` counts every vector-memory instruction this wave has issued (ring DMA, Q staging DMA, stores);
    // an operation's sequence number is vm right after its issue, and "wait for it" is s_waitcnt vmcnt(vm - seq).
    int vm = 0;
    // The ring is filled by ONE wave per tile, all 16 pieces (8 K + 8 V): an LDS-DMA instruction occupies the CU's
    // address path for ~40 cycles, and with every wave issuing its own piece right behind the barrier each of them sat
    // 500-800 cycles in that queue per tile (stamped build).  The loader of step j is wave 12 + (j & 3): the local-role waves of
    // the upper half of a chunk, which hold queries only where a modality keeps more than 128 tokens -- at the usual splits they
    // are idle, so the ~230 cycles an LDS-DMA instruction costs its issuer (16 per tile) are nobody's critical path.  Only the
    // loader waits for the step's pieces (vmcnt) before the barrier.
    int lt = 0, lh = 0, lpt = npass * ntile;                        // tile / head of ring step `lj`, tiles left in that head
    int lj = 0, lstage = 0, myseq = 0;
    int q0 = 0, q1 = 0, q2 = 0, nfl = 0;                            // VAR 2: sequence numbers of my pieces of the (up to three) tiles in flight
    auto issue_ring = [&]() {                                       // every wave advances the position; the step's loader issues
        if (VAR == 2) {
            if (MODE != 2 && wave >= 12) {
                const long row0 = t_row(lt); const int n = t_info(lt) & 255;
                const bool isv = wave >= 14;
                const bf16* b_ = (isv ? vg + row0 * p.v_stride : kg + row0 * p.k_stride) + (h0 + lh) * 64;
                bf16* dst = isv ? &ringV[lstage][0] : &ringK[lstage][0];
                const int pc0 = 4 * (wave & 1);
#pragma unroll
                for (int j = 0; j < 4; ++j) sh_dma(b_, n, kvsb, kvoff ^ (16 * (j & 1)), 8 * (pc0 + j) * kvsb, dst + (pc0 + j) * 512);
                vm += 4;
            }
            q0 = q1; q1 = q2; q2 = vm; ++nfl;
        } else if (MODE != 2 && wave == 12 + (lj & 3)) {
            const long row0 = t_row(lt); const int n = t_info(lt) & 255;
            const bf16* kb_ = kg + row0 * p.k_stride + (h0 + lh) * 64;
            const bf16* vb_ = vg + row0 * p.v_stride + (h0 + lh) * 64;
#pragma unroll
            for (int pc = 0; pc < 8; ++pc) {
                sh_dma(kb_, n, kvsb, kvoff ^ (16 * (pc & 1)), 8 * pc * kvsb, &ringK[lstage][pc * 512]);
                sh_dma(vb_, n, kvsb, kvoff ^ (16 * (pc & 1)), 8 * pc * kvsb, &ringV[lstage][pc * 512]);
            }
            vm += 16; myseq = vm;
        }
        ++lj;
        if (++lstage == SH_NS) lstage = 0;
        if (++lt == ntile) lt = 0;
        if (--lpt == 0) { lpt = npass * ntile; ++lh; }
    };
    // Q staging: rows [256 chunk + 32 qw, +32) of query segment s, head h -> qst[wave]
    auto rows_of = [&](int s, int chunk) { return min(32, st.ql(s) - 256 * chunk - 32 * qw); };
    auto issue_q = [&](int s, int chunk, int h) {
        const int nq = rows_of(s, chunk);
        const long row0 = (long)st.qs(s) + 256 * chunk + 32 * qw;
#pragma unroll
        for (int j = 0; j < 4; ++j) sh_dma(qg + row0 * p.q_stride + h * 64, nq, qsb, qvoff ^ (16 * (j & 1)), 8 * j * qsb, &qst[wave][8 * j * 64]);
        vm += 4;
    };
    ShSlot S;
    auto activate = [&](int sg, int chunk, bool zero_q) {
        // `z` is an opaque zero: whatever is added to it cannot be precomputed outside the tile loop (four staging addresses and
        // a row index kept across the loop were spilled to scratch, and a scratch reload waits for vmcnt(0): the DMA queue)
        int z = 0;
        asm volatile("" : "+s"(z));
        S.nq = rows_of(sg, chunk);
        S.row = st.qs(sg) + 256 * chunk + 32 * qw + z + r;
        const bf16* src = &qst[wave][0];
        const int qbase = ad.krow0 + z;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 raw = sh_ld8(src + (qbase ^ (16 * ks)));
#pragma unroll
            for (int j = 0; j < 8; ++j) S.q[ks][j] = (bf16)((float)raw[j] * cq);
        }
        if (zero_q)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) S.q[ks] = __builtin_bit_cast(bf16x8, u32x4{0u, 0u, 0u, 0u});
        S.mref = 0.f; S.lsum = 0.f;                               // (the accumulators are defined by the first PV product)
        S.ext = hh == 0 ? sh_pack2(-0.f, 1.0f) : 0u;
    };
    // this wave's next target, scanning forward from (head hi, pass pi, segment s0): the (head, pass[, segment]) at which it has
    // queries.  Global role: passes with a global segment; local role: segments with keys in passes that carry local chunks.
    int nh = -1, np_ = 0, nsg = 0, mark = 0;
    auto find_next = [&](int hi, int pi, int s0) {
        nh = -1;
        for (; hi < hpb; ++hi, pi = 0, s0 = 0)
            for (; pi < npass; ++pi, s0 = 0) {
                const int pinf = p_info(pi);
                if (!local) {
                    const int gs = (pinf & 255) - 1;
                    if (gs >= 0 && rows_of(gs, (pinf >> 8) & 255) > 0) { nh = hi; np_ = pi; nsg = gs; return; }
                } else {
                    const int c = (pinf >> 16) - 1;
                    if (c < 0) continue;
                    for (int s = s0; s < fus; ++s)
                        if (st.kl(s) > 0 && rows_of(s, c) > 0) { nh = hi; np_ = pi; nsg = s; return; }
                }
            }
    };
    auto chunk_of = [&](int pi) { const int pinf = p_info(pi); return local ? (pinf >> 16) - 1 : (pinf >> 8) & 255; };

    // ---- prologue: the first Q rows and the first SH_D tiles
    find_next(0, 0, 0);
    if (nh >= 0) issue_q(nsg, chunk_of(np_), h0 + nh);
    mark = vm;
    for (int i = 0; i < (VAR == 1 ? SH_NS : SH_D) && i < G; ++i) issue_ring();

    bool act = false, fresh = false;
    int t = 0, pi = 0, hi = 0;
    constexpr bool ST = MODE == 3;
    unsigned long long tt[16], tl = 0, t_loop = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) tt[i] = 0;
    if (ST) { tl = sh_now(); t_loop = tl; }
    int stage = 0;                                                  // ring stage of tile g
    for (int g = 0; g < G; ++g) {
        if (VAR != 1 || (g & 1) == 0) {
            if (VAR == 2) { if (wave >= 12) sh_wait_vm(vm - (nfl >= 3 ? q0 : nfl == 2 ? q1 : q2)); --nfl; }     // my four pieces of tile g: the oldest in flight
            else if (wave == 12 + (g & 3)) sh_wait_vm(vm - myseq);     // the pieces of tile g were mine to fetch
            if (VAR == 1 && g + 1 < G && wave == 12 + ((g + 1) & 3)) sh_wait_vm(vm - myseq);
            SH_T(0);
            __builtin_amdgcn_s_barrier();
            SH_T(1);
            if (VAR != 1) { if (lj < G) issue_ring(); }
            else if (g >= 2) { if (lj < G) issue_ring(); if (lj < G) issue_ring(); }   // into the stages of the pair just finished
            if (VAR == 1 && SH_SKEW > 0 && wave >= 4 && wave < 8) __builtin_amdgcn_s_sleep(SH_SKEW);   // experiment: phase offset
                                                                    // between the two global waves of a SIMD (units of 64 cycles)
            if (SH_SLEEP_CLASS >= 0 && (wave >> 2) == SH_SLEEP_CLASS) __builtin_amdgcn_s_sleep(8);   // slack probe: 512 idle cycles per
                                                                    // step in ONE wave class (0: waves 0-3 ... 3: 12-15, the loaders)
        }
        const int tinf = t_info(t), kn = tinf & 255, sg = (tinf >> 8) & 255, fl = tinf >> 16;
        const int h = h0 + hi;
        SH_T(2);
        // slot switch: a global wave at the first tile of its pass, a local wave at the first tile of its segment
        if (nh == hi && np_ == pi && (local ? ((fl & 1) && nsg == sg) : t == 0)) {
            unsigned long long tl2 = tl;
            sh_wait_vm(vm - mark);
            if (ST) { __builtin_amdgcn_sched_barrier(0); const unsigned long long n_ = sh_now(); tt[12] += n_ - tl2; tl2 = n_; __builtin_amdgcn_sched_barrier(0); }
            activate(nsg, chunk_of(pi), !local && nsg != fus);
            act = true; fresh = true;
            __builtin_amdgcn_sched_barrier(0);
            if (ST) { const unsigned long long n_ = sh_now(); tt[13] += n_ - tl2; tl2 = n_; __builtin_amdgcn_sched_barrier(0); }
            if (local) find_next(hi, pi, sg + 1); else find_next(hi, pi + 1, 0);
            if (ST) { __builtin_amdgcn_sched_barrier(0); const unsigned long long n_ = sh_now(); tt[14] += n_ - tl2; tl2 = n_; __builtin_amdgcn_sched_barrier(0); }
            if (nh >= 0) { issue_q(nsg, chunk_of(np_), h0 + nh); mark = vm; }
        }
        SH_T(3);
        if (act) {
            unsigned kext[2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) kext[kb] = hh == 0 ? sh_pack2(1.0f, (32 * kb + r >= kn) ? -1.0e30f : 0.f) : 0u;
            if (MODE != 1) sh_attend<ST>(S, fresh, ringK[0], ringV[0], stage * 4096, ad, kext, hh, tt, tl);
            fresh = false;
            if (ST) tt[9] += 1;
            if (local ? (fl & 2) != 0 : t == ntile - 1) { sh_finish(S, p, h, r, hh); act = false; vm += 5; }   // 4 row stores + lse
            SH_T(7);
        }
        if (++stage == SH_NS) stage = 0;
        if (++t == ntile) { t = 0; if (++pi == npass) { pi = 0; ++hi; } }
    }
    if (ST) {
        const unsigned w = blockIdx.x * 16 + wave;
        if (lane == 0 && w < SH_STAMP_WAVES) {
            tt[8] = sh_now() - t_loop; tt[10] = (unsigned long long)G; tt[11] = 1;
#pragma unroll
            for (int i = 0; i < 16; ++i) g_sh_stamps[w][i] = tt[i];
        }
    }
}

// ------------------------------------------------------------------------------------------------------ backward: dK, dV
// Key-stationary counterpart of the forward kernel.  One 12-wave workgroup per sample walks its heads; a wave owns 32 keys
// ("key block": K and V fragments in registers, dK^T / dV^T accumulators in registers -- 168 VGPRs, three waves per SIMD) and
// the workgroup sweeps the 64-row query tiles that may attend the 12 key blocks of the current PASS.  Q / dO tiles and their
// row constants (-lse in the log2 domain, -delta: planes 1 and 2 of the workspace, written by the dQ kernel) arrive by
// LDS-DMA into a 3-stage ring, fetched once per pass by a rotating loader wave; the key blocks of the NEXT pass are prefetched
// into wave-private staging.  Per (64 queries x 32 keys): S = Q K~^T and dP = dO V^T with the row constants as the initial
// accumulators (the query sits on the accumulator register index, so the seed is a 16-float vector read from LDS), P = exp2(S),
// dS = P o dP', then dV^T += dO^T P and dK^T += Q^T dS with the transposed operands read straight from the row-major images:
// 32 MFMAs per ~100 VALU instructions.  No per-element masks: padded query rows are zero rows with zero constants (the DMA's
// range check), padded keys only pollute their own never-stored columns.
#define SD_NS 3                 // ring stages
#define SD_D 2                  // tiles in flight ahead of the consumer
#define SD_W 12                 // waves = key blocks per pass
#define SD_MAXB 64              // key blocks per sample (one lane each)
#define SD_MAXS 64              // ring steps per head (one lane each)

template <int MODE>
__global__ __launch_bounds__(SD_W * 64) void mha_sh_dkdv_kernel(MhaDesc p, int hpb) {
    __shared__ __attribute__((aligned(1024))) bf16 ringQ[SD_NS][4096];
    __shared__ __attribute__((aligned(1024))) bf16 ringO[SD_NS][4096];
    __shared__ __attribute__((aligned(1024))) float ringL[SD_NS][128];       // [0..63] -lse2, [64..127] -delta of the tile's rows
    __shared__ __attribute__((aligned(1024))) bf16 kst[SD_W][2][32 * 64];    // wave-private K / V staging of the next pass's key block
    __shared__ int kb_row_s[SD_MAXB], kb_info_s[SD_MAXB];                    // key blocks (read at pass switches only: not worth two VGPRs)

    const int tid = threadIdx.x, lane = tid & 63, wave = sh_uni(tid >> 6), r = lane & 31, hh = lane >> 5;
    const int hgroups = p.H / hpb;
    const int b = blockIdx.x / hgroups, h0 = (blockIdx.x % hgroups) * hpb;
    const int nseg = p.nseg, fus = nseg - 1;
    ShSeg st; st.load(p, b, lane);
    // ---- schedule of this sample (registers, one lane per entry)
    // key blocks: first row, n | seg << 8.   steps: first query row, n | seg << 8 | mode << 12 | pass << 16 | last-of-pass << 24
    // mode 1: ordinary tile, 2: fully masked query rows attending every key uniformly (P = 1 / keys, dS = 0), 3: no-op filler
    int kb_row = 0, kb_info = 0, st_row = 0, st_info = 0, NB = 0, nsteps = 0, npass = 0;
    {
        for (int s = 0; s < nseg; ++s) {
            const int L = st.kl(s), nb = (L + 31) >> 5, j = lane - NB;
            if (j >= 0 && j < nb) { kb_row = st.ks(s) + 32 * j; kb_info = min(32, L - 32 * j) | (s << 8); }
            NB += nb;
        }
        NB = min(NB, SD_MAXB);
        npass = (NB + SD_W - 1) / SD_W;
        if (wave == 0) { kb_row_s[lane] = kb_row; kb_info_s[lane] = kb_info; }
        for (int ps = 0; ps < npass; ++ps) {
            int segmask = 0;
            for (int w = 0; w < SD_W; ++w) {
                const int kbi = SD_W * ps + w;
                if (kbi < NB) segmask |= 1 << ((__builtin_amdgcn_readlane(kb_info, kbi) >> 8) & 15);
            }
            const int first = nsteps;
            for (int sq = 0; sq < nseg; ++sq) {
                const int QL = st.ql(sq);
                if (QL == 0) continue;
                int mode = 0;
                if (sq == fus) mode = 1;
                else if (st.kl(sq) > 0) mode = (segmask >> sq) & 1;
                else if (p.empty_mode == 0) mode = 2;
                if (!mode) continue;
                const int nt = (QL + 63) >> 6, j = lane - nsteps;
                if (j >= 0 && j < nt) { st_row = st.qs(sq) + 64 * j; st_info = min(64, QL - 64 * j) | (sq << 8) | (mode << 12) | (ps << 16); }
                nsteps += nt;
            }
            if (nsteps == first) {                                 // a pass without queries still writes its (zero) gradients
                if (lane == nsteps) { st_row = 0; st_info = (3 << 12) | (ps << 16); }
                ++nsteps;
            }
            if (lane == nsteps - 1) st_info |= 1 << 24;
        }
        nsteps = min(nsteps, SD_MAXS);
    }
    __syncthreads();
    if (NB == 0) return;
    auto s_row = [&](int i) { return __builtin_amdgcn_readlane(st_row, i); };
    auto s_info = [&](int i) { return __builtin_amdgcn_readlane(st_info, i); };

    const bf16* qg = reinterpret_cast<const bf16*>(p.q);
    const bf16* dog = reinterpret_cast<const bf16*>(p.dout);
    const bf16* kg = reinterpret_cast<const bf16*>(p.k);
    const bf16* vg = reinterpret_cast<const bf16*>(p.v);
    const int qsb = (int)p.q_stride * 2, dosb = (int)p.do_stride * 2, ksb = (int)p.k_stride * 2, vsb = (int)p.v_stride * 2;
    // (DMA source offsets are rebuilt from the lane id at every issue, behind an opaque zero: kept across the tile loop they were
    // what the register allocator spilled)
    auto dma_voff = [&](int row_bytes, int z) { const int pr = (lane >> 3) + z; return pr * row_bytes + 16 * ((lane & 7) ^ sh_f(pr)); };
    ShAddr ad; ad.init(lane);
    const float cq = p.scale * SH_LOG2E;
    const int G = hpb * nsteps;
    int vm = 0, myseq = 0, mark = 0;
    // ring: loader of step j is wave 8 + (j & 3) -- the waves that idle while the fusion keys' pass runs on eight waves
    int lj = 0, li = 0, lh = 0, lstage = 0;
    auto issue_ring = [&]() {
        if (MODE != 2 && wave == 8 + (lj & 3)) {
            const long row0 = s_row(li); const int n = s_info(li) & 255, h = h0 + lh;
            const bf16* qb_ = qg + row0 * p.q_stride + h * 64;
            const bf16* ob_ = dog + row0 * p.do_stride + h * 64;
            int z = 0;
            asm volatile("" : "+s"(z));
            const int qv = dma_voff(qsb, z), ov = dma_voff(dosb, z);
#pragma unroll
            for (int pc = 0; pc < 8; ++pc) {
                sh_dma(qb_, n, qsb, qv ^ (16 * (pc & 1)), 8 * pc * qsb, &ringQ[lstage][pc * 512]);
                sh_dma(ob_, n, dosb, ov ^ (16 * (pc & 1)), 8 * pc * dosb, &ringO[lstage][pc * 512]);
            }
            // row constants: 64 floats each, 4 bytes per lane
            const float* lp = p.delta + ((long)p.H + h) * p.stat_stride + row0;
            const float* dp = p.delta + (2L * p.H + h) * p.stat_stride + row0;
            sh_dma4(lp, n, &ringL[lstage][0], lane);
            sh_dma4(dp, n, &ringL[lstage][64], lane);
            vm += 18; myseq = vm;
        }
        ++lj;
        if (++lstage == SD_NS) lstage = 0;
        if (++li == nsteps) { li = 0; ++lh; }
    };
    // key-block staging of pass ps, head h: this wave's 32 keys -> kst[wave][0 = K, 1 = V]
    auto kb_of = [&](int ps) { return SD_W * ps + wave; };
    auto issue_kv = [&](int ps, int h) {
        const int kbi = kb_of(ps);
        const long row0 = sh_uni(kb_row_s[kbi]); const int n = sh_uni(kb_info_s[kbi]) & 255;
        int z = 0;
        asm volatile("" : "+s"(z));
        const int kv_ = dma_voff(ksb, z), vv_ = dma_voff(vsb, z);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            sh_dma(kg + row0 * p.k_stride + h * 64, n, ksb, kv_ ^ (16 * (j & 1)), 8 * j * ksb, &kst[wave][0][8 * j * 64]);
            sh_dma(vg + row0 * p.v_stride + h * 64, n, vsb, vv_ ^ (16 * (j & 1)), 8 * j * vsb, &kst[wave][1][8 * j * 64]);
        }
        vm += 8;
    };

    bf16x8 kf[4], vf[4];
    f32x16 dk[2], dv[2];
    int my_seg = -1, my_row = 0, my_n = 0;
    bool have = false;

    // ---- prologue
    int nxt_ps = 0, nxt_h = 0;                                      // next (pass, head) whose key block is staged
    bool nxt_valid = kb_of(0) < NB;
    if (nxt_valid) issue_kv(0, h0);
    mark = vm;
    for (int i = 0; i < SD_D && i < G; ++i) issue_ring();

    int si = 0, hi = 0, stage = 0, cur_ps = -1;
    for (int g = 0; g < G; ++g) {
        if (wave == 8 + (g & 3)) sh_wait_vm(vm - myseq);
        __builtin_amdgcn_s_barrier();
        if (lj < G) issue_ring();
        const int inf = s_info(si), qn = inf & 255, sq = (inf >> 8) & 15, mode = (inf >> 12) & 15, ps = (inf >> 16) & 255, last = inf >> 24;
        const int h = h0 + hi;
        if (ps != cur_ps) {                                        // first step of a pass: take over the staged key block
            cur_ps = ps;
            have = kb_of(ps) < NB;
            if (have) {
                int z = 0;
                asm volatile("" : "+s"(z));
                const int kbi = kb_of(ps);
                const int ki = sh_uni(kb_info_s[kbi]);
                my_row = sh_uni(kb_row_s[kbi]); my_n = ki & 255; my_seg = (ki >> 8) & 15;
                sh_wait_vm(vm - mark);
                const int kbase = ad.krow0 + z;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const bf16x8 raw = sh_ld8(&kst[wave][0][0] + (kbase ^ (16 * ks)));
#pragma unroll
                    for (int j = 0; j < 8; ++j) kf[ks][j] = (bf16)((float)raw[j] * cq);
                    vf[ks] = sh_ld8(&kst[wave][1][0] + (kbase ^ (16 * ks)));
                }
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int i = 0; i < 16; ++i) { dk[d][i] = 0.f; dv[d][i] = 0.f; }
            }
            __builtin_amdgcn_sched_barrier(0);
            // stage the key block of the next pass (or of pass 0 of the next head)
            nxt_ps = ps + 1; nxt_h = hi;
            if (nxt_ps == npass) { nxt_ps = 0; ++nxt_h; }
            nxt_valid = nxt_h < hpb && kb_of(nxt_ps) < NB;
            if (nxt_valid) { issue_kv(nxt_ps, h0 + nxt_h); mark = vm; }
        }
        const bool part = have && (mode == 2 || (mode == 1 && (sq == fus || sq == my_seg)));
        if (part && MODE != 1) {
            const bf16* Qs = ringQ[0]; const bf16* Os = ringO[0];
            const int sel = stage * 4096;
            const int rbase = ad.krow0 + sel, tbase = ad.tr00 + sel;
            const float* Ls = &ringL[stage][0];
            f32x16 z16;
#pragma unroll
            for (int i = 0; i < 16; ++i) z16[i] = 0.f;
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                // S first, packed to bf16 P as soon as it is exponentiated, THEN the dP chain: 24 live accumulator registers
                // instead of 32 (the kernel sits on the 168-VGPR line; the price is 16 shifts to widen P again for dS = P o dP')
                bf16x8 pb[2], dsb[2];
                {
                    f32x16 sacc;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x4 a = *reinterpret_cast<const f32x4*>(Ls + 32 * qb + 8 * j + 4 * hh);
#pragma unroll
                        for (int i = 0; i < 4; ++i) sacc[4 * j + i] = a[i];
                    }
                    if (mode == 1)
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) sacc = sh_mma(sh_ld8(Qs + 2048 * qb + (rbase ^ (16 * ks))), kf[ks], sacc);
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                        for (int j = 0; j < 8; ++j) pb[s2][j] = (bf16)sh_exp2(sacc[8 * s2 + j]);   // (uniform rows: 2^(-lse2) = 1 / keys)
                }
                if (mode == 1) {
                    f32x16 dpacc;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x4 c = *reinterpret_cast<const f32x4*>(Ls + 64 + 32 * qb + 8 * j + 4 * hh);
#pragma unroll
                        for (int i = 0; i < 4; ++i) dpacc[4 * j + i] = c[i];
                    }
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) dpacc = sh_mma(sh_ld8(Os + 2048 * qb + (rbase ^ (16 * ks))), vf[ks], dpacc);
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                        for (int j = 0; j < 8; ++j) dsb[s2][j] = (bf16)((float)pb[s2][j] * dpacc[8 * s2 + j]);
                } else {
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) dsb[s2] = __builtin_bit_cast(bf16x8, u32x4{0u, 0u, 0u, 0u});     // uniform rows: dS = 0
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const int ro = 64 * (32 * qb + 16 * s2);
#pragma unroll
                    for (int d = 0; d < 2; ++d) {
                        dv[d] = sh_mma(sh_tr(Os + ro, tbase ^ (32 * d), tbase ^ (32 * d + 520)), pb[s2], dv[d]);
                        dk[d] = sh_mma(sh_tr(Qs + ro, tbase ^ (32 * d), tbase ^ (32 * d + 520)), dsb[s2], dk[d]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (last && have) {                                        // last step of the pass: this key block's gradients are complete
            const bool valid = r < my_n;
            bf16* dkp = reinterpret_cast<bf16*>(p.dk) + (long)(my_row + r) * p.dk_stride + h * 64 + 8 * hh;
            bf16* dvp = reinterpret_cast<bf16*>(p.dv) + (long)(my_row + r) * p.dv_stride + h * 64 + 8 * hh;
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int i = 0; i < 4; i += 2) {
                    unsigned a0 = sh_pack2(dk[d][4 * i] * p.scale, dk[d][4 * i + 1] * p.scale), a1 = sh_pack2(dk[d][4 * i + 2] * p.scale, dk[d][4 * i + 3] * p.scale);
                    unsigned b0 = sh_pack2(dk[d][4 * i + 4] * p.scale, dk[d][4 * i + 5] * p.scale), b1 = sh_pack2(dk[d][4 * i + 6] * p.scale, dk[d][4 * i + 7] * p.scale);
                    auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false); a0 = r0[0]; b0 = r0[1];
                    auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false); a1 = r1[0]; b1 = r1[1];
                    if (valid) *reinterpret_cast<u32x4*>(dkp + 32 * d + 8 * i) = u32x4{a0, a1, b0, b1};
                    unsigned c0 = sh_pack2(dv[d][4 * i], dv[d][4 * i + 1]), c1 = sh_pack2(dv[d][4 * i + 2], dv[d][4 * i + 3]);
                    unsigned e0 = sh_pack2(dv[d][4 * i + 4], dv[d][4 * i + 5]), e1 = sh_pack2(dv[d][4 * i + 6], dv[d][4 * i + 7]);
                    auto r2 = __builtin_amdgcn_permlane32_swap(c0, e0, false, false); c0 = r2[0]; e0 = r2[1];
                    auto r3 = __builtin_amdgcn_permlane32_swap(c1, e1, false, false); c1 = r3[0]; e1 = r3[1];
                    if (valid) *reinterpret_cast<u32x4*>(dvp + 32 * d + 8 * i) = u32x4{c0, c1, e0, e1};
                }
            vm += 8;
        }
        if (++stage == SD_NS) stage = 0;
        if (++si == nsteps) { si = 0; ++hi; cur_ps = -1; }
    }
}

// ------------------------------------------------------------------------------------------------------ backward: dQ (+ row constants)
// Query-stationary like the forward kernel: one 12-wave workgroup per sample walks its heads; waves 0-7 hold the "global"
// queries of the pass (fusion chunk of 256 rows), waves 8-11 the "local" queries of the segment being swept (chunk of 128
// rows); K/V tiles arrive once per (sample, head, pass) by LDS-DMA (ring of 3, loader = a rotating local wave).  Per
// (32 queries x 64 keys):  S^T = K Q~^T and dP^T = V dO^T, each seeded through one extra k-step with the row constant split
// into two bf16 terms (-lse2 = hi + lo to 2^-17: [1, 1] on the key side, [hi, lo] on the query side) so that only exp2 and one
// multiply remain per score; dS^T = P^T o dP'^T; dQ^T += K^T dS^T with K^T read transposed from the same row-major image.
// Padded keys are zero rows of the images (range check of the DMA): they produce finite P and contribute K^T dS = 0.
// A slot's operands (Q, dO, and O for delta = rowsum(dO o O)) are staged by LDS-DMA in two steps one segment / pass ahead:
// first O and dO -> delta, then Q into O's place (12 waves x 8 KB of staging + the ring fit the 160 KB of LDS; a third
// buffer would not).  The kernel also writes the planes of the workspace the dK/dV kernel reads: delta, -lse2, -delta.
#define SQ_NS 3
#define SQ_D 2
#define SQ_W 12
#define SQ_MAXT 64
#define SQ_MAXP 64

__device__ __forceinline__ void sq_zero_rows(const MhaDesc& p, long row0, int n, int h, int tid) {
    const long plane = (long)p.H * p.stat_stride;
    for (int i = tid; i < n * 8; i += SQ_W * 64) {
        const int rr = i >> 3, c = i & 7;
        *reinterpret_cast<u32x4*>(reinterpret_cast<bf16*>(p.dq) + (row0 + rr) * p.dq_stride + h * 64 + 8 * c) = u32x4{0u, 0u, 0u, 0u};
        if (c == 0) {
            const long at = (long)h * p.stat_stride + row0 + rr;
            p.delta[at] = 0.f; p.delta[at + plane] = -p.lse[at] * SH_LOG2E; p.delta[at + 2 * plane] = 0.f;
        }
    }
}
// (hi, lo) bf16 split of x, packed for the query side of the seed k-step
__device__ __forceinline__ unsigned sq_split(float x) {
    const float hi = sh_bf16_round(x);
    return sh_pack2(hi, x - hi);
}

template <int MODE>
__global__ __launch_bounds__(SQ_W * 64) void mha_sh_dq_kernel(MhaDesc p, int hpb) {
    __shared__ __attribute__((aligned(1024))) bf16 ringK[SQ_NS][4096];
    __shared__ __attribute__((aligned(1024))) bf16 ringV[SQ_NS][4096];
    __shared__ __attribute__((aligned(1024))) bf16 stA[SQ_W][32 * 64];       // wave-private: O rows, then Q rows
    __shared__ __attribute__((aligned(1024))) bf16 stB[SQ_W][32 * 64];       // wave-private: dO rows
    __shared__ __attribute__((aligned(256))) float stL[SQ_W][64];            // wave-private: lse of the slot's rows

    const int tid = threadIdx.x, lane = tid & 63, wave = sh_uni(tid >> 6), r = lane & 31, hh = lane >> 5;
    const bool local = wave >= 8;
    const int qw = local ? wave - 8 : wave, CH = local ? 128 : 256;    // 32-query block of a 256-row (global) / 128-row (local) chunk
    const int hgroups = p.H / hpb;
    const int b = blockIdx.x / hgroups, h0 = (blockIdx.x % hgroups) * hpb;
    const int nseg = p.nseg, fus = nseg - 1;
    ShSeg st; st.load(p, b, lane);
    // tiles: first key row, n | seg << 8 | (first | last << 1) << 16.   passes: (global seg + 1) | chunk << 4 | has-local << 12 |
    // segment mask of the tiles to sweep << 16 (a pass without global queries only sweeps the segments that still have local rows)
    int tile_row = 0, tile_info = 0, pass_info = 0, ntile = 0, npass = 0;
    {
        int acc = 0;
        for (int s = 0; s < nseg; ++s) {
            const int L = st.kl(s), nt = (L + 63) >> 6, j = lane - acc;
            if (j >= 0 && j < nt) { tile_row = st.ks(s) + 64 * j; tile_info = min(64, L - 64 * j) | (s << 8) | (((j == 0 ? 1 : 0) | (j == nt - 1 ? 2 : 0)) << 16); }
            acc += nt;
        }
        ntile = min(acc, SQ_MAXT);
        int nch = (st.ql(fus) + 255) >> 8, allmask = 0;
        for (int s = 0; s < nseg; ++s) if (st.kl(s) > 0) allmask |= 1 << s;
        for (int s = 0; s < fus; ++s) if (st.kl(s) > 0) nch = max(nch, (st.ql(s) + 127) >> 7);
        nch = min(nch, SQ_MAXP);
        if (lane < nch) {
            int lmask = 0;
            for (int s = 0; s < fus; ++s) if (st.kl(s) > 0 && st.ql(s) > 128 * lane) lmask |= 1 << s;
            const int gs = 256 * lane < st.ql(fus) ? fus : -1;
            pass_info = (gs + 1) | (lane << 4) | ((lmask != 0 ? 1 : 0) << 12) | ((gs >= 0 ? allmask : lmask) << 16);
        }
        npass = nch;
    }
    auto t_row = [&](int t) { return __builtin_amdgcn_readlane(tile_row, t); };
    auto t_info = [&](int t) { return __builtin_amdgcn_readlane(tile_info, t); };
    auto p_info = [&](int pi) { return __builtin_amdgcn_readlane(pass_info, pi); };
    // rows whose gradient is zero by construction: no keys at all, an empty key segment (zeros or the uniform rows: dS = 0)
    for (int hi = 0; hi < hpb; ++hi)
        for (int s = 0; s < nseg; ++s) {
            const int QL = st.ql(s);
            if (QL > 0 && (ntile == 0 || (s < fus && st.kl(s) == 0))) sq_zero_rows(p, st.qs(s), QL, h0 + hi, tid);
        }
    if (ntile == 0 || npass == 0) return;
    // first valid tile at or after t in pass pi (ntile: none)
    auto seek = [&](int t, int pi) {
        const int m = p_info(pi) >> 16;
        while (t < ntile && !((m >> ((t_info(t) >> 8) & 15)) & 1)) ++t;
        return t;
    };
    int G = 0;
    for (int pi = 0; pi < npass; ++pi) { const int m = p_info(pi) >> 16; for (int t = 0; t < ntile; ++t) G += (m >> ((t_info(t) >> 8) & 15)) & 1; }
    G *= hpb;

    const bf16* qg = reinterpret_cast<const bf16*>(p.q);
    const bf16* og = reinterpret_cast<const bf16*>(p.o);
    const bf16* dog = reinterpret_cast<const bf16*>(p.dout);
    const bf16* kg = reinterpret_cast<const bf16*>(p.k);
    const bf16* vg = reinterpret_cast<const bf16*>(p.v);
    const int qsb = (int)p.q_stride * 2, osb = (int)p.o_stride * 2, dosb = (int)p.do_stride * 2, ksb = (int)p.k_stride * 2, vsb = (int)p.v_stride * 2;
    auto dma_voff = [&](int row_bytes, int z) { const int pr = (lane >> 3) + z; return pr * row_bytes + 16 * ((lane & 7) ^ sh_f(pr)); };
    ShAddr ad; ad.init(lane);
    const float cq = p.scale * SH_LOG2E;
    int vm = 0, myseq = 0;
    // ---- ring loader (rotating over the local-role waves)
    int lt = 0, lp = 0, lh = 0, lj = 0, lstage = 0;
    lt = seek(0, 0);
    auto issue_ring = [&]() {
        if (MODE != 2 && wave == 8 + (lj & 3)) {
            const long row0 = t_row(lt); const int n = t_info(lt) & 255, h = h0 + lh;
            int z = 0;
            asm volatile("" : "+s"(z));
            const int kv_ = dma_voff(ksb, z), vv_ = dma_voff(vsb, z);
            const bf16* kb_ = kg + row0 * p.k_stride + h * 64;
            const bf16* vb_ = vg + row0 * p.v_stride + h * 64;
#pragma unroll
            for (int pc = 0; pc < 8; ++pc) {
                sh_dma(kb_, n, ksb, kv_ ^ (16 * (pc & 1)), 8 * pc * ksb, &ringK[lstage][pc * 512]);
                sh_dma(vb_, n, vsb, vv_ ^ (16 * (pc & 1)), 8 * pc * vsb, &ringV[lstage][pc * 512]);
            }
            vm += 16; myseq = vm;
        }
        ++lj;
        if (++lstage == SQ_NS) lstage = 0;
        lt = seek(lt + 1, lp);
        while (lt >= ntile && lh < hpb) { if (++lp == npass) { lp = 0; ++lh; } if (lh < hpb) lt = seek(0, lp); }
    };
    auto rows_of = [&](int s, int chunk) { return min(32, st.ql(s) - CH * chunk - 32 * qw); };
    // this wave's next target (head, pass, segment)
    int nh = -1, np_ = 0, nsg = 0;
    auto find_next = [&](int hi, int pi, int s0) {
        nh = -1;
        for (; hi < hpb; ++hi, pi = 0, s0 = 0)
            for (; pi < npass; ++pi, s0 = 0) {
                const int pinf = p_info(pi), c = (pinf >> 4) & 255;
                if (!local) {
                    const int gs = (pinf & 15) - 1;
                    if (gs >= 0 && rows_of(gs, c) > 0) { nh = hi; np_ = pi; nsg = gs; return; }
                } else {
                    if (!((pinf >> 12) & 1)) continue;
                    for (int s = s0; s < fus; ++s)
                        if (st.kl(s) > 0 && rows_of(s, c) > 0) { nh = hi; np_ = pi; nsg = s; return; }
                }
            }
    };
    auto chunk_of = [&](int pi) { return (p_info(pi) >> 4) & 255; };
    // two-step operand prefetch of the next target: (1) O -> stA, dO -> stB, lse -> stL; (2) delta from (1), then Q -> stA
    int pf = 0, mark1 = 0, mark2 = 0, pf_iter = 0;
    float delta_next = 0.f;
    auto row0_of = [&](int s, int chunk) { return (long)st.qs(s) + CH * chunk + 32 * qw; };
    auto issue_stage1 = [&](int g) {
        const int c = chunk_of(np_), nq = rows_of(nsg, c), h = h0 + nh;
        const long row0 = row0_of(nsg, c);
        int z = 0;
        asm volatile("" : "+s"(z));
        const int ov = dma_voff(osb, z), dv_ = dma_voff(dosb, z);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            sh_dma(og + row0 * p.o_stride + h * 64, nq, osb, ov ^ (16 * (j & 1)), 8 * j * osb, &stA[wave][8 * j * 64]);
            sh_dma(dog + row0 * p.do_stride + h * 64, nq, dosb, dv_ ^ (16 * (j & 1)), 8 * j * dosb, &stB[wave][8 * j * 64]);
        }
        sh_dma4(p.lse + (long)h * p.stat_stride + row0, nq, &stL[wave][0], lane);
        vm += 9; mark1 = vm; pf = 1; pf_iter = g;
    };
    auto do_stage2 = [&]() {
        int z = 0;
        asm volatile("" : "+s"(z));
        sh_wait_vm(vm - mark1);
        const int base = ad.krow0 + z;
        float dpart = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 o8 = sh_ld8(&stA[wave][0] + (base ^ (16 * ks))), d8 = sh_ld8(&stB[wave][0] + (base ^ (16 * ks)));
#pragma unroll
            for (int j = 0; j < 8; ++j) dpart += (float)o8[j] * (float)d8[j];
        }
        delta_next = sh_swap_sum(dpart);
        __builtin_amdgcn_sched_barrier(0);
        const int c = chunk_of(np_), nq = rows_of(nsg, c), h = h0 + nh;
        const long row0 = row0_of(nsg, c);
        const int qv = dma_voff(qsb, z);
#pragma unroll
        for (int j = 0; j < 4; ++j) sh_dma(qg + row0 * p.q_stride + h * 64, nq, qsb, qv ^ (16 * (j & 1)), 8 * j * qsb, &stA[wave][8 * j * 64]);
        vm += 4; mark2 = vm; pf = 2;
    };

    bf16x8 q[4], dO[4];
    f32x16 dq[2];
    unsigned extS = 0, extD = 0;
    int my_row = 0, my_nq = 0;

    // ---- prologue
    find_next(0, 0, 0);
    if (nh >= 0) issue_stage1(-1);
    for (int i = 0; i < SQ_D && i < G; ++i) issue_ring();

    bool act = false;
    int t = seek(0, 0), pi = 0, hi = 0, stage = 0;
    for (int g = 0; g < G; ++g) {
        if (wave == 8 + (g & 3)) sh_wait_vm(vm - myseq);
        __builtin_amdgcn_s_barrier();
        if (lj < G) issue_ring();
        const int tinf = t_info(t), sg = (tinf >> 8) & 255, fl = tinf >> 16;
        const int h = h0 + hi;
        const bool first_of_pass = t == seek(0, pi);
        if (nh == hi && np_ == pi && (local ? ((fl & 1) && nsg == sg) : first_of_pass)) {       // slot switch
            if (pf == 1) do_stage2();
            int z = 0;
            asm volatile("" : "+s"(z));
            sh_wait_vm(vm - mark2);
            const int c = chunk_of(pi);
            my_nq = rows_of(nsg, c);
            my_row = (int)row0_of(nsg, c) + z + r;
            const int base = ad.krow0 + z;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 raw = sh_ld8(&stA[wave][0] + (base ^ (16 * ks)));
#pragma unroll
                for (int j = 0; j < 8; ++j) q[ks][j] = (bf16)((float)raw[j] * cq);
                dO[ks] = sh_ld8(&stB[wave][0] + (base ^ (16 * ks)));
            }
            const float lse2 = stL[wave][r] * SH_LOG2E, delta = delta_next;
            extS = hh == 0 ? sq_split(-lse2) : 0u;
            extD = hh == 0 ? sq_split(-delta) : 0u;
            if (r < my_nq && hh == 0) {                            // the row constants the dK/dV kernel reads
                const long at = (long)h * p.stat_stride + my_row, plane = (long)p.H * p.stat_stride;
                p.delta[at] = delta; p.delta[at + plane] = -lse2; p.delta[at + 2 * plane] = -delta;
            }
            vm += 3;
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int i = 0; i < 16; ++i) dq[d][i] = 0.f;
            act = true; pf = 0;
            __builtin_amdgcn_sched_barrier(0);
            if (local) find_next(hi, pi, sg + 1); else find_next(hi, pi + 1, 0);
            if (nh >= 0) issue_stage1(g);
        } else if (pf == 1 && g > pf_iter) {
            do_stage2();
        }
        if (act) {
            if (MODE != 1) {
                const bf16* Kst = ringK[0]; const bf16* Vst = ringV[0];
                const int sel = stage * 4096, kbase = ad.krow0 + sel, vbase = ad.tr00 + sel;
                int zc = 0;                                        // (opaque zero: keeps the constant operand out of the loop-invariant set,
                asm volatile("" : "+s"(zc));                       //  where it was spilled and reloaded from scratch every tile)
                const bf16x8 kone = sh_ext((hh == 0 ? 0x3f803f80u : 0u) | (unsigned)zc), qeS = sh_ext(extS), qeD = sh_ext(extD);
                f32x16 z16;
#pragma unroll
                for (int i = 0; i < 16; ++i) z16[i] = 0.f;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    bf16x8 pb[2], dsb[2];
                    {
                        f32x16 sacc = sh_mma(kone, qeS, z16);
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) sacc = sh_mma(sh_ld8(Kst + 2048 * kb + (kbase ^ (16 * ks))), q[ks], sacc);
#pragma unroll
                        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                            for (int j = 0; j < 8; ++j) pb[s2][j] = (bf16)sh_exp2(sacc[8 * s2 + j]);
                    }
                    {
                        f32x16 dpacc = sh_mma(kone, qeD, z16);
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) dpacc = sh_mma(sh_ld8(Vst + 2048 * kb + (kbase ^ (16 * ks))), dO[ks], dpacc);
#pragma unroll
                        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                            for (int j = 0; j < 8; ++j) dsb[s2][j] = (bf16)((float)pb[s2][j] * dpacc[8 * s2 + j]);
                    }
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        const bf16* kb_ = Kst + 64 * (32 * kb + 16 * s2);
#pragma unroll
                        for (int d = 0; d < 2; ++d) dq[d] = sh_mma(sh_tr(kb_, vbase ^ (32 * d), vbase ^ (32 * d + 520)), dsb[s2], dq[d]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // last tile of the slot: a local wave's segment ends, a global wave's pass ends
            const int tn = seek(t + 1, pi);
            if (local ? (fl & 2) != 0 : tn >= ntile) {
                const bool valid = r < my_nq;
                bf16* dqp = reinterpret_cast<bf16*>(p.dq) + (long)my_row * p.dq_stride + h * 64 + 8 * hh;
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int i = 0; i < 4; i += 2) {
                        unsigned a0 = sh_pack2(dq[d][4 * i] * p.scale, dq[d][4 * i + 1] * p.scale), a1 = sh_pack2(dq[d][4 * i + 2] * p.scale, dq[d][4 * i + 3] * p.scale);
                        unsigned b0 = sh_pack2(dq[d][4 * i + 4] * p.scale, dq[d][4 * i + 5] * p.scale), b1 = sh_pack2(dq[d][4 * i + 6] * p.scale, dq[d][4 * i + 7] * p.scale);
                        auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false); a0 = r0[0]; b0 = r0[1];
                        auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false); a1 = r1[0]; b1 = r1[1];
                        if (valid) *reinterpret_cast<u32x4*>(dqp + 32 * d + 8 * i) = u32x4{a0, a1, b0, b1};
                    }
                act = false; vm += 4;
            }
        }
        if (++stage == SQ_NS) stage = 0;
        t = seek(t + 1, pi);
        while (t >= ntile && hi < hpb) { if (++pi == npass) { pi = 0; ++hi; } if (hi < hpb) t = seek(0, pi); }
    }
}

// ------------------------------------------------------------------------------------------------------ backward: dQ + dK + dV fused
// ONE key-stationary kernel for the whole backward (cdna_hip_programming.md, "Attention backward": five tile products -- S = Q K~^T and
// dP = dO V^T once, dV^T += dO^T P, dK^T += Q^T dS, dQ += dS K -- instead of the seven of the dQ + dK/dV kernel pair).  Built on
// mha_sh_dkdv_kernel: a wave owns 32 keys of the current pass (K / V fragments and dK^T / dV^T accumulators in registers) and the
// workgroup sweeps the 64-row query tiles that may attend them.  What dQ adds:
//   * dS sits with the KEY on the lane (the layout dK^T / dV^T contract over) while dQ contracts over keys: every wave writes its
//     64 q x 32 k block of dS, as [key][q], into one LDS image of the step (dsx, swizzled like the tiles); after a second workgroup
//     barrier waves 0..3 each own one 32 d x 32 q quarter of the tile's dQ^T and sum  K^T dS^T  over the key blocks that took part,
//     both operands by transposed reads -- the K rows from images that stay resident for the whole pass (kimg);
//   * the LDS this needs (64 KB of K images, double-buffered so the next pass's keys arrive under the current pass, + 32 KB of dS)
//     does not fit beside twelve waves' staging: EIGHT waves (256 keys per pass), V of the next pass prefetched into registers
//     instead of LDS (two waves per SIMD leave room), ring of three (Q, dO) tiles: 146 KB;
//   * a query tile met in several passes (the fusion queries see every key) carries its fp32 dQ^T partial from pass to pass through
//     a workspace slot private to the (tile, head): written by the first pass, prefetched as the accumulator's start value by the
//     later ones, the last pass stores bf16 -- same workgroup, program order, so the sum is bitwise reproducible without atomics;
//   * the row constants (-lse log2 e, -delta) come from a pre-pass (mha_rowconst_kernel: delta = rowsum(dO o O), one HBM pass over
//     O and dO) -- the dQ kernels used to produce them on the way.
// Loader waves of the (Q, dO) ring are 4..7: they skip the dQ phase, and waves 0..3 then never have ring DMA outstanding behind
// the plain loads of a partial (hipcc waits vmcnt(0) for those).
#define SB_NS 3
#define SB_D 2
#define SB_W 8
#define SB_MAXB 64
#define SB_MAXS 128

// One workgroup = 32 rows x H heads (H <= 8): a lane owns one 16-byte piece of a (row, head)'s 64 values of O and dO (fully coalesced
// reads: a row's heads are contiguous), eight lanes add up one (row, head), the 32 x H results go through LDS so that every plane is
// written in runs of 32 consecutive rows.
__global__ __launch_bounds__(256) void mha_rowconst_kernel(MhaDesc p) {
    __shared__ float res[8][33];
    const int t = threadIdx.x, c = t & 7, h = (t >> 3) & 7;
    const long plane = (long)p.H * p.stat_stride;
    for (int pass = 0; pass < 8; ++pass) {                           // 4 rows per pass: 256 threads = 4 rows x 8 heads x 8 pieces
        const int lr = 4 * pass + (t >> 6);
        const long row = (long)blockIdx.x * 32 + lr;
        float acc = 0.f;
        if (row < p.stat_stride && h < p.H) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16*>(p.o) + row * p.o_stride + h * 64 + 8 * c);
            const bf16x8 b = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16*>(p.dout) + row * p.do_stride + h * 64 + 8 * c);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += (float)a[j] * (float)b[j];
        }
        acc += __shfl_xor(acc, 1); acc += __shfl_xor(acc, 2); acc += __shfl_xor(acc, 4);
        if (c == 0) res[h][lr] = acc;
    }
    __syncthreads();
    const int hh = t >> 5, lr = t & 31;                              // 8 heads x 32 rows
    const long row = (long)blockIdx.x * 32 + lr;
    if (hh < p.H && row < p.stat_stride) {
        const long at = (long)hh * p.stat_stride + row;
        const float d = res[hh][lr];
        p.delta[at] = d; p.delta[at + plane] = -p.lse[at] * SH_LOG2E; p.delta[at + 2 * plane] = -d;
    }
}

// MODE (diagnostic builds, never the product path; wrong dQ): 1 = no dS exchange and no dQ phase (the dK / dV part alone on eight waves),
// 2 = dQ phase without the workspace traffic of the partials
// RC (variant 55, not the product): the row constants of a tile (-lse log2 e, -delta = -rowsum(dO o O)) are formed HERE, one step ahead,
// by the four loader waves while waves 0..3 run the dQ products, and written into the ring stage of the next step, instead of being read
// from the planes of the pre-pass mha_rowconst_kernel.  Saves the pre-pass (62 us) and costs the kernel 35 (255 VGPRs instead of 232, the
// loader waves reach the barrier later): -25 us per call in isolation, nothing measurable in the step.
template <int MODE, bool RC = false>
__global__ __launch_bounds__(SB_W * 64) void mha_sh_bwd_kernel(MhaDesc p, int hpb) {
    __shared__ __attribute__((aligned(1024))) bf16 ringQ[SB_NS][4096];
    __shared__ __attribute__((aligned(1024))) bf16 ringO[SB_NS][4096];
    __shared__ __attribute__((aligned(1024))) float ringL[SB_NS][128];       // [0..63] -lse2, [64..127] -delta of the tile's rows
    __shared__ __attribute__((aligned(1024))) bf16 kimg[2][SB_W][2048];      // K images of the current / the next pass (one 32-key block per wave)
    __shared__ __attribute__((aligned(1024))) bf16 dsx[SB_W * 2048];         // dS of the step: [32 w + key][64 q]
    __shared__ int kb_row_s[SB_MAXB], kb_info_s[SB_MAXB];

    const int tid = threadIdx.x, lane = tid & 63, wave = sh_uni(tid >> 6), r = lane & 31, hh = lane >> 5;
    const int hgroups = p.H / hpb;
    const int b = blockIdx.x / hgroups, h0 = (blockIdx.x % hgroups) * hpb;
    const int nseg = p.nseg, fus = nseg - 1;
    ShSeg st; st.load(p, b, lane);
    // ---- schedule of this sample (registers, one lane per entry)
    // key blocks: first row, n | seg << 8.
    // steps: first query row, tile slot, n | seg << 8 | mode << 12 | pass << 16 | last-of-pass << 24 | first pass of the tile << 25 | last pass << 26
    // mode 1: ordinary tile, 2: fully masked query rows attending every key uniformly (P = 1 / keys, dS = 0), 3: no-op filler
    // (the step table holds 2 x 64 entries: entry i lives on lane i & 63 of register set i >> 6)
    int kb_row = 0, kb_info = 0, NB = 0, nsteps = 0, npass = 0;
    int st_row[2] = {0, 0}, st_info[2] = {0, 0}, st_tid[2] = {0, 0};
    {
        // Order of the key blocks over the passes (SB_W per pass; any order is correct: a block's pass only decides which query tiles
        // it meets when).  A modality segment that would straddle a pass boundary although it fits one pass is pushed to the next pass
        // and the gap is filled with FUSION key blocks -- their queries' tiles are swept in every pass anyway, while a straddling
        // modality gets its query tiles swept twice, each time by a few waves only.
        const int FB = (st.kl(fus) + 31) >> 5;
        int fus_used = 0;
        auto place = [&](int s, int first, int count) {           // blocks first .. first + count - 1 of segment s -> slots NB ..
            const int j = lane - NB, L = st.kl(s);
            if (j >= 0 && j < count) { kb_row = st.ks(s) + 32 * (first + j); kb_info = min(32, L - 32 * (first + j)) | (s << 8); }
            NB += count;
        };
        for (int s = 0; s < fus; ++s) {
            const int nb = (st.kl(s) + 31) >> 5;
            if (nb == 0) continue;
            const int room = (SB_W - NB % SB_W) % SB_W;
            if (room > 0 && nb > room && nb <= SB_W) {
                const int pad = min(room, FB - fus_used);
                if (pad > 0) { place(fus, fus_used, pad); fus_used += pad; }
            }
            place(s, 0, nb);
        }
        if (FB > fus_used) place(fus, fus_used, FB - fus_used);
        NB = min(NB, SB_MAXB);
        npass = (NB + SB_W - 1) / SB_W;
        if (wave == 0) { kb_row_s[lane] = kb_row; kb_info_s[lane] = kb_info; }
        // first / last pass in which a segment's keys appear (4 bits each)
        unsigned firstp = 0, lastp = 0, seen = 0;
        for (int ps = 0; ps < npass; ++ps)
            for (int w = 0; w < SB_W; ++w) {
                const int kbi = SB_W * ps + w;
                if (kbi < NB) {
                    const int s = (__builtin_amdgcn_readlane(kb_info, kbi) >> 8) & 15;
                    if (!((seen >> s) & 1)) { seen |= 1u << s; firstp |= (unsigned)ps << (4 * s); }
                    lastp = (lastp & ~(15u << (4 * s))) | ((unsigned)ps << (4 * s));
                }
            }
        for (int ps = 0; ps < npass; ++ps) {
            int segmask = 0;
            for (int w = 0; w < SB_W; ++w) {
                const int kbi = SB_W * ps + w;
                if (kbi < NB) segmask |= 1 << ((__builtin_amdgcn_readlane(kb_info, kbi) >> 8) & 15);
            }
            const int first = nsteps;
            int tbase = 0;
            for (int sq = 0; sq < nseg; ++sq) {
                const int QL = st.ql(sq), nt = (QL + 63) >> 6;
                const int tb = tbase;
                tbase += nt;
                if (QL == 0) continue;
                int mode = 0;
                if (sq == fus) mode = 1;
                else if (st.kl(sq) > 0) mode = (segmask >> sq) & 1;
                else if (p.empty_mode == 0) mode = 2;
                if (!mode) continue;
                // passes of this tile's dQ: the fusion queries meet every pass, a modality's queries the passes that hold its keys
                const int fp = sq == fus ? 0 : (int)((firstp >> (4 * sq)) & 15), lp = sq == fus ? npass - 1 : (int)((lastp >> (4 * sq)) & 15);
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int j = lane + 64 * k - nsteps;
                    if (j >= 0 && j < nt) {
                        st_row[k] = st.qs(sq) + 64 * j; st_tid[k] = tb + j;
                        st_info[k] = min(64, QL - 64 * j) | (sq << 8) | (mode << 12) | (ps << 16) | ((ps == fp ? 1 : 0) << 25) | ((ps == lp ? 1 : 0) << 26);
                    }
                }
                nsteps += nt;
            }
            if (nsteps == first) {                                 // a pass without queries still writes its (zero) gradients
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    if (lane + 64 * k == nsteps) { st_row[k] = 0; st_tid[k] = 0; st_info[k] = (3 << 12) | (ps << 16); }
                ++nsteps;
            }
#pragma unroll
            for (int k = 0; k < 2; ++k)
                if (lane + 64 * k == nsteps - 1) st_info[k] |= 1 << 24;
        }
        nsteps = min(nsteps, SB_MAXS);
    }
    __syncthreads();
    // query rows whose dQ is zero by construction: no keys at all, or an empty key segment (zeros, or the uniform rows: dS = 0)
    for (int hi = 0; hi < hpb; ++hi)
        for (int s = 0; s < nseg; ++s) {
            const int QL = st.ql(s);
            if (QL > 0 && (NB == 0 || (s < fus && st.kl(s) == 0))) {
                const long row0 = st.qs(s);
                for (int i = tid; i < QL * 8; i += SB_W * 64)
                    *reinterpret_cast<u32x4*>(reinterpret_cast<bf16*>(p.dq) + (row0 + (i >> 3)) * p.dq_stride + (h0 + hi) * 64 + 8 * (i & 7)) = u32x4{0u, 0u, 0u, 0u};
            }
        }
    if (NB == 0) return;
    auto s_row = [&](int i) { return i < 64 ? __builtin_amdgcn_readlane(st_row[0], i) : __builtin_amdgcn_readlane(st_row[1], i - 64); };
    auto s_info = [&](int i) { return i < 64 ? __builtin_amdgcn_readlane(st_info[0], i) : __builtin_amdgcn_readlane(st_info[1], i - 64); };
    auto s_tid = [&](int i) { return i < 64 ? __builtin_amdgcn_readlane(st_tid[0], i) : __builtin_amdgcn_readlane(st_tid[1], i - 64); };

    const bf16* qg = reinterpret_cast<const bf16*>(p.q);
    const bf16* dog = reinterpret_cast<const bf16*>(p.dout);
    const bf16* kg = reinterpret_cast<const bf16*>(p.k);
    const bf16* vg = reinterpret_cast<const bf16*>(p.v);
    const int qsb = (int)p.q_stride * 2, dosb = (int)p.do_stride * 2, ksb = (int)p.k_stride * 2, vsb = (int)p.v_stride * 2;
    auto dma_voff = [&](int row_bytes, int z) { const int pr = (lane >> 3) + z; return pr * row_bytes + 16 * ((lane & 7) ^ sh_f(pr)); };
    ShAddr ad; ad.init(lane);
    const float cq = p.scale * SH_LOG2E;
    const int G = hpb * nsteps;
    int vm = 0, myseq = 0, mark = 0;
    // ring: loader of step j is wave 4 + (j & 3) -- the waves that skip the dQ phase
    int lj = 0, li = 0, lh = 0, lstage = 0;
    auto issue_ring = [&]() {
        if (wave == 4 + (lj & 3)) {
            const long row0 = s_row(li); const int n = s_info(li) & 255, h = h0 + lh;
            const bf16* qb_ = qg + row0 * p.q_stride + h * 64;
            const bf16* ob_ = dog + row0 * p.do_stride + h * 64;
            int z = 0;
            asm volatile("" : "+s"(z));
            const int qv = dma_voff(qsb, z), ov = dma_voff(dosb, z);
#pragma unroll
            for (int pc = 0; pc < 8; ++pc) {
                sh_dma(qb_, n, qsb, qv ^ (16 * (pc & 1)), 8 * pc * qsb, &ringQ[lstage][pc * 512]);
                sh_dma(ob_, n, dosb, ov ^ (16 * (pc & 1)), 8 * pc * dosb, &ringO[lstage][pc * 512]);
            }
            if (!RC) {
                const float* lp = p.delta + ((long)p.H + h) * p.stat_stride + row0;
                const float* dp = p.delta + (2L * p.H + h) * p.stat_stride + row0;
                sh_dma4(lp, n, &ringL[lstage][0], lane);
                sh_dma4(dp, n, &ringL[lstage][64], lane);
            }
            vm += RC ? 16 : 18; myseq = vm;
        }
        ++lj;
        if (++lstage == SB_NS) lstage = 0;
        if (++li == nsteps) { li = 0; ++lh; }
    };
    // row constants of step i of head index hx (RC), formed by waves 4..7 -- the loaders, which sit out the dQ phase: a lane holds one
    // 32-byte piece of a row of O and of dO (four lanes per row, 16 rows per wave) and the row's lse from rc_issue to rc_finish one step
    // later, both after the step's second barrier, while waves 0..3 run the dQ products
    u32x4 rc_o[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}}, rc_d[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
    unsigned rc_l = 0u;
    int rcseq = 0;
    const bool rc_wave = RC && wave >= 4;
    const int rc_row = 16 * (wave & 3) + (lane >> 2), rc_c = lane & 3;
    auto rc_issue = [&](int i, int hx) {
        const long row0 = s_row(i); const int n = s_info(i) & 255, h = h0 + hx;
        const int osb = (int)p.o_stride * 2;
        const bf16* ob = reinterpret_cast<const bf16*>(p.o) + row0 * p.o_stride + h * 64;
        const bf16* db = dog + row0 * p.do_stride + h * 64;
        rc_o[0] = sh_ld128_async(ob, n * osb, rc_row * osb + 32 * rc_c);
        rc_o[1] = sh_ld128_async(ob, n * osb, rc_row * osb + 32 * rc_c + 16);
        rc_d[0] = sh_ld128_async(db, n * dosb, rc_row * dosb + 32 * rc_c);
        rc_d[1] = sh_ld128_async(db, n * dosb, rc_row * dosb + 32 * rc_c + 16);
        rc_l = sh_ld32_async(p.lse + (long)h * p.stat_stride + row0, n * 4, rc_row * 4);
        vm += 5; rcseq = vm;
    };
    auto rc_finish = [&](int to_stage) {
        sh_wait_vm(vm - rcseq);
        sh_pin(rc_o[0], rc_d[0], rc_l);
        sh_pin(rc_o[1], rc_d[1], rc_l);
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const bf16x8 a = __builtin_bit_cast(bf16x8, rc_o[k]), d8 = __builtin_bit_cast(bf16x8, rc_d[k]);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += (float)a[j] * (float)d8[j];
        }
        acc = sh_add_dpp<0xB1>(acc);                              // lanes ^ 1, ^ 2 of the quad
        acc = sh_add_dpp<0x4E>(acc);
        if (rc_c == 0) {
            ringL[to_stage][rc_row] = -__builtin_bit_cast(float, rc_l) * SH_LOG2E;
            ringL[to_stage][64 + rc_row] = -acc;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // in LDS before this wave reaches the next step's barrier
    };
    auto kb_of = [&](int ps) { return SB_W * ps + wave; };
    // next pass's key block: K rows -> kimg[buf][wave] (LDS-DMA), V rows -> registers in the operand layout (rows >= n read as zero)
    bf16x8 vnext[4];
    auto issue_kv = [&](int ps, int h, int buf) {
        const int kbi = kb_of(ps);
        const long row0 = sh_uni(kb_row_s[kbi]); const int n = sh_uni(kb_info_s[kbi]) & 255;
        int z = 0;
        asm volatile("" : "+s"(z));
        const int kv_ = dma_voff(ksb, z);
#pragma unroll
        for (int j = 0; j < 4; ++j) sh_dma(kg + row0 * p.k_stride + h * 64, n, ksb, kv_ ^ (16 * (j & 1)), 8 * j * ksb, &kimg[buf][wave][8 * j * 64]);
        const bf16* vb = vg + row0 * p.v_stride + h * 64;
        const unsigned long long a = reinterpret_cast<unsigned long long>(vb);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
        bf16* ub = reinterpret_cast<bf16*>(((unsigned long long)hi << 32) | lo);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(ub, 0, sh_uni(n * vsb), 0x00020000);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            vnext[ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, (r + z) * vsb + 32 * ks + 16 * hh, 0, 0));
        vm += 8;
    };

    bf16x8 kf[4], vf[4];
    f32x16 dk[2], dv[2];
    int my_seg = -1, my_row = 0, my_n = 0;
    bool have = false;
    int nvalid = 0;                                                // key blocks of the current pass (a prefix of the waves)
    const int dq_d = wave & 1, dq_q = (wave >> 1) & 1;             // waves 0..3: their quarter of the dQ^T tile (32 d x 32 q)
    const int dsw = (32 * wave + r) * 64, dsf = sh_f(r);

    // ---- prologue
    int nxt_ps = 0, nxt_h = 0, cur_buf = 0, nxt_buf = 0;
    bool nxt_valid = kb_of(0) < NB;
    int rci = 0, rch = 0;                                          // the step whose constants are fetched next
    auto rc_issue_next = [&]() { rc_issue(rci, rch); if (++rci == nsteps) { rci = 0; ++rch; } };
    if (rc_wave) rc_issue_next();
    if (nxt_valid) issue_kv(0, h0, 0);
    mark = vm;
    for (int i = 0; i < SB_D && i < G; ++i) issue_ring();
    if (rc_wave) { rc_finish(0); if (G > 1) rc_issue_next(); }     // step 0's constants: published by the first barrier below

    int si = 0, hi = 0, stage = 0, cur_ps = -1;
    for (int g = 0; g < G; ++g) {
        if (wave == 4 + (g & 3)) sh_wait_vm(vm - myseq);
        __builtin_amdgcn_s_barrier();
        if (lj < G) issue_ring();
        const int inf = s_info(si), qn = inf & 255, sq = (inf >> 8) & 15, mode = (inf >> 12) & 15, ps = (inf >> 16) & 255;
        const bool last = (inf >> 24) & 1, tfirst = (inf >> 25) & 1, tlast = (inf >> 26) & 1;
        const int h = h0 + hi;
        if (ps != cur_ps) {                                        // first step of a pass: take over the staged key block
            cur_ps = ps;
            cur_buf = nxt_buf;
            have = kb_of(ps) < NB;
            if (have) {
                int z = 0;
                asm volatile("" : "+s"(z));
                const int kbi = kb_of(ps);
                const int ki = sh_uni(kb_info_s[kbi]);
                my_row = sh_uni(kb_row_s[kbi]); my_n = ki & 255; my_seg = (ki >> 8) & 15;
                sh_wait_vm(vm - mark);
                const int kbase = ad.krow0 + z;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const bf16x8 raw = sh_ld8(&kimg[cur_buf][wave][0] + (kbase ^ (16 * ks)));
#pragma unroll
                    for (int j = 0; j < 8; ++j) kf[ks][j] = (bf16)((float)raw[j] * cq);
                    vf[ks] = vnext[ks];
                }
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int i = 0; i < 16; ++i) { dk[d][i] = 0.f; dv[d][i] = 0.f; }
            }
            nvalid = min(SB_W, NB - SB_W * ps);                   // the pass's key blocks are waves 0 .. nvalid - 1
            __builtin_amdgcn_sched_barrier(0);
            // stage the key block of the next pass (or of pass 0 of the next head) into the other image buffer
            nxt_ps = ps + 1; nxt_h = hi;
            if (nxt_ps == npass) { nxt_ps = 0; ++nxt_h; }
            nxt_valid = nxt_h < hpb && kb_of(nxt_ps) < NB;
            nxt_buf = cur_buf ^ 1;
            if (nxt_valid) { issue_kv(nxt_ps, h0 + nxt_h, nxt_buf); mark = vm; }
        }
        const bool dq_wave = wave < 4 && mode == 1 && MODE != 1 && MODE != 4;      // MODE 4: dS exchange and barrier, no dQ products
        // dQ^T quarter of waves 0..3: starts from the tile's partial of the earlier passes (prefetched here, used after the barrier)
        f32x16 dqa;
        float* wsq = p.dq_ws + ((((long)b * p.max_qt + s_tid(si)) * p.H + h) * 4 + (2 * dq_d + dq_q)) * 1024 + 4 * lane;
        if (dq_wave && !tfirst && MODE == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 v4 = *reinterpret_cast<const f32x4*>(wsq + 256 * i);
                dqa[4 * i] = v4[0]; dqa[4 * i + 1] = v4[1]; dqa[4 * i + 2] = v4[2]; dqa[4 * i + 3] = v4[3];
            }
            vm += 4;
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) dqa[i] = 0.f;
        }
        const bool part = have && (mode == 2 || (mode == 1 && (sq == fus || sq == my_seg)));
        if (MODE != 1 && mode == 1 && have && !part) {
            // a key block that does not take part in this tile contributes dS = 0: the dQ phase then runs ONE branch-free chain over
            // every block of the pass (its K rows are finite: staged keys, zero rows beyond a block's end)
#pragma unroll
            for (int c = 0; c < 8; ++c) *reinterpret_cast<u32x2*>(dsx + dsw + 8 * (c ^ dsf) + 4 * hh) = u32x2{0u, 0u};
        }
        if (part) {
            const bf16* Qs = ringQ[0]; const bf16* Os = ringO[0];
            const int sel = stage * 4096;
            const int rbase = ad.krow0 + sel, tbase = ad.tr00 + sel;
            const float* Ls = &ringL[stage][0];
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                bf16x8 pb[2], dsb[2];
                {
                    f32x16 sacc;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x4 a = *reinterpret_cast<const f32x4*>(Ls + 32 * qb + 8 * j + 4 * hh);
#pragma unroll
                        for (int i = 0; i < 4; ++i) sacc[4 * j + i] = a[i];
                    }
                    if (mode == 1)
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) sacc = sh_mma(sh_ld8(Qs + 2048 * qb + (rbase ^ (16 * ks))), kf[ks], sacc);
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                        for (int j = 0; j < 8; ++j) pb[s2][j] = (bf16)sh_exp2(sacc[8 * s2 + j]);
                }
                if (mode == 1) {
                    f32x16 dpacc;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x4 c = *reinterpret_cast<const f32x4*>(Ls + 64 + 32 * qb + 8 * j + 4 * hh);
#pragma unroll
                        for (int i = 0; i < 4; ++i) dpacc[4 * j + i] = c[i];
                    }
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) dpacc = sh_mma(sh_ld8(Os + 2048 * qb + (rbase ^ (16 * ks))), vf[ks], dpacc);
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) dsb[s2][j] = (bf16)((float)pb[s2][j] * dpacc[8 * s2 + j]);
                        // dS for the dQ phase: element j of this lane is query 32 qb + 16 s2 + 8 (j >> 2) + 4 hh + (j & 3) of key (wave, r):
                        // two 8-byte pieces of row 32 wave + r of the [key][q] image
                        if (MODE != 1) {
                            const u32x4 w4 = __builtin_bit_cast(u32x4, dsb[s2]);
                            const int c0 = 4 * qb + 2 * s2;
                            *reinterpret_cast<u32x2*>(dsx + dsw + 8 * (c0 ^ dsf) + 4 * hh) = u32x2{w4[0], w4[1]};
                            *reinterpret_cast<u32x2*>(dsx + dsw + 8 * ((c0 + 1) ^ dsf) + 4 * hh) = u32x2{w4[2], w4[3]};
                        }
                    }
                } else {
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) dsb[s2] = __builtin_bit_cast(bf16x8, u32x4{0u, 0u, 0u, 0u});     // uniform rows: dS = 0
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const int ro = 64 * (32 * qb + 16 * s2);
#pragma unroll
                    for (int d = 0; d < 2; ++d) {
                        dv[d] = sh_mma(sh_tr(Os + ro, tbase ^ (32 * d), tbase ^ (32 * d + 520)), pb[s2], dv[d]);
                        dk[d] = sh_mma(sh_tr(Qs + ro, tbase ^ (32 * d), tbase ^ (32 * d + 520)), dsb[s2], dk[d]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (last && have) {                                        // last step of the pass: this key block's gradients are complete
            const bool valid = r < my_n;
            bf16* dkp = reinterpret_cast<bf16*>(p.dk) + (long)(my_row + r) * p.dk_stride + h * 64 + 8 * hh;
            bf16* dvp = reinterpret_cast<bf16*>(p.dv) + (long)(my_row + r) * p.dv_stride + h * 64 + 8 * hh;
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int i = 0; i < 4; i += 2) {
                    unsigned a0 = sh_pack2(dk[d][4 * i] * p.scale, dk[d][4 * i + 1] * p.scale), a1 = sh_pack2(dk[d][4 * i + 2] * p.scale, dk[d][4 * i + 3] * p.scale);
                    unsigned b0 = sh_pack2(dk[d][4 * i + 4] * p.scale, dk[d][4 * i + 5] * p.scale), b1 = sh_pack2(dk[d][4 * i + 6] * p.scale, dk[d][4 * i + 7] * p.scale);
                    auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false); a0 = r0[0]; b0 = r0[1];
                    auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false); a1 = r1[0]; b1 = r1[1];
                    if (valid) *reinterpret_cast<u32x4*>(dkp + 32 * d + 8 * i) = u32x4{a0, a1, b0, b1};
                    unsigned c0 = sh_pack2(dv[d][4 * i], dv[d][4 * i + 1]), c1 = sh_pack2(dv[d][4 * i + 2], dv[d][4 * i + 3]);
                    unsigned e0 = sh_pack2(dv[d][4 * i + 4], dv[d][4 * i + 5]), e1 = sh_pack2(dv[d][4 * i + 6], dv[d][4 * i + 7]);
                    auto r2 = __builtin_amdgcn_permlane32_swap(c0, e0, false, false); c0 = r2[0]; e0 = r2[1];
                    auto r3 = __builtin_amdgcn_permlane32_swap(c1, e1, false, false); c1 = r3[0]; e1 = r3[1];
                    if (valid) *reinterpret_cast<u32x4*>(dvp + 32 * d + 8 * i) = u32x4{c0, c1, e0, e1};
                }
            vm += 8;
        }
        // ---- dQ phase: every wave's dS block is in dsx, the pass's K images have been resident since its first step
        if (MODE != 1) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        if (rc_wave && g + 1 < G) {
            rc_finish(stage + 1 == SB_NS ? 0 : stage + 1);         // the next step's stage: its readers finished two barriers ago
            if (g + 2 < G) rc_issue_next();                         // a whole step in flight
        }
        if (dq_wave) {
            const bf16* Kc = &kimg[cur_buf][0][0];
            const int ka = ad.tr00 ^ (32 * dq_d), kb2 = ad.tr00 ^ (32 * dq_d + 520), sa = ad.tr00 ^ (32 * dq_q), sb2 = ad.tr00 ^ (32 * dq_q + 520);
            if (nvalid == SB_W) {                                  // the usual pass: every wave holds a key block -- one unrolled chain of 16
#pragma unroll
                for (int w = 0; w < SB_W; ++w)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const int ro = 64 * (32 * w + 16 * ks);
                        dqa = sh_mma(sh_tr(Kc + ro, ka, kb2), sh_tr(dsx + ro, sa, sb2), dqa);
                    }
            } else {
                for (int w = 0; w < nvalid; ++w)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const int ro = 64 * (32 * w + 16 * ks);
                        dqa = sh_mma(sh_tr(Kc + ro, ka, kb2), sh_tr(dsx + ro, sa, sb2), dqa);
                    }
            }
            if (!tlast && MODE == 2) {
                asm volatile("" :: "v"(dqa));                      // keep the products live
            } else if (!tlast) {                                   // carry the partial to the tile's next pass
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    *reinterpret_cast<f32x4*>(wsq + 256 * i) = f32x4{dqa[4 * i], dqa[4 * i + 1], dqa[4 * i + 2], dqa[4 * i + 3]};
                vm += 4;
            } else {
                const int qr = 32 * dq_q + r;
                const bool valid = qr < qn;
                bf16* dqp = reinterpret_cast<bf16*>(p.dq) + (long)(s_row(si) + qr) * p.dq_stride + h * 64 + 32 * dq_d + 8 * hh;
                // (a tile of <= 32 rows leaves the upper query half without a valid row: those waves issue no store, and `vm` must count
                // exactly the vector-memory instructions issued -- an over-count would let a later counted wait return early)
                if (qn > 32 * dq_q) {
#pragma unroll
                    for (int i = 0; i < 4; i += 2) {
                        unsigned a0 = sh_pack2(dqa[4 * i] * p.scale, dqa[4 * i + 1] * p.scale), a1 = sh_pack2(dqa[4 * i + 2] * p.scale, dqa[4 * i + 3] * p.scale);
                        unsigned b0 = sh_pack2(dqa[4 * i + 4] * p.scale, dqa[4 * i + 5] * p.scale), b1 = sh_pack2(dqa[4 * i + 6] * p.scale, dqa[4 * i + 7] * p.scale);
                        auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false); a0 = r0[0]; b0 = r0[1];
                        auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false); a1 = r1[0]; b1 = r1[1];
                        if (valid) *reinterpret_cast<u32x4*>(dqp + 8 * i) = u32x4{a0, a1, b0, b1};
                    }
                    vm += 2;
                }
            }
        }
        if (++stage == SB_NS) stage = 0;
        if (++si == nsteps) { si = 0; ++hi; cur_ps = -1; }
    }
}

// ------------------------------------------------------------------------------------------------------ host side
static int sh_heads_per_block(int B, int H, int req) {
    // one workgroup per CU when the batch allows it: a block walks `hpb` heads of its sample (hpb divides H)
    if (req > 0 && req <= H && H % req == 0) return req;      // per-call override (MhaDesc::hpb_req: parity tests of the head walk)
    int hpb = H;
    while (hpb > 1 && (long)B * (H / hpb) < 256) {
        int d = hpb - 1;
        while (d > 1 && H % d) --d;
        hpb = d;
    }
    return hpb;
}

bool mha_sh_applicable(const MhaDesc& d) {
    return d.max_q_rows >= 128 && d.max_k_rows / 64 + d.nseg <= SH_MAXT && d.k_stride == d.v_stride && d.nseg <= MAXSEG &&
           d.max_q_rows / 256 + 2 * d.nseg <= SH_MAXP;
}

int mha_sh_fwd(const MhaDesc& d, int mode, hipStream_t st) {
    if (d.max_k_rows / 64 + d.nseg > SH_MAXT || d.k_stride != d.v_stride || d.nseg > MAXSEG) return MMAE_ERR_ARG;
    const int hpb = sh_heads_per_block(d.B, d.H, d.hpb_req);
    const dim3 grid(d.B * (d.H / hpb)), blk(1024);
    if (mode == 1) MMAE_LAUNCH((mha_sh_fwd_kernel<1, 0>), grid, blk, 0, st, d, hpb);
    else if (mode == 2) MMAE_LAUNCH((mha_sh_fwd_kernel<2, 0>), grid, blk, 0, st, d, hpb);
#if MMAE_DIAG
    else if (mode == 3) MMAE_LAUNCH((mha_sh_fwd_kernel<3, 0>), grid, blk, 0, st, d, hpb);
#else
    else if (mode == 3) return MMAE_ERR_ARG;
#endif
    else if (mode == 10) MMAE_LAUNCH((mha_sh_fwd_kernel<0, 1>), grid, blk, 0, st, d, hpb);     // two tiles per barrier
    else if (mode == 11) MMAE_LAUNCH((mha_sh_fwd_kernel<1, 1>), grid, blk, 0, st, d, hpb);
    else if (mode == 12) MMAE_LAUNCH((mha_sh_fwd_kernel<2, 1>), grid, blk, 0, st, d, hpb);
    else if (mode == 4) MMAE_LAUNCH((mha_sh_fwd_kernel<4, 0>), grid, blk, 0, st, d, hpb);      // diagnostic: sixteen global waves
    else if (mode == 20) MMAE_LAUNCH((mha_sh_fwd_kernel<0, 2>), grid, blk, 0, st, d, hpb);     // four loader waves
    else if (mode == 21) MMAE_LAUNCH((mha_sh_fwd_kernel<1, 2>), grid, blk, 0, st, d, hpb);
    else MMAE_LAUNCH((mha_sh_fwd_kernel<0, 0>), grid, blk, 0, st, d, hpb);
    MMAE_CHECK_LAUNCH();
    return MMAE_OK;
}

// key-stationary dK / dV (mha_sh_dkdv_kernel): needs planes 1 (-lse * log2 e) and 2 (-delta) of the workspace d.delta, which the
// dQ kernel of mha_bf16.hip writes next to delta itself (plane 0)
bool mha_sh_dkdv_supported(const MhaDesc& d) {
    const int nb = d.max_k_rows / 32 + d.nseg, passes = (nb + SD_W - 1) / SD_W;
    return nb <= SD_MAXB && d.nseg <= MAXSEG && passes * (d.max_q_rows / 64 + d.nseg + 1) <= SD_MAXS;
}

int mha_sh_dkdv(const MhaDesc& d, int mode, hipStream_t st) {
    if (!mha_sh_dkdv_supported(d)) return MMAE_ERR_ARG;
    const int hpb = sh_heads_per_block(d.B, d.H, d.hpb_req);
    const dim3 grid(d.B * (d.H / hpb)), blk(SD_W * 64);
    if (mode == 1) MMAE_LAUNCH(mha_sh_dkdv_kernel<1>, grid, blk, 0, st, d, hpb);
    else if (mode == 2) MMAE_LAUNCH(mha_sh_dkdv_kernel<2>, grid, blk, 0, st, d, hpb);
    else MMAE_LAUNCH(mha_sh_dkdv_kernel<0>, grid, blk, 0, st, d, hpb);
    MMAE_CHECK_LAUNCH();
    return MMAE_OK;
}

// fused backward (mha_sh_bwd_kernel) + its row-constant pre-pass
bool mha_sh_fused_supported(const MhaDesc& d) {
    const int nb = d.max_k_rows / 32 + d.nseg, passes = (nb + SB_W - 1) / SB_W;
    return nb <= SB_MAXB && passes <= 15 && d.nseg <= MAXSEG && d.H <= 8 && passes * (d.max_q_rows / 64 + d.nseg + 1) <= SB_MAXS && d.dq_ws != nullptr &&
           d.max_qt >= d.max_q_rows / 64 + d.nseg;
}

int mha_sh_bwd_fused(const MhaDesc& d, int mode, hipStream_t st) {
    if (!mha_sh_fused_supported(d)) return MMAE_ERR_ARG;
    // mode 0: product (row constants from the pre-pass mha_rowconst_kernel); 1 / 2 / 4: its diagnostic builds; 3: without the pre-pass (stale
    // planes, timing only); 5: row constants formed inside the kernel by the loader waves (RC; correct results: isolated -25 us per call at
    // the bench shape, +-0 in the step -- DESIGN.md section 4)
    if (mode != 3 && mode != 5) {
        MMAE_LAUNCH(mha_rowconst_kernel, dim3((unsigned)((d.stat_stride + 31) / 32)), dim3(256), 0, st, d);
        MMAE_CHECK_LAUNCH();
    }
    const int hpb = sh_heads_per_block(d.B, d.H, d.hpb_req);
    const dim3 grid(d.B * (d.H / hpb)), blk(SB_W * 64);
    if (mode == 1) MMAE_LAUNCH((mha_sh_bwd_kernel<1, false>), grid, blk, 0, st, d, hpb);
    else if (mode == 2) MMAE_LAUNCH((mha_sh_bwd_kernel<2, false>), grid, blk, 0, st, d, hpb);
    else if (mode == 4) MMAE_LAUNCH((mha_sh_bwd_kernel<4, false>), grid, blk, 0, st, d, hpb);
    else if (mode == 5) MMAE_LAUNCH((mha_sh_bwd_kernel<0, true>), grid, blk, 0, st, d, hpb);
    else MMAE_LAUNCH((mha_sh_bwd_kernel<0, false>), grid, blk, 0, st, d, hpb);
    MMAE_CHECK_LAUNCH();
    return MMAE_OK;
}

// query-stationary dQ (mha_sh_dq_kernel); also produces the workspace planes (delta, -lse2, -delta)
bool mha_sh_dq_supported(const MhaDesc& d) {
    return d.max_k_rows / 64 + d.nseg <= SQ_MAXT && d.nseg <= MAXSEG && d.max_q_rows / 128 + 1 <= SQ_MAXP;
}

int mha_sh_dq(const MhaDesc& d, int mode, hipStream_t st) {
    if (!mha_sh_dq_supported(d)) return MMAE_ERR_ARG;
    const int hpb = sh_heads_per_block(d.B, d.H, d.hpb_req);
    const dim3 grid(d.B * (d.H / hpb)), blk(SQ_W * 64);
    if (mode == 1) MMAE_LAUNCH(mha_sh_dq_kernel<1>, grid, blk, 0, st, d, hpb);
    else if (mode == 2) MMAE_LAUNCH(mha_sh_dq_kernel<2>, grid, blk, 0, st, d, hpb);
    else MMAE_LAUNCH(mha_sh_dq_kernel<0>, grid, blk, 0, st, d, hpb);
    MMAE_CHECK_LAUNCH();
    return MMAE_OK;
}
