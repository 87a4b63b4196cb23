// Dense projections C[M, N] = A[M, K] . W[N, K]^T (bf16 in / bf16 out, fp32 accumulate) as a hand-written gfx950 kernel -- the forward and
// input-gradient GEMMs of the encoder (reference: the nn.Linear calls of DSI-MM/zorro_utils.py:181-182,192 (to_q / to_kv / to_out) and
// :125-127 (FeedForward), autograd's dX = dY . W for the same layers) -- with an optional GEGLU epilogue (:115-118) for FeedForward[1].
//
// Structure (cdna_hip_programming.md section 5, "The 256^2 8-phase template", written from that description; go / no-go history and the
// diagnostic builds: tools/probes/gemm8p_probe.hip): 256 x 256 x 64 tiles, 8 waves (2 along M x 4 along N, 128 x 64 of C per wave in 128
// accumulator registers), v_mfma_f32_16x16x32_bf16; operands by LDS-DMA (buffer_load ... lds, 1 KiB per wave-instruction) into two 64 KB tile
// buffers, swizzled on the SOURCE address + the same XOR on the b128 fragment reads (conflict-free); 4 phases per K-tile (one 64 x 32
// quadrant of the wave's tile x K 64 = 16 MFMAs), each = { ds_read the quadrant's fragments, ONE staging step (2 DMA per lane), [counted
// s_waitcnt vmcnt(N), never 0 in the loop], raw s_barrier, MFMAs under s_setprio(1), raw s_barrier }; waves 4..7 run one barrier behind
// waves 0..3, so on every SIMD one wave issues MFMAs while its partner reads LDS and issues DMA.
// PERSISTENT: 256 workgroups (one per CU: 128 KB of LDS each) walk the tiles in an XCD-aware order (the 32 workgroups of an XCD hold a few
// A row-panels x up to 8 W panels at a time); the staging schedule runs on into the next tile's K-tiles 0 / 1, and the epilogue ROLLS by
// halves: rows 0..63 of a wave's tile are final after phase 1 of the last K-tile and are stored in its phases 2 / 3, rows 64..127 in phases
// 0 / 1 of the next tile's first K-tile.  The product is oriented so that an MFMA column is an output column and the W tile sits in LDS in a
// permuted row order (free: an LDS-DMA lane fetches any source row): lane j of a 16-lane group then holds 4 CONSECUTIVE columns of one C
// row and the group writes one whole 128-byte line -- every store instruction = 4 complete lines -- with the nontemporal policy (C is
// written once and read by a later kernel: streaming lines must not evict the A / W panels from L2; worth +18 % on the FF1 shape).
// Measured in isolation (M 163 840, N 4096, K 768, uniform random operands): 1303 TFLOP/s against 1206-1228 for the tuned library GEMM.
//
// Staging granule = a quarter tile (64 rows x 64 k = 8 KB = one DMA per lane), two per phase.  Liveness (K-tile t lives in buffer t & 1):
//     A-lo quarters (rows 0..63 of both 128-row halves): read in phase 0           -> restaged in phase 2 with K-tile t+2
//     W half 0 / half 1                                : read in phases 0, 1        -> restaged in phase 3 (t+2) / phase 0 of t+1 (t+2)
//     A-hi quarters                                    : read in phase 2            -> restaged in phase 1 of t+1 (t+2)
//   a region is restaged >= 2 phases after its last read, every read is >= 1 phase after the counted wait that retires its DMA, and 3-4
//   staging steps stay in flight across the barriers.  vmcnt counts DMA loads AND stores in issue order: the counts below include the rolling
//   epilogue's stores (too low a count only waits longer, too high a count would read a tile early).
#include "common.hpp"
#include "mmae_hip.h"
#include <atomic>

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define LDS_AS __attribute__((address_space(3)))
#define GM_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define GM_STORE_AUX 2            // nt

namespace {

__device__ __forceinline__ unsigned gm_pack2(float lo, float hi) {          // one v_cvt_pk_bf16_f32
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}
// GELU of the FF1 + GEGLU epilogue (exact-erf GELU of the reference: F.gelu, DSI-MM/zorro_utils.py:115-118).  Three forms, chosen at build time:
//   GM_GELU_FAST 1 (default since round 5): x * sigma(x (c0 + c1 x^2 + c2 x^4)), coefficients fitted (minimax on |x| <= 9, x^2 clamped there) to
//     |gelu - x Phi(x)| <= 2.6e-5 for EVERY bf16 input: the product g = gelu(gate) * val is rounded to bf16 (relative 2^-9) right after, and the
//     bf16-rounded gelu differs from the correctly rounded erf form for 18 of the 17 745 bf16 gates with |gelu| >= 0.02, by one ulp -- asserted
//     ON THE DEVICE over all 65 280 finite gates by tests/test_gpu_gemm.py::test_gemm_geglu_epilogue_gelu_over_every_bf16_gate.  9 instructions
//     / 11 issue slots per gate.  A documented deviation: the fastest form measured.
//   GM_GELU_LUT 1 (round 6, `make EXTRA=-DGM_GELU_LUT=1`): x * Phi(x) with Phi LOOKED UP -- the gate the product is formed from is a bf16 value,
//     i.e. one of 65 536 numbers; Phi of the 2 x 2048 of them with |x| in [2^-12, 16) sits in LDS as fp32 (16 KB behind the tile buffers,
//     csrc/gelu_phi_table.inc from tools/gen_phi_table.py: erfc in double, rounded once), the index is clamped (|x| < 2^-12: 0.5 +- 1e-4;
//     |x| >= 16: exactly 1 / 0).  EXACT: 0 of the 17 745 gates differ from the correctly rounded erf form.  6 integer instructions + one
//     ds_read_b32 + one multiply per gate (8 slots) -- and SLOWER: the 64-address gathers cost the LDS more than the 3 slots save
//     (epilogue 200.5 vs 190.6 us per launch at the FF1 shape, step 155.5 vs 154.9 ms, three alternations on one box).  Not the default.
//   GM_GELU_FAST 0: x * Phi(x), Phi from one exp2 + one rcp (Abramowitz-Stegun 7.1.26, |err| <= 1.5e-7) -- the approximation of
//     rowops.hip's geglu kernels (fp32 mode and the un-fused path); 13 instructions / 15 slots; +0.8 ms per step (round 5).
#ifndef GM_GELU_LUT
#define GM_GELU_LUT 0
#endif
#ifndef GM_GELU_FAST
#define GM_GELU_FAST 1
#endif
#define GM_PHI_LDS 131072          // byte offset of the Phi table in LDS (EPI 1 only: 128 KB of tiles + 16 KB)
#if GM_GELU_LUT
__device__ const unsigned gm_phi_table[4096] = {
#include "gelu_phi_table.inc"
};
// Phi of the bf16 value in bits [SH, SH + 16) of w
template <int SH>
__device__ __forceinline__ float gm_phi_lut(const char* lds, unsigned w) {
    const int t = (int)__builtin_amdgcn_ubfe(w, SH, 15) - 0x3980;
    const unsigned tc = (unsigned)min(max(t, 0), 2047);                      // one v_med3_i32
    const unsigned idx = ((w >> (SH + 4)) & 0x800u) | tc;                 // sign bit -> bit 11: the negative half of the table
    return *reinterpret_cast<const float*>(lds + GM_PHI_LDS + (idx << 2));
}
#endif
__device__ __forceinline__ float gm_gelu(float x) {
#if GM_GELU_FAST
    // the polynomial is clamped through its argument x^2 (not x): beyond |x| = 9 the exponent x * p(81) keeps growing linearly, so the
    // factor goes to exactly 1 (x -> +big: g == x) and exactly 0 (x -> -big: g == -0; with x itself clamped a gate of -1e30 came out
    // as -2.3e18).  Same instruction count as the v_med3 form; identical results on |x| <= 9.
    const float x2 = fminf(x * x, 81.f);
    // -log2(e) * {1.59501577, 7.40112920e-2, -7.03033577e-4}
    const float p = fmaf(fmaf(1.01426305e-3f, x2, -1.06775723e-1f), x2, -2.30112136f);
    const float e = __builtin_amdgcn_exp2f(x * p);
    return x * __builtin_amdgcn_rcpf(1.f + e);
#else
    const float e = __builtin_amdgcn_exp2f(-0.72134752044448170f * x * x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.23164189f, fabsf(x), 1.f));
    float poly = fmaf(0.5f * 1.061405429f, t, 0.5f * -1.453152027f);
    poly = fmaf(poly, t, 0.5f * 1.421413741f);
    poly = fmaf(poly, t, 0.5f * -0.284496736f);
    poly = fmaf(poly, t, 0.5f * 0.254829592f);
    const float half_erfc = poly * (t * e);
    return fmaf(-fabsf(x), half_erfc, fmaxf(x, 0.f));
#endif
}

#ifndef GM_EPI_DIAG
#define GM_EPI_DIAG 0             // 1 / 2 / 3: timing-only builds of the GEGLU epilogue (no GELU / no product store / h in 8-byte stores) for
                                  // tools/probes/geglu_epi_probe.py -- wrong results by construction
#endif
struct GemmArgs {
    const bf16* A; const bf16* W; bf16* C; bf16* G;      // G: GEGLU product (EPI 1), else unused
    int M, N, K;                                         // EPI 1: N = F output pairs (W has 2 F rows: val rows [0, F), gate rows [F, 2 F))
    int lda, ldw, ldc, ldg;                              // elements
    int ntm, ntn;                                        // tiles: ceil(M / 256) x N / 256 (EPI 1: F / 128)
};

#ifndef GM_GN
#define GM_GN 8                   // N-tiles of a chunk: the W panels an XCD keeps L2-resident while its A panels stream (A/B knob)
#endif
struct TileXY { int tm, tn; };
// Tiles of one XCD group x (= workgroup % 8: the workgroups that share an L2 under round-robin placement -- speed only): M-panels
// [m0, m1), walked in chunks of up to 8 N-tiles (a chunk's W panels stay L2-resident while the A panels stream); local tile index i.
__device__ __forceinline__ TileXY tile_of(int x, int i, int ntm, int ntn) {
    const int q = ntm / 8, r = ntm % 8;
    const int m0 = x * q + (x < r ? x : r), rows = q + (x < r ? 1 : 0);
    const int GN = ntn < GM_GN ? ntn : GM_GN, nc = ntn / GN, rem = ntn - nc * GN, full = nc * rows * GN;
    TileXY t;
    if (i < full) { const int c = i / (rows * GN), rr = i - c * rows * GN; t.tm = m0 + rr / GN; t.tn = c * GN + rr % GN; }
    else { const int rr = i - full; t.tm = m0 + rr / rem; t.tn = nc * GN + rr % rem; }
    return t;
}
__device__ __forceinline__ int tiles_of_xcd(int x, int ntm, int ntn) { return (ntm / 8 + (x < ntm % 8 ? 1 : 0)) * ntn; }

// Buffer descriptors are rebuilt at every use from (kernel-argument pointer, record count): a descriptor held live costs 4 SGPRs, and the
// kernel addresses six of them (A, W, C, G of the current tile, A / W of the next, C / G of the previous one) -- kept whole they spill.
struct Ctx {
    unsigned nA, nW;                     // record counts (bytes) of A / W; 0 for a tile that does not exist (the DMA then zero-fills dead LDS)
    int sA, sW;                          // scalar byte offsets of this wave's first piece row in A / W
};
struct Out { unsigned nC, nG; int c, g; };     // record counts of C / G (0: stores dropped) and the wave's tile origin (bytes) in them
__device__ __forceinline__ __amdgpu_buffer_rsrc_t gm_rsrc(const void* p, unsigned n) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, n, 0x00020000);
}
struct Lane {
    char* lds;
    int voffA, voffW;                    // per-lane DMA source offsets (8 rows x 128 B piece)
    int rdA[2], rdW[2];                  // fragment read offsets, k-step 0 / 1
    int voffC, voffG;                    // per-lane byte offsets inside C / G
    int lda2, ldw2, ldc2, ldg2;          // bytes per row
    int wgrp;                            // W rows per 64-row LDS group step in the source (EPI 0: 64, EPI 1: 32), in bytes of W: wgrp * ldw2
    int gate_off;                        // EPI 1: byte offset of the gate columns inside a row of C (F * 2)
    const bf16* A; const bf16* W; bf16* C; bf16* G;
};

__device__ __forceinline__ void gm_dma(const void* base, unsigned nrec, int voff, int soff, char* dst) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(gm_rsrc(base, nrec), (LDS_AS void*)dst, 16, voff, __builtin_amdgcn_readfirstlane(soff), 0, 0);
}
// A quarter pair hi (0: rows 0..63 of both 128-row halves, 1: rows 64..127) of K-tile kt into buffer buf
__device__ __forceinline__ void stageA(const Lane& L, const Ctx& c, int kt, int buf, int hi, int wave) {
    const int row = 64 * hi + 8 * wave;
    gm_dma(L.A, c.nA, L.voffA, c.sA + 64 * hi * L.lda2 + kt * 128, L.lds + buf * 65536 + row * 128);
    gm_dma(L.A, c.nA, L.voffA, c.sA + (128 + 64 * hi) * L.lda2 + kt * 128, L.lds + buf * 65536 + (128 + row) * 128);
}
// W half (0: LDS rows 0..127 = the 64-row groups of waves wc 0, 1; 1: rows 128..255) of K-tile kt into buffer buf
__device__ __forceinline__ void stageW(const Lane& L, const Ctx& c, int kt, int buf, int half, int wave) {
    const int row = 128 * half + 8 * wave;
    gm_dma(L.W, c.nW, L.voffW, c.sW + 2 * half * L.wgrp + kt * 128, L.lds + buf * 65536 + 32768 + row * 128);
    gm_dma(L.W, c.nW, L.voffW, c.sW + (2 * half + 1) * L.wgrp + kt * 128, L.lds + buf * 65536 + 32768 + (row + 64) * 128);
}
struct Frags {
    bf16x8 a[4][2];                      // A fragments of the current 64-row group (a operand: row = m)
    bf16x8 wlo[2][2], whi[2][2];         // W fragments, n-tiles 0, 1 / 2, 3 (b operand: column j)
};
__device__ __forceinline__ bf16x8 lds8(const char* p) { return *reinterpret_cast<const bf16x8*>(p); }
template <int Q>
__device__ __forceinline__ void load_frags(const Lane& L, int buf, Frags& f) {
    const char* base = L.lds + buf * 65536;
    if (Q == 0) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) f.wlo[ni][ks] = lds8(base + L.rdW[ks] + ni * 2048);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (Q == 1) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) f.whi[ni][ks] = lds8(base + L.rdW[ks] + (2 + ni) * 2048);
    }
    if (Q == 0 || Q == 2) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) f.a[mi][ks] = lds8(base + L.rdA[ks] + ((Q == 2 ? 4 : 0) + mi) * 2048);
    }
}
// quadrant Q = (m group, n pair) (0,0) (0,1) (1,1) (1,0); acc[mi][ni]: rows 16 mi + 4 g + reg of the wave's tile, MFMA column j
template <int Q, bool FIRST>
__device__ __forceinline__ void mfma_phase(const Frags& f, f32x4 (&acc)[8][4]) {
    constexpr int qa = Q >= 2, qb = (Q == 1 || Q == 2);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const f32x4 cin = (FIRST && ks == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[4 * qa + mi][2 * qb + ni];
                acc[4 * qa + mi][2 * qb + ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.a[mi][ks], qb ? f.whi[ni][ks] : f.wlo[ni][ks], cin, 0, 0, 0);
            }
}
// Rows 64 QA + 32 PART .. + 32 of the wave's tile.
//   EPI 0: MFMA column j of n-tile ni is C column 4 j + ni of the wave's 64: 8 stores of 8 bytes per lane, each = 4 whole 128-byte lines.
//   EPI 1: n-tiles 0, 1 are val columns 2 j, 2 j + 1 of the wave's 32, n-tiles 2, 3 the gate columns of the same pair: per row one 4-byte
//          store each for val, gate (into h = C) and gelu(gate) * val (into G); the 16 lanes of a group write 64 contiguous bytes.
template <int EPI, int QA, int PART>
__device__ __forceinline__ void store_half(const f32x4 (&acc)[8][4], const Lane& L, const Out& o) {
    const __amdgpu_buffer_rsrc_t rsC = gm_rsrc(L.C, o.nC), rsG = gm_rsrc(EPI == 0 ? L.C : L.G, o.nG);
    const int corigin = o.c, gorigin = o.g;
#pragma unroll
    for (int m2 = 0; m2 < 2; ++m2) {
        const int mi = 4 * QA + 2 * PART + m2;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            if (EPI == 0) {
                u32x2 v;
                v[0] = gm_pack2(acc[mi][0][reg], acc[mi][1][reg]); v[1] = gm_pack2(acc[mi][2][reg], acc[mi][3][reg]);
                __builtin_amdgcn_raw_buffer_store_b64(v, rsC, L.voffC, __builtin_amdgcn_readfirstlane(corigin + (16 * mi + reg) * L.ldc2), GM_STORE_AUX);
            } else {
                // h keeps the values the backward reads (GEGLU' needs val and gate), rounded to bf16 exactly as a separate GEMM would
                // round them; the product is formed from those rounded values (= what geglu_fwd_kernel computes from h)
                const unsigned hv = gm_pack2(acc[mi][0][reg], acc[mi][1][reg]), hg = gm_pack2(acc[mi][2][reg], acc[mi][3][reg]);
                const bf16x2 bv = __builtin_bit_cast(bf16x2, hv), bg = __builtin_bit_cast(bf16x2, hg);
#if GM_EPI_DIAG == 1          // timing only: no GELU
                const unsigned pr = gm_pack2((float)bg[0] * (float)bv[0], (float)bg[1] * (float)bv[1]);
#elif GM_GELU_LUT
                const unsigned pr = gm_pack2((float)bg[0] * gm_phi_lut<0>(L.lds, hg) * (float)bv[0], (float)bg[1] * gm_phi_lut<16>(L.lds, hg) * (float)bv[1]);
#else
                const unsigned pr = gm_pack2(gm_gelu((float)bg[0]) * (float)bv[0], gm_gelu((float)bg[1]) * (float)bv[1]);
#endif
                const int so = __builtin_amdgcn_readfirstlane(corigin + (16 * mi + reg) * L.ldc2);
#if GM_EPI_DIAG == 3          // timing only: h as ONE 8-byte store per lane (val / gate pairs interleaved inside the tile's 256 columns)
                __builtin_amdgcn_raw_buffer_store_b64(u32x2{hv, hg}, rsC, L.voffC, so, GM_STORE_AUX);
#else
                __builtin_amdgcn_raw_buffer_store_b32(hv, rsC, L.voffC, so, GM_STORE_AUX);
                __builtin_amdgcn_raw_buffer_store_b32(hg, rsC, L.voffC, so + L.gate_off, GM_STORE_AUX);
#endif
#if GM_EPI_DIAG == 2          // timing only: the product is computed and kept live, its store dropped by a record count of 0
                __builtin_amdgcn_raw_buffer_store_b32(pr, gm_rsrc(L.G, 0), L.voffG, __builtin_amdgcn_readfirstlane(gorigin + (16 * mi + reg) * L.ldg2), GM_STORE_AUX);
#else
                __builtin_amdgcn_raw_buffer_store_b32(pr, rsG, L.voffG, __builtin_amdgcn_readfirstlane(gorigin + (16 * mi + reg) * L.ldg2), GM_STORE_AUX);
#endif
            }
        }
    }
}
#define GM_PHASE_TAIL(Q, FIRST)                                            \
    __builtin_amdgcn_sched_barrier(0);                                     \
    __builtin_amdgcn_s_barrier();                                          \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                     \
    __builtin_amdgcn_sched_barrier(0);                                     \
    __builtin_amdgcn_s_setprio(1);                                         \
    mfma_phase<Q, FIRST>(f, acc);                                          \
    __builtin_amdgcn_s_setprio(0);                                         \
    __builtin_amdgcn_sched_barrier(0);                                     \
    __builtin_amdgcn_s_barrier();                                          \
    __builtin_amdgcn_sched_barrier(0);

constexpr int vmc(int n) { return n > 63 ? 63 : n; }       // the vmcnt field has 6 bits; a smaller count only waits for more

// One K-tile.  KIND 0: steady; 1: FIRST K-tile of an output tile (chains start from zero; rows 64..127 of the PREVIOUS tile are stored in
// phases 0 / 1); 2: the K-tile after it; 3: LAST K-tile (rows 0..63 of this tile stored in phases 2 / 3).  (c1, kt1) / (c2, kt2): where the
// K-tiles "t+1" / "t+2" live (they may belong to the next output tile).  SPP = stores per storing phase (after the phase's DMA and wait).
template <int EPI, int KIND>
__device__ __forceinline__ void ktile(const Lane& L, const Ctx& c1, int kt1, const Ctx& c2, int kt2, int b, int wave, Frags& f, f32x4 (&acc)[8][4],
                                      const Out& prev, const Out& cur) {
    constexpr bool FIRST = KIND == 1, LAST = KIND == 3;
    constexpr int SPP = EPI == 0 ? 8 : (GM_EPI_DIAG == 3 ? 16 : 24);
    load_frags<0>(L, b, f);
    stageW(L, c1, kt1, b ^ 1, 1, wave);
    if (FIRST) store_half<EPI, 1, 0>(acc, L, prev);
    GM_PHASE_TAIL(0, FIRST)
    load_frags<1>(L, b, f);
    stageA(L, c1, kt1, b ^ 1, 1, wave);
    // A-hi of THIS K-tile (issued 4 staging steps ago) has landed
    if (FIRST) GM_VMCNT(vmc(8 + 3 * SPP)); else if (KIND == 2) GM_VMCNT(vmc(8 + SPP)); else GM_VMCNT(8);
    if (FIRST) store_half<EPI, 1, 1>(acc, L, prev);
    GM_PHASE_TAIL(1, FIRST)
    load_frags<2>(L, b, f);
    stageA(L, c2, kt2, b, 0, wave);
    if (LAST) store_half<EPI, 0, 0>(acc, L, cur);
    GM_PHASE_TAIL(2, FIRST)
    stageW(L, c2, kt2, b, 0, wave);
    // A-lo, W half 0, W half 1 of the next K-tile have landed
    if (LAST) GM_VMCNT(vmc(6 + SPP)); else if (FIRST) GM_VMCNT(vmc(6 + 2 * SPP)); else GM_VMCNT(6);
    if (LAST) store_half<EPI, 0, 1>(acc, L, cur);
    GM_PHASE_TAIL(3, FIRST)
}

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm8p_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int x = blockIdx.x & 7, jw = blockIdx.x >> 3, nper = gridDim.x >> 3;            // grid = 8 * nper workgroups
    const int txcd = tiles_of_xcd(x, p.ntm, p.ntn);
    const int ntile = (txcd - jw + nper - 1) / nper;                                       // local indices jw, jw + nper, ...
    if (ntile <= 0) return;
    const int wrows = EPI == 0 ? p.N : 2 * p.N;
    const unsigned szA = (unsigned)(((long)(p.M - 1) * p.lda + p.K) * 2), szW = (unsigned)(((long)(wrows - 1) * p.ldw + p.K) * 2);
    const unsigned szC = (unsigned)(((long)(p.M - 1) * p.ldc + wrows) * 2), szG = EPI == 0 ? 0u : (unsigned)(((long)(p.M - 1) * p.ldg + p.N) * 2);
    Lane L;
    L.lds = lds;
    L.lda2 = p.lda * 2; L.ldw2 = p.ldw * 2; L.ldc2 = p.ldc * 2; L.ldg2 = p.ldg * 2;
    L.wgrp = (EPI == 0 ? 64 : 32) * L.ldw2;
    L.gate_off = p.N * 2;
    L.A = p.A; L.W = p.W; L.C = p.C; L.G = p.G;
    {
        const int pr = lane >> 3, ch = lane & 7, j = lane & 15, g = lane >> 4;
        L.voffA = pr * L.lda2 + 16 * (ch ^ pr);
        L.voffW = (EPI == 0 ? 4 : 2) * pr * L.ldw2 + 16 * (ch ^ pr);       // the 8 LDS rows of a piece are MFMA columns j0 .. j0 + 7 of one n-tile
        const int ra = (128 * wr + j) * 128 + 16 * (g ^ (j & 7));
        const int rw = 32768 + (64 * wc + j) * 128 + 16 * (g ^ (j & 7));
        L.rdA[0] = ra; L.rdA[1] = ra ^ 64; L.rdW[0] = rw; L.rdW[1] = rw ^ 64;
        L.voffC = 4 * g * L.ldc2 + (EPI == 0 || GM_EPI_DIAG == 3 ? 8 : 4) * j;
        L.voffG = 4 * g * L.ldg2 + 4 * j;
    }
    // this wave's pieces cover LDS rows 8 wave .. + 7 of a 64-row group = MFMA columns 8 (wave & 1) .. + 7 of n-tile wave >> 1
    //   EPI 0: source row 4 j + ni  -> 32 (wave & 1) + (wave >> 1) + 4 (lane >> 3)
    //   EPI 1: n-tiles 0, 1: val row 2 j + ni; n-tiles 2, 3: gate row F + 2 j + (ni - 2)  -> 16 (wave & 1) + (wave >> 1 & 1) [+ F] + 2 (lane >> 3)
    const int wrow0 = EPI == 0 ? 32 * (wave & 1) + (wave >> 1) : 16 * (wave & 1) + ((wave >> 1) & 1) + (wave >= 4 ? p.N : 0);
    auto origin = [&](Ctx& c, TileXY t, bool real) {
        c.nA = real ? szA : 0u; c.nW = real ? szW : 0u;
        c.sA = (t.tm * 256 + 8 * wave) * L.lda2;
        c.sW = (t.tn * (EPI == 0 ? 256 : 128) + wrow0) * L.ldw2;
    };
    auto out_of = [&](TileXY t) {
        Out o;
        o.nC = szC; o.nG = szG;
        o.c = (t.tm * 256 + 128 * wr) * L.ldc2 + (t.tn * (EPI == 0 || GM_EPI_DIAG == 3 ? 256 : 128) + (EPI == 0 || GM_EPI_DIAG == 3 ? 64 : 32) * wc) * 2;
        o.g = EPI == 0 ? 0 : (t.tm * 256 + 128 * wr) * L.ldg2 + (t.tn * 128 + 32 * wc) * 2;
        return o;
    };
    f32x4 acc[8][4];
    Frags f;
    const int NT = p.K >> 6;                                 // even, >= 6 (host-checked)
    TileXY cur = tile_of(x, jw, p.ntm, p.ntn);
    Ctx c, cn;
    origin(c, cur, true);
#if GM_GELU_LUT
    if (EPI == 1) {                                         // the Phi table: 16 pieces of 1 KiB, two per wave, OLDER than the tile pieces the vmcnt(6) below leaves in flight
        gm_dma(gm_phi_table, 16384u, 16 * lane, 2048 * wave, lds + GM_PHI_LDS + 2048 * wave);
        gm_dma(gm_phi_table, 16384u, 16 * lane, 2048 * wave + 1024, lds + GM_PHI_LDS + 2048 * wave + 1024);
    }
#endif
    stageA(L, c, 0, 0, 0, wave); stageW(L, c, 0, 0, 0, wave); stageW(L, c, 0, 0, 1, wave); stageA(L, c, 0, 0, 1, wave);
    stageA(L, c, 1, 1, 0, wave); stageW(L, c, 1, 1, 0, wave);
    GM_VMCNT(6);
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();              // waves 4..7 run one barrier behind (MFMAs of one group over the other's loads)
    Out prev{0u, 0u, 0, 0};                                  // the first tile has no predecessor: its "previous rows" go nowhere (zero records)
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < ntile; ++it) {
        const bool has_next = it + 1 < ntile;
        const TileXY nxt = tile_of(x, jw + (has_next ? it + 1 : it) * nper, p.ntm, p.ntn);
        origin(cn, nxt, has_next);
        const Out ocur = out_of(cur);
        ktile<EPI, 1>(L, c, 1, c, 2, 0, wave, f, acc, prev, ocur);
        ktile<EPI, 2>(L, c, 2, c, 3, 1, wave, f, acc, prev, ocur);
        for (int t = 2; t < NT - 2; t += 2) {
            ktile<EPI, 0>(L, c, t + 1, c, t + 2, 0, wave, f, acc, prev, ocur);
            ktile<EPI, 0>(L, c, t + 2, c, t + 3, 1, wave, f, acc, prev, ocur);
        }
        ktile<EPI, 0>(L, c, NT - 1, cn, 0, 0, wave, f, acc, prev, ocur);          // t = NT - 2
        ktile<EPI, 3>(L, cn, 0, cn, 1, 1, wave, f, acc, prev, ocur);              // t = NT - 1
        prev = ocur;
        cur = nxt; c = cn;
    }
    store_half<EPI, 1, 0>(acc, L, prev);
    store_half<EPI, 1, 1>(acc, L, prev);
    if (wr == 0) __builtin_amdgcn_s_barrier();
}

bool gemm_shape_ok(long M, long N, long K, long lda, long ldw, long ldc, long wrows, int ncol_tile) {
    if (M <= 0 || N <= 0 || K < 384 || (K % 128) || (N % ncol_tile)) return false;
    if (lda < K || ldw < K || ldc < wrows || (lda % 8) || (ldw % 8) || (ldc % 2)) return false;
    // 31-bit byte offsets inside the buffer descriptors, incl. the rows a partial last M-tile addresses beyond M (range-checked away)
    const long mpad = (M + 255) / 256 * 256;
    return mpad * lda * 2 < (1L << 31) && wrows * ldw * 2 < (1L << 31) && mpad * ldc * 2 < (1L << 31);      // (int offset arithmetic in the kernel)
}

// The opt-in to more than 64 KB of dynamic LDS is a PER-DEVICE function attribute: set once per (kernel, device) -- a process that drives
// several GPUs launches on each of them -- behind an atomic flag (racing threads set the same value twice at worst).
template <typename K>
bool gm_lds_opt_in(K kernel, int bytes, std::atomic<bool> (&done)[64]) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    if (!done[dev].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
        done[dev].store(true, std::memory_order_release);
    }
    return true;
}

template <int EPI>
int launch_gemm(const GemmArgs& a, hipStream_t st) {
    static std::atomic<bool> attr_set[64];
    constexpr int ldsb = 131072 + ((EPI == 1 && GM_GELU_LUT) ? 16384 : 0);
    if (!gm_lds_opt_in(gemm8p_kernel<EPI>, ldsb, attr_set)) return MMAE_ERR_LAUNCH;
    const long tiles = (long)a.ntm * a.ntn;
    const int nper = tiles >= 256 ? 32 : (int)((tiles + 7) / 8);          // workgroups per XCD group; one workgroup per CU (128 KB of LDS)
    MMAE_LAUNCH((gemm8p_kernel<EPI>), dim3(8 * nper), dim3(512), ldsb, st, a);
    MMAE_CHECK_LAUNCH();
    return MMAE_OK;
}

}  // namespace

extern "C" int mmae_gemm_nt_supported(long M, long N, long K, long lda, long ldw, long ldc) {
    return gemm_shape_ok(M, N, K, lda, ldw, ldc, N, 256) ? 1 : 0;
}

extern "C" int mmae_gemm_nt(long M, long N, long K, const void* A, long lda, const void* W, long ldw, void* C, long ldc, void* stream) {
    if (!A || !W || !C || !gemm_shape_ok(M, N, K, lda, ldw, ldc, N, 256)) return MMAE_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(W) & 15) || (reinterpret_cast<uintptr_t>(C) & 7)) return MMAE_ERR_ARG;
    GemmArgs a{};
    a.A = (const bf16*)A; a.W = (const bf16*)W; a.C = (bf16*)C; a.G = nullptr;
    a.M = (int)M; a.N = (int)N; a.K = (int)K; a.lda = (int)lda; a.ldw = (int)ldw; a.ldc = (int)ldc; a.ldg = 0;
    a.ntm = (int)((M + 255) / 256); a.ntn = (int)(N / 256);
    return launch_gemm<0>(a, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int mmae_gemm_geglu_supported(long M, long F, long K, long lda, long ldw, long ldh, long ldg) {
    return (gemm_shape_ok(M, F, K, lda, ldw, ldh, 2 * F, 128) && ldg >= F && (ldg % 2) == 0 && (M + 255) / 256 * 256 * ldg * 2 < (1L << 31)) ? 1 : 0;
}

extern "C" int mmae_gemm_geglu(long M, long F, long K, const void* A, long lda, const void* W1, long ldw, void* h, long ldh, void* g, long ldg,
                               void* stream) {
    if (!A || !W1 || !h || !g || !mmae_gemm_geglu_supported(M, F, K, lda, ldw, ldh, ldg)) return MMAE_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(W1) & 15) || (reinterpret_cast<uintptr_t>(h) & 3) ||
        (reinterpret_cast<uintptr_t>(g) & 3)) return MMAE_ERR_ARG;
    GemmArgs a{};
    a.A = (const bf16*)A; a.W = (const bf16*)W1; a.C = (bf16*)h; a.G = (bf16*)g;
    a.M = (int)M; a.N = (int)F; a.K = (int)K; a.lda = (int)lda; a.ldw = (int)ldw; a.ldc = (int)ldh; a.ldg = (int)ldg;
    a.ntm = (int)((M + 255) / 256); a.ntn = (int)(F / 128);
    return launch_gemm<1>(a, reinterpret_cast<hipStream_t>(stream));
}

// =====================================================================================================================================
// Weight gradients: dW[N, Kin] = G[rows, N]^T . X[rows, Kin] (bf16 in, fp32 out) -- autograd's dW = dY^T X of the same nn.Linear layers.
// Both operands are contracted over their ROW index, so a K-tile is 64 rows of G and of X, staged row-major exactly as they lie in HBM
// ([64 r][256 columns], 512-byte rows, 1 KiB LDS-DMA pieces = 2 rows) and every MFMA fragment is fetched with the hardware transpose read
// ds_read_b64_tr_b16 (two per fragment: 4 rows x 16 columns each, column-major to the lanes).  Swizzle: the 32-byte column-tile index of a
// row is XORed with m(row) = (row & 3) | ((row >> 3 & 1) << 2) on the DMA source and on the read address -- the 32 lanes of a half-wave
// then touch 8 rows x 32 bytes on 64 distinct banks (conflict-free), and m does not change between a fragment's two reads (rows + 4) nor
// between the k-steps (rows + 32): one v_xor per fragment address, everything else is an immediate offset.
// Same pipeline as above (8 waves, 2 LDS buffers, staggered wave groups, counted vmcnt), but the phases of a K-tile are (k-step, M half):
// rows 0..31 of both tiles are read in phases 0 / 1 and restaged in phase 3 / phase 0 of the next K-tile, rows 32..63 in phases 2 / 3 and
// restaged in phases 1 / 2 -- so the counted waits sit in phases 1 and 3 with three staging steps in flight each.
// The output has few tiles (6..48) while the contraction is 65k..165k rows long: SPLIT-K over workgroups -- (tile, split) pairs fill the
// 256 CUs, each writes its fp32 partial tile into slab `split`, and splitk_sum_f32_kernel adds the slabs in a fixed order (no atomics:
// bitwise reproducible).  No rolling epilogue: one tile per workgroup, 500+ K-tiles each.
namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

struct TnArgs {
    const bf16* G; const bf16* X; float* P;      // P: S slabs of (N, Kin) fp32 (S == 1: the destination itself)
    int rows, N, Kin, ldg, ldx;
    int tkin, T, S, rows_per_split;              // tiles along Kin, tiles in all, splits, rows per split (multiple of 128)
};

__device__ __forceinline__ bf16x8 tn_frag(const char* p) {          // rows R..R+3 and R+4..R+7 of one 16-column tile -> 8 k-slots
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(p + 2048));
    s16x8 x;
    x[0] = lo[0]; x[1] = lo[1]; x[2] = lo[2]; x[3] = lo[3]; x[4] = hi[0]; x[5] = hi[1]; x[6] = hi[2]; x[7] = hi[3];
    return __builtin_bit_cast(bf16x8, x);
}

struct TnLane {
    char* lds;
    const bf16* G; const bf16* X;
    unsigned nG, nX;                 // record counts: bytes up to the end of this split's last row (later rows read as zero)
    int voffG[2], voffX[2];          // per-lane DMA source offsets of the two pieces of a staging step (rows 4 w + half, + 2: m differs in bit 1)
    int sG, sX;                      // scalar byte offsets: this split's first row + this wave's rows inside a 32-row block, tile column origin
    int ldg2, ldx2;
    int lpA, lpB;                    // fragment read bases: (8 g + q) * 512 + 8 p, XORed with 32 m and the wave's column-tile origin
};

// LDS = a ring of TEN 16 KB slots (all 160 KB of the CU); a staging STEP = 32 rows x 256 columns of one operand (2 DMA per lane), four
// steps per K-tile in the order G rows 0..31, X rows 0..31, G rows 32..63, X rows 32..63; global step s lives in slot s mod 10.
// Phase (t, q) issues step 4 t + q + 7: a slot is refilled >= 2 phases after its last read, every read is >= 1 phase after the counted wait
// that retires its DMA, and FIVE steps (80 KB per CU) stay in flight across the barriers -- the loop is bound by the latency of its
// operand stream (every tile row is wanted by 3..16 workgroups at the same moment, so all of them see the HBM latency), not by MFMA, LDS or
// HBM bandwidth: with the two-buffer form of the projection kernel above (3 steps in flight) it ran at 1040 TFLOP/s, with every DMA hitting
// L2 at 1546.  The slot pattern repeats every 5 K-tiles, so the loop is unrolled by 5 with compile-time LDS addresses.
//   step kind j = s mod 4: read in phases  j 0: (t,0) (t,1)   j 1: (t,0)   j 2: (t,2) (t,3)   j 3: (t,2)
__device__ __forceinline__ void tn_stage(const TnLane& L, int which, int kt, int blk, int slot, int wave) {
    char* dst = L.lds + slot * 16384 + 4 * wave * 512;                  // wave w brings rows 4 w .. 4 w + 3 of the 32-row block
    const int ld2 = which ? L.ldx2 : L.ldg2;
    const int so = (which ? L.sX : L.sG) + (kt * 64 + 32 * blk) * ld2;
    const void* base = which ? (const void*)L.X : (const void*)L.G;
    const unsigned n = which ? L.nX : L.nG;
    gm_dma(base, n, which ? L.voffX[0] : L.voffG[0], so, dst);
    gm_dma(base, n, which ? L.voffX[1] : L.voffG[1], so + 2 * ld2, dst + 1024);
}
// global step s (any s >= 0) of the K-loop whose first K-tile index is given implicitly: kt = s / 4, kind = s % 4
template <int KIND>
__device__ __forceinline__ void tn_stage_step(const TnLane& L, int kt, int slot, int wave) {
    tn_stage(L, KIND & 1, kt, KIND >> 1, slot, wave);
}

struct TnFrags { bf16x8 a[4], b[4]; };

// phase (KS, QA) of the K-tile whose steps sit in slots SA (G rows of k-step KS) / SB (X rows of k-step KS)
template <int QA, int SA, int SB>
__device__ __forceinline__ void tn_load(const TnLane& L, TnFrags& f) {
    if (QA == 0) {
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) f.b[tj] = tn_frag(L.lds + SB * 16384 + (L.lpB ^ (tj * 32)));
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) f.a[ti] = tn_frag(L.lds + SA * 16384 + (L.lpA ^ ((4 * QA + ti) * 32)));
}
template <int QA>
__device__ __forceinline__ void tn_mfma(const TnFrags& f, f32x4 (&acc)[8][4]) {
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int tj = 0; tj < 4; ++tj)
            acc[4 * QA + ti][tj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.a[ti], f.b[tj], acc[4 * QA + ti][tj], 0, 0, 0);
}
#define TN_PHASE_TAIL(QA)                                                  \
    __builtin_amdgcn_sched_barrier(0);                                     \
    __builtin_amdgcn_s_barrier();                                          \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                     \
    __builtin_amdgcn_sched_barrier(0);                                     \
    __builtin_amdgcn_s_setprio(1);                                         \
    tn_mfma<QA>(f, acc);                                                   \
    __builtin_amdgcn_s_setprio(0);                                         \
    __builtin_amdgcn_sched_barrier(0);                                     \
    __builtin_amdgcn_s_barrier();                                          \
    __builtin_amdgcn_sched_barrier(0);

// K-tile t with t mod 5 == R: its steps 4 t .. 4 t + 3 sit in slots (4 R + j) mod 10; phase q issues step 4 t + q + 7 into slot (4 R + q + 7) mod 10
// (kinds 3, 0, 1, 2 of K-tiles t+1, t+2, t+2, t+2).  Waits in phases 1 and 3: vmcnt(10) = five steps stay in flight.
template <int R>
__device__ __forceinline__ void tn_ktile(const TnLane& L, int t, int wave, TnFrags& f, f32x4 (&acc)[8][4]) {
    constexpr int S0 = (4 * R) % 10, S1 = (4 * R + 1) % 10, S2 = (4 * R + 2) % 10, S3 = (4 * R + 3) % 10;
    tn_load<0, S0, S1>(L, f);
    tn_stage_step<3>(L, t + 1, (4 * R + 7) % 10, wave);
    TN_PHASE_TAIL(0)
    tn_load<1, S0, S1>(L, f);
    tn_stage_step<0>(L, t + 2, (4 * R + 8) % 10, wave);
    GM_VMCNT(10);
    TN_PHASE_TAIL(1)
    tn_load<0, S2, S3>(L, f);
    tn_stage_step<1>(L, t + 2, (4 * R + 9) % 10, wave);
    TN_PHASE_TAIL(0)
    tn_load<1, S2, S3>(L, f);
    tn_stage_step<2>(L, t + 2, (4 * R + 10) % 10, wave);
    GM_VMCNT(10);
    TN_PHASE_TAIL(1)
}

__global__ __launch_bounds__(512, 2) void gemm_tn8p_kernel(TnArgs p) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    // bijective XCD-aware order: the workgroups of one XCD group take consecutive (split, tile) pairs -- mostly ONE split, i.e. the same
    // rows of G and X through that XCD's L2
    const int nwg = gridDim.x, x = blockIdx.x & 7, idx = blockIdx.x >> 3, q8 = nwg >> 3, r8 = nwg & 7;
    const int lin = (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + idx;
    const int split = lin / p.T, tile = lin - split * p.T;
    const int tn = tile / p.tkin, tk = tile - tn * p.tkin;
    const int r_begin = split * p.rows_per_split, r_end = min(p.rows, r_begin + p.rows_per_split);
    const int NT5 = (r_end - r_begin + 319) / 320;                       // groups of 5 K-tiles of 64 rows (rows beyond r_end stage zeros)
    TnLane L;
    L.lds = lds; L.G = p.G; L.X = p.X;
    L.ldg2 = p.ldg * 2; L.ldx2 = p.ldx * 2;
    L.nG = (unsigned)(((long)(r_end - 1) * p.ldg + p.N) * 2);
    L.nX = (unsigned)(((long)(r_end - 1) * p.ldx + p.Kin) * 2);
    {
        const int half = lane >> 5, ch = lane & 31;
        const int m0 = half | (((wave >> 1) & 1) << 2);                   // m(row) of piece 0's rows 4 w + half (piece 1: rows + 2 -> m ^ 2)
        L.voffG[0] = half * L.ldg2 + 16 * (ch ^ (m0 << 1)); L.voffG[1] = half * L.ldg2 + 16 * (ch ^ ((m0 ^ 2) << 1));
        L.voffX[0] = half * L.ldx2 + 16 * (ch ^ (m0 << 1)); L.voffX[1] = half * L.ldx2 + 16 * (ch ^ ((m0 ^ 2) << 1));
        L.sG = (r_begin + 4 * wave) * L.ldg2 + tn * 512;
        L.sX = (r_begin + 4 * wave) * L.ldx2 + tk * 512;
        const int g = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
        const int m = qq | ((g & 1) << 2);
        const int lp = (8 * g + qq) * 512 + 8 * pp;
        L.lpA = lp ^ (32 * m) ^ (256 * wr);                               // n-tiles 8 wr .. of the G block
        L.lpB = lp ^ (32 * m) ^ (128 * wc);                               // kin-tiles 4 wc .. of the X block
    }
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    TnFrags f;
    // prologue: steps 0 .. 6 (K-tile 0 complete, K-tile 1 up to its G rows 32..63) into slots 0 .. 6; steps 0, 1 landed before phase 0
    tn_stage_step<0>(L, 0, 0, wave); tn_stage_step<1>(L, 0, 1, wave); tn_stage_step<2>(L, 0, 2, wave); tn_stage_step<3>(L, 0, 3, wave);
    tn_stage_step<0>(L, 1, 4, wave); tn_stage_step<1>(L, 1, 5, wave); tn_stage_step<2>(L, 1, 6, wave);
    GM_VMCNT(10);
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();                           // waves 4..7 run one barrier behind
    for (int t5 = 0; t5 < NT5; ++t5) {
        const int t = 5 * t5;
        tn_ktile<0>(L, t, wave, f, acc);
        tn_ktile<1>(L, t + 1, wave, f, acc);
        tn_ktile<2>(L, t + 2, wave, f, acc);
        tn_ktile<3>(L, t + 3, wave, f, acc);
        tn_ktile<4>(L, t + 4, wave, f, acc);
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();
    GM_VMCNT(0);                                                          // the zero-fill staging of the K-tiles beyond the last one
    // acc[it][jt][reg] = dW[tn 256 + 128 wr + 16 it + 4 g + reg][tk 256 + 64 wc + 16 jt + (lane & 15)]: once per workgroup
    float* out = p.P + (long)split * p.N * p.Kin + ((long)tn * 256 + 128 * wr + 4 * (lane >> 4)) * p.Kin + tk * 256 + 64 * wc + (lane & 15);
#pragma unroll
    for (int it = 0; it < 8; ++it)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) out[(long)(16 * it + reg) * p.Kin + 16 * jt] = acc[it][jt][reg];
}

__global__ __launch_bounds__(256) void splitk_sum_f32_kernel(const float* __restrict__ part, int S, long n, float* __restrict__ out) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    f32x4 a = *reinterpret_cast<const f32x4*>(part + i);
    for (int s = 1; s < S; ++s) a += *reinterpret_cast<const f32x4*>(part + (long)s * n + i);      // fixed order: reproducible
    *reinterpret_cast<f32x4*>(out + i) = a;
}

struct TnPlan { int tkin, T, S, rps; };
bool tn_plan(long rows, long N, long Kin, long ldg, long ldx, TnPlan* pl) {
    if (rows < 128 || N <= 0 || Kin <= 0 || (N % 256) || (Kin % 256) || ldg < N || ldx < Kin || (ldg % 8) || (ldx % 8)) return false;
    if ((rows + 256) * ldg * 2 >= (1L << 31) || (rows + 256) * ldx * 2 >= (1L << 31) || N * Kin >= (1L << 31)) return false;   // int offsets
    const long T = (N / 256) * (Kin / 256);
    long S = T >= 256 ? 1 : 256 / T;
    const long max_s = rows / 2048 > 0 ? rows / 2048 : 1;                 // at least 32 K-tiles per split
    if (S > max_s) S = max_s;
    long rps = (rows + S - 1) / S;
    rps = (rps + 127) / 128 * 128;
    S = (rows + rps - 1) / rps;                                           // no empty split
    pl->tkin = (int)(Kin / 256); pl->T = (int)T; pl->S = (int)S; pl->rps = (int)rps;
    return true;
}

}  // namespace

extern "C" int mmae_gemm_tn_supported(long rows, long N, long Kin, long ldg, long ldx) {
    TnPlan pl;
    return tn_plan(rows, N, Kin, ldg, ldx, &pl) ? 1 : 0;
}
// fp32 workspace for the split-K slabs (0 when the shape runs unsplit or is unsupported)
extern "C" long mmae_gemm_tn_ws_floats(long rows, long N, long Kin) {
    TnPlan pl;
    if (!tn_plan(rows, N, Kin, N, Kin, &pl)) return 0;
    return pl.S > 1 ? (long)pl.S * N * Kin : 0;
}
extern "C" int mmae_gemm_tn(long rows, long N, long Kin, const void* G, long ldg, const void* X, long ldx, float* out, float* ws, void* stream) {
    TnPlan pl;
    if (!G || !X || !out || !tn_plan(rows, N, Kin, ldg, ldx, &pl)) return MMAE_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(G) & 15) || (reinterpret_cast<uintptr_t>(X) & 15) || (reinterpret_cast<uintptr_t>(out) & 15)) return MMAE_ERR_ARG;
    if (pl.S > 1 && (!ws || (reinterpret_cast<uintptr_t>(ws) & 15))) return MMAE_ERR_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    static std::atomic<bool> attr_set[64];
    if (!gm_lds_opt_in(gemm_tn8p_kernel, 163840, attr_set)) return MMAE_ERR_LAUNCH;
    TnArgs a{};
    a.G = (const bf16*)G; a.X = (const bf16*)X; a.P = pl.S > 1 ? ws : out;
    a.rows = (int)rows; a.N = (int)N; a.Kin = (int)Kin; a.ldg = (int)ldg; a.ldx = (int)ldx;
    a.tkin = pl.tkin; a.T = pl.T; a.S = pl.S; a.rows_per_split = pl.rps;
    MMAE_LAUNCH(gemm_tn8p_kernel, dim3(pl.T * pl.S), dim3(512), 163840, st, a);
    MMAE_CHECK_LAUNCH();
    if (pl.S > 1) {
        const long n = N * Kin;
        MMAE_LAUNCH(splitk_sum_f32_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, (const float*)ws, pl.S, n, out);
        MMAE_CHECK_LAUNCH();
    }
    return MMAE_OK;
}
