// Shared declarations of the masked-attention kernels (mha.hip: generic fp32/bf16; mha_bf16.hip: bf16 fast path).
#pragma once
#include "common.hpp"

#define MAXSEG 8

struct MhaDesc {
    const void* q; const void* k; const void* v;
    void* o;                 // fwd: out; bwd: forward output O (read)
    const void* dout;        // bwd
    void* dq; void* dk; void* dv;
    float* lse;              // (H, q_rows_total)
    float* delta;            // (H, q_rows_total)
    long q_stride, k_stride, v_stride, o_stride, do_stride, dq_stride, dk_stride, dv_stride;
    const int* q_start; const int* q_len; const int* k_start; const int* k_len;   // (B, nseg) each
    long stat_stride;        // q_rows_total
    int B, H, nseg, max_tiles;
    int max_q_rows, max_k_rows;   // upper bounds of a sample's query / key rows (all segments): kernel selection only
    float scale;
    int empty_mode;
    int hpb_req;             // sample-head kernels: heads one workgroup walks; 0 = chosen from (B, H) (tests force 1 / 2 / H through
                             // bits 8..11 of the `variant` argument of mmae_internal.h)
    float* dq_ws;            // fused backward (mha_sh_bwd_kernel): fp32 partial dQ tiles, B x max_qt x H x 4096 floats (else null)
    int max_qt;              // its tile slots per sample: upper bound of a sample's 64-row query tiles
};

struct TileSel { int seg, t0, n; };

// XCD-aware block -> (sample, head, tile) map.  Workgroups are dealt round-robin over the 8 XCDs (MI355X_MICROARCH.md,
// dispatch section), so L % 8 names the XCD group of linear block L.  All tiles of one (sample, head) pair are put on the
// same group: they stream the same K/V (or Q/dO) rows, which then stay in that XCD's 4 MiB L2 instead of being fetched
// from HBM once per XCD.  Speed only -- correctness does not depend on the placement.
struct BlockSel { int b, h, t; };
__device__ __forceinline__ BlockSel decode_block(int L, int max_tiles, int B, int H) {
    const int xcd = L & 7, slot = L >> 3;
    const int g = (slot / max_tiles) * 8 + xcd;         // (sample, head) pair index
    BlockSel r;
    r.t = slot % max_tiles;
    r.b = g / H;
    r.h = g % H;
    if (g >= B * H) r.b = -1;
    return r;
}
static inline int xcd_grid(int B, int H, int max_tiles) { return ((B * H + 7) / 8) * 8 * max_tiles; }
// Which (segment, 64-row tile) does linear tile index `t` of this sample denote?  seg = -1: none (block exits).
__device__ __forceinline__ TileSel select_tile(const int* len, int nseg, int t) {
    TileSel r; r.seg = -1; r.t0 = 0; r.n = 0;
    for (int s = 0; s < nseg; ++s) {
        const int L = len[s];
        const int nt = (L + 63) >> 6;
        if (t < nt) { r.seg = s; r.t0 = t * 64; r.n = min(64, L - t * 64); return r; }
        t -= nt;
    }
    return r;
}


// bf16 fast path (mha_bf16.hip)
int mha_bf16_fwd(const MhaDesc& d, int head_dim, int variant, hipStream_t st);
int mha_bf16_bwd(MhaDesc d, int head_dim, int max_q_tiles, int max_k_tiles, int variant, hipStream_t st);
// sample-head kernels (mha_sh.hip): bf16, head_dim 64; d.max_tiles = key tiles per sample (upper bound)
bool mha_sh_applicable(const MhaDesc& d);
int mha_sh_fwd(const MhaDesc& d, int mode, hipStream_t st);
bool mha_sh_dkdv_supported(const MhaDesc& d);
bool mha_sh_dq_supported(const MhaDesc& d);
int mha_sh_dq(const MhaDesc& d, int mode, hipStream_t st);
bool mha_sh_fused_supported(const MhaDesc& d);
int mha_sh_bwd_fused(const MhaDesc& d, int mode, hipStream_t st);   // dQ + dK + dV in one kernel (+ the row-constant pre-pass); needs d.dq_ws; mode 1..3: diagnostics
int mha_sh_dkdv(const MhaDesc& d, int mode, hipStream_t st);   // needs workspace planes 1, 2 (see mha_bf16_bwd_dq_kernel)   // mode 0: product; 1 / 2: stream-only / compute-only diagnostics
