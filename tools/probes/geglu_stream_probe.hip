// Which launch shape lets the GEGLU row kernels reach the streaming rate of the chip?  (torch's elementwise add moves
// 2.7 GB at 6.25 TB/s on this GPU; geglu_fwd/bwd run at 5.2.)  Variants of the same arithmetic:
//   U   vectors per thread in flight (1 = the shipped kernel), G = 0 grid-stride with 4096 blocks, 1 = one pass (n/(256*U) blocks)
//   NT  nontemporal loads/stores
// hipcc --offload-arch=gfx950 -O3 tools/probes/geglu_stream_probe.hip -o /tmp/geglu_probe && /tmp/geglu_probe
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <cstdio>
typedef __bf16 bf16;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
struct GeluParts { float cdf, pdf; };
__device__ __forceinline__ GeluParts gelu_parts(float x) {
    const float e = __builtin_amdgcn_exp2f(-0.72134752044448170f * x * x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.23164189f, fabsf(x), 1.f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f); poly = fmaf(poly, t, -0.284496736f); poly = fmaf(poly, t, 0.254829592f);
    const float half_erfc = 0.5f * poly * t * e;
    GeluParts r; r.cdf = x >= 0.f ? 1.f - half_erfc : half_erfc; r.pdf = 0.39894228040143268f * e; return r;
}
template <bool NT> __device__ __forceinline__ void ld8(const bf16* p, float* o) {
    union { u32x4 q; bf16 e[8]; } u;
    if constexpr (NT) u.q = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)); else u.q = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (float)u.e[j];
}
template <bool NT> __device__ __forceinline__ void st8(bf16* p, const float* o) {
    union { u32x4 q; bf16 e[8]; } u;
#pragma unroll
    for (int j = 0; j < 8; ++j) u.e[j] = (bf16)o[j];
    if constexpr (NT) __builtin_nontemporal_store(u.q, reinterpret_cast<u32x4*>(p)); else *reinterpret_cast<u32x4*>(p) = u.q;
}
// F = 2048 -> 256 vectors per row: r = i >> 8 (the probe fixes the shape; the product kernel divides)
template <int U, bool NT> __global__ __launch_bounds__(256) void fwd(const bf16* __restrict__ h, bf16* __restrict__ out, long n, int F, int sh) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += stride * U) {
        float val[U][8], gate[U][8];
#pragma unroll
        for (int u = 0; u < U; ++u) { const long i = i0 + u * stride; if (i < n) { const long r = i >> sh; const int c = (int)(i - (r << sh)) * 8;
            ld8<NT>(h + r * 2 * F + c, val[u]); ld8<NT>(h + r * 2 * F + F + c, gate[u]); } }
#pragma unroll
        for (int u = 0; u < U; ++u) { const long i = i0 + u * stride; if (i < n) { const long r = i >> sh; const int c = (int)(i - (r << sh)) * 8; float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = gate[u][j] * gelu_parts(gate[u][j]).cdf * val[u][j];
            st8<NT>(out + r * F + c, o); } }
    }
}
template <int U, bool NT> __global__ __launch_bounds__(256) void bwd(const bf16* __restrict__ h, const bf16* __restrict__ g, bf16* __restrict__ dh, long n, int F, int sh) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += stride * U) {
        float val[U][8], gate[U][8], gg[U][8];
#pragma unroll
        for (int u = 0; u < U; ++u) { const long i = i0 + u * stride; if (i < n) { const long r = i >> sh; const int c = (int)(i - (r << sh)) * 8;
            ld8<NT>(h + r * 2 * F + c, val[u]); ld8<NT>(h + r * 2 * F + F + c, gate[u]); ld8<NT>(g + r * F + c, gg[u]); } }
#pragma unroll
        for (int u = 0; u < U; ++u) { const long i = i0 + u * stride; if (i < n) { const long r = i >> sh; const int c = (int)(i - (r << sh)) * 8; float dv[8], dg[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { const GeluParts gp = gelu_parts(gate[u][j]); dv[j] = gg[u][j] * gate[u][j] * gp.cdf; dg[j] = gg[u][j] * val[u][j] * fmaf(gate[u][j], gp.pdf, gp.cdf); }
            st8<NT>(dh + r * 2 * F + c, dv); st8<NT>(dh + r * 2 * F + F + c, dg); } }
    }
}
__global__ void fill(bf16* p, long n) { for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) p[i] = (bf16)(((i * 2654435761u) & 1023) / 512.f - 1.f); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main() {
    const long rows = 163840; const int F = 2048, sh = 8; const long n = rows * (F / 8);
    bf16 *h, *out, *g, *dh;
    CK(hipMalloc(&h, rows * 2 * F * 2)); CK(hipMalloc(&dh, rows * 2 * F * 2)); CK(hipMalloc(&out, rows * F * 2)); CK(hipMalloc(&g, rows * F * 2));
    fill<<<4096, 256>>>(h, rows * 2 * F); fill<<<4096, 256>>>(g, rows * F);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double fb = rows * 3.0 * F * 2, bb = rows * 5.0 * F * 2;
#define RUN(name, bytes, grid, ...) { for (int w = 0; w < 3; ++w) { __VA_ARGS__; } CK(hipEventRecord(e0)); for (int it = 0; it < 20; ++it) { __VA_ARGS__; } CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); \
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); printf("%-34s grid %7d  %.1f us  %.2f TB/s\n", name, (int)(grid), ms / 20 * 1e3, bytes / (ms / 20 * 1e-3) / 1e12); }
#define BOTH(U, NT, G) { const int grid = G ? (int)((n + 256L * U - 1) / (256L * U)) : 4096; char nm[64]; \
        snprintf(nm, 64, "fwd U=%d NT=%d %s", U, NT, G ? "one-pass" : "grid-stride"); RUN(nm, fb, grid, (fwd<U, NT><<<grid, 256>>>(h, out, n, F, sh))); \
        snprintf(nm, 64, "bwd U=%d NT=%d %s", U, NT, G ? "one-pass" : "grid-stride"); RUN(nm, bb, grid, (bwd<U, NT><<<grid, 256>>>(h, g, dh, n, F, sh))); }
    BOTH(1, false, 0) BOTH(2, false, 0) BOTH(4, false, 0) BOTH(1, false, 1) BOTH(2, false, 1) BOTH(4, false, 1)
    BOTH(1, true, 0) BOTH(2, true, 0) BOTH(2, true, 1) BOTH(4, true, 1)
    { const int grids[] = {1024, 2048, 8192, 16384}; for (int gi = 0; gi < 4; ++gi) { const int grid = grids[gi]; char nm[64];
        snprintf(nm, 64, "fwd U=2 NT=0 grid-stride"); RUN(nm, fb, grid, (fwd<2, false><<<grid, 256>>>(h, out, n, F, sh)));
        snprintf(nm, 64, "bwd U=2 NT=0 grid-stride"); RUN(nm, bb, grid, (bwd<2, false><<<grid, 256>>>(h, g, dh, n, F, sh))); } }
    return 0;
}
