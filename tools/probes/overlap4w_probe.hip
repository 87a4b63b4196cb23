// Can an HBM-bound row kernel run UNDER a GEMM?  The 8-wave product GEMM fills every SIMD's register file (2 waves x 248 registers) and
// tools/probes/overlap_probe.py found what follows from that: a GEMM and a streaming kernel on two streams take the sum of their times.
// The 4-wave GEMM of gemm4w_probe.hip (one wave per SIMD: 120 VGPRs + 256 AGPRs = 376 of 512 registers, 128 of 160 KB of LDS) leaves 136
// registers per lane and 24 wave slots per CU free -- room for a streaming kernel of <= 128 registers beside it.  This probe measures, on
// two streams: the GEMM alone, a LayerNorm-like streaming kernel alone (reads 6 B and writes 6 B per element of a 163 840 x 768 matrix,
// one wave per row, ~1.5 GB per launch), and both at once -- for the 4-wave and for the 8-wave GEMM.  If the 4-wave pair overlaps, the
// 32 ms of HBM-bound kernels of the step could hide under its 94 ms of GEMMs (two half-batches in flight); the structural "next" after
// the kernels themselves have hit their ceilings.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/overlap4w_probe.hip -o /tmp/ovl4 && /tmp/ovl4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <algorithm>
#define PROBE_NO_MAIN
namespace w8 {
#include "gemm8p_probe.hip"
}
#undef VMCNT
#undef STORE_AUX
#undef LDS_AS
#undef PRIO
#undef STAGGER
namespace w4 {
#include "gemm4w_probe.hip"
}

typedef float f32x4_ __attribute__((ext_vector_type(4)));
// one wave per row of 768: x (fp32) + delta (bf16) -> x_new (fp32) + y = bf16(normalised x_new): the byte pattern of add_ln_fwd
__global__ __launch_bounds__(256, 2) void stream_rows(const float* __restrict__ x, const __bf16* __restrict__ d, float* __restrict__ xo, __bf16* __restrict__ y, long rows) {
    const int lane = threadIdx.x & 63;
    const long row0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6));
    for (long r = row0; r < rows; r += (long)gridDim.x * 4) {
        float v[12];
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const f32x4_ a = __builtin_nontemporal_load(reinterpret_cast<const f32x4_*>(x + r * 768 + 256 * c + 4 * lane));
#pragma unroll
            for (int i = 0; i < 4; ++i) { v[4 * c + i] = a[i] + (float)d[r * 768 + 256 * c + 4 * lane + i]; s += v[4 * c + i]; q += v[4 * c + i] * v[4 * c + i]; }
        }
        for (int o = 32; o; o >>= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
        const float mu = s * (1.f / 768), rs = rsqrtf(q * (1.f / 768) - mu * mu + 1e-5f);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            f32x4_ a;
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = v[4 * c + i]; y[r * 768 + 256 * c + 4 * lane + i] = (__bf16)((v[4 * c + i] - mu) * rs); }
            __builtin_nontemporal_store(a, reinterpret_cast<f32x4_*>(xo + r * 768 + 256 * c + 4 * lane));
        }
    }
}

int main() {
    const int M = 163840, N = 4096, K = 768;
    typedef __bf16 bf;
    std::vector<bf> hA((size_t)M * K), hW((size_t)N * K);
    srand(1);
    for (auto& v : hA) v = (bf)((rand() % 2001 - 1000) / 1000.0f);
    for (auto& v : hW) v = (bf)((rand() % 2001 - 1000) / 1000.0f);
    bf *A, *W, *C, *D, *Y; float *X, *XO;
    hipMalloc(&A, hA.size() * 2); hipMalloc(&W, hW.size() * 2); hipMalloc(&C, (size_t)M * N * 2);
    hipMalloc(&D, (size_t)M * 768 * 2); hipMalloc(&Y, (size_t)M * 768 * 2); hipMalloc(&X, (size_t)M * 768 * 4); hipMalloc(&XO, (size_t)M * 768 * 4);
    hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice); hipMemcpy(W, hW.data(), hW.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(D, hA.data(), (size_t)M * 768 * 2, hipMemcpyHostToDevice);
    hipMemset(X, 0, (size_t)M * 768 * 4);
    hipFuncSetAttribute((const void*)w4::gemm4w_kernel<768>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute((const void*)w8::gemm8p_persist_kernel<768>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    hipEvent_t e0, e1, e2; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
    const int REP = 6;                                        // launches of each kernel per measurement
    auto gemm4 = [&](hipStream_t s) { w4::gemm4w_kernel<768><<<256, 256, 131072, s>>>(A, W, C, M, N, 0); };
    auto gemm8 = [&](hipStream_t s) { w8::gemm8p_persist_kernel<768><<<256, 512, 131072, s>>>(A, W, C, M, N, 0); };
    auto rows = [&](hipStream_t s, int blocks) { stream_rows<<<blocks, 256, 0, s>>>(X, D, XO, Y, M); };
    auto wall = [&](auto fa, auto fb, int na, int nb) {      // na launches of fa on stream 1, nb of fb on stream 2, started together
        hipDeviceSynchronize();
        hipEventRecord(e0, s1); hipStreamWaitEvent(s2, e0, 0);
        for (int i = 0; i < na; ++i) fa(s1);
        for (int i = 0; i < nb; ++i) fb(s2);
        hipEventRecord(e1, s1); hipEventRecord(e2, s2);
        hipEventSynchronize(e1); hipEventSynchronize(e2);
        float a, b; hipEventElapsedTime(&a, e0, e1); hipEventElapsedTime(&b, e0, e2);
        return std::max(a, b);
    };
    for (int blocks : {1024, 512, 2048}) {
        auto rb = [&](hipStream_t s) { rows(s, blocks); };
        auto none = [&](hipStream_t) {};
        for (int w = 0; w < 2; ++w) { wall(gemm4, rb, 2, 2); wall(gemm8, rb, 2, 2); }
        float g4 = 1e9f, g8 = 1e9f, r = 1e9f, b4 = 1e9f, b8 = 1e9f;
        for (int t = 0; t < 3; ++t) {
            g4 = std::min(g4, wall(gemm4, none, REP, 0)); g8 = std::min(g8, wall(gemm8, none, REP, 0));
            r = std::min(r, wall(none, rb, 0, 2 * REP));
            b4 = std::min(b4, wall(gemm4, rb, REP, 2 * REP)); b8 = std::min(b8, wall(gemm8, rb, REP, 2 * REP));
        }
        printf("streaming grid %4d blocks: %d GEMMs alone: 4-wave %.2f ms, 8-wave %.2f ms | %d row kernels alone %.2f ms (%.2f TB/s) | together: 4-wave %.2f ms "
               "(sum %.2f, max %.2f), 8-wave %.2f ms (sum %.2f, max %.2f)\n", blocks, REP, g4, g8, 2 * REP, r, 2.0 * REP * M * 768 * 12 / (r * 1e-3) / 1e12,
               b4, g4 + r, std::max(g4, r), b8, g8 + r, std::max(g8, r));
    }
    return 0;
}
