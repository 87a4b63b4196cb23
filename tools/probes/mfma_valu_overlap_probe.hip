// Probe (gfx950): do MFMA and VALU work overlap on a SIMD?  256 workgroups x 16 waves (4 per SIMD); every wave runs ITERS rounds of
//   mode 0: 16 independent v_mfma_f32_32x32x16_bf16 (4 accumulators)            -> MFMA pipe only
//   mode 1: 64 v_exp_f32 + 64 v_fma_f32 on independent registers                 -> VALU only
//   mode 2: both blocks back to back in every wave (same-wave mix)
//   mode 3: even waves run the MFMA block, odd waves the VALU block (cross-wave mix; per-SIMD work = half of each)
// If the pipes overlap, t(2) ~ max(t(0), t(1)) and t(3) ~ max/2; if they serialise, t(2) ~ t(0) + t(1).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_valu_overlap_probe.hip -o /tmp/ovl && /tmp/ovl
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define ITERS 2000
template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out) {
    const int wave = threadIdx.x >> 6;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.01f * (threadIdx.x + j)); b[j] = (__bf16)(0.02f * j); }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    float v[16];
    for (int j = 0; j < 16; ++j) v[j] = 0.001f * (threadIdx.x + j);
    const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && (wave & 1) == 0);
    const bool do_v = MODE == 1 || MODE == 2 || (MODE == 3 && (wave & 1) == 1);
    for (int it = 0; it < ITERS; ++it) {
        if (do_m) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
        }
        if (do_v) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 16; ++j) { v[j] = __builtin_amdgcn_exp2f(v[j]); v[j] = __builtin_fmaf(v[j], 0.5f, -0.25f); }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    for (int j = 0; j < 16; ++j) s += v[j];
    if (s == 123.456f) out[threadIdx.x] = s;
}
template <int MODE> float run(float* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, d); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, d); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    float* d; hipMalloc(&d, 4096);
    const float t0 = run<0>(d), t1 = run<1>(d), t2 = run<2>(d), t3 = run<3>(d);
    // per SIMD per iteration: 4 waves x 16 MFMA x 32 cycles = 2048 cycles (mode 0); 4 waves x (64 exp x 16 + 64 fma x 4) = 5120 (mode 1)
    printf("mfma only %.3f ms | valu only %.3f ms | both in every wave %.3f ms (sum %.3f, max %.3f) | split across waves %.3f ms (half-sum %.3f, half-max %.3f)\n",
           t0, t1, t2, t0 + t1, t0 > t1 ? t0 : t1, t3, 0.5f * (t0 + t1), 0.5f * (t0 > t1 ? t0 : t1));
    printf("cycles per iteration per SIMD at 2.4 GHz: mfma %.0f valu %.0f both %.0f split %.0f\n", t0 * 2.4e6 / ITERS, t1 * 2.4e6 / ITERS,
           t2 * 2.4e6 / ITERS, t3 * 2.4e6 / ITERS);
    return 0;
}
