// Issue-cost probe for VERDICT r5 item 1 (the 4-wave x 128x128 wave-tile GEMM with all 256 accumulators of a wave in the unified
// VGPR/AGPR file): ONE wave per SIMD has no partner that issues MFMAs while it reads LDS, issues LDS-DMA or waits at a barrier, so every
// non-MFMA instruction of a K-step must fit into the issue slots the wave's own MFMAs leave.  What does each ingredient cost there?
//
// 256 workgroups (one per CU) x 4 waves; a wave holds a 128 x 128 fp32 tile (acc[8][8] of v_mfma_f32_16x16x32_bf16, or acc[4][4] of
// v_mfma_f32_32x32x16_bf16: 256 registers either way) and runs ITERS "K-steps" of 32: 64 (resp. 32) MFMAs = 1024 matrix-pipe cycles, with
//   NRD  ds_read_b128 per K-step (16 = the step's 8 A + 8 W fragments, double-buffered in registers: read one step ahead),
//   NDMA buffer_load ... lds per K-step (8 = the wave's quarter of a 256x32 A slab + a 256x32 W slab; sources shared per XCD -> L2 hits),
//   BAR  one raw s_barrier per K-step.
// Reported per mode: ns per K-step, matrix-pipe utilisation against the bare MFMA loop of the same build, chip TFLOP/s.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/gemm4w_issue_probe.hip -o /tmp/g4i && /tmp/g4i
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define LDS_AS __attribute__((address_space(3)))
#define ITERS 1536

// hipcc's own allocation of 256 accumulator registers + 128 fragment registers spills (340-530 B of scratch per lane, accumulators moved
// between the two halves of the file, MFMAs with D != C): the accumulators are pinned to the AGPR half by the operand constraint and
// every MFMA accumulates in place (cdna_hip_programming.md 5.7 item 4; operands come from ds_read, an accumulate chain needs no wait states)
#define MFMA16(c, a, b) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b))
#define MFMA32(c, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b))
__device__ __forceinline__ bf16x8 lds8(const char* p) { return *reinterpret_cast<const bf16x8*>(p); }

template <int MF, int NRD, int NDMA, int BAR>
__global__ __launch_bounds__(256, 1) void probe(const bf16* __restrict__ src, float* __restrict__ out, long long* __restrict__ clk) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(src) + (size_t)(blockIdx.x & 7) * 256 * 768, 0, 256 * 768 * 2, 0x00020000);
    const int voff = (lane >> 3) * 1536 + 16 * ((lane & 7) ^ (lane >> 3));
    // fill the fragment area (lower 64 KB) with random operand data once
    for (int p = 0; p < 16; ++p)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_AS void*)(lds + (wave * 16 + p) * 1024), 16, voff, __builtin_amdgcn_readfirstlane(((wave * 16 + p) & 31) * 12288), 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int rd = (lane & 15) * 128 + 16 * ((lane >> 4) ^ (lane & 7));
    bf16x8 fa[2][8], fb[2][8];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 8; ++i) { fa[s][i] = lds8(lds + rd + (s * 8 + i) * 2048); fb[s][i] = lds8(lds + rd + 32768 + (s * 8 + i) * 2048 - (i & 1) * 64 + (i & 1) * 64); }
    f32x4 acc[8][8];
    f32x16 acc32[4][4];
    if constexpr (MF == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc32[i][j][e] = 0.f;
    }
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; it += 2) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {                        // step it + s computes from fragment set s and refills set s ^ 1 for step it + s + 1
            const int step = it + s;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if constexpr (MF == 0) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) MFMA16(acc[i][j], fa[s][i], fb[s][j]);
                } else {
                    // 32 MFMAs of 32x32x16 per step: (4 x 4 tiles) x 2 k-halves; operand registers reused as 32-row fragments
#pragma unroll
                    for (int j = 0; j < 4; ++j) MFMA32(acc32[i & 3][j], fa[s][i], fb[s][(i >> 2) * 4 + j]);
                }
                // the OTHER fragment set (consumed in the previous step, needed in the next one) is refilled under this step's MFMAs, two reads
                // behind every group of 8: the W fragments first (the next step's first MFMA group needs all of them), then the A fragments
                if (2 * i < NRD) {
                    if (i < 4) { fb[s ^ 1][2 * i] = lds8(lds + rd + 32768 + ((s ^ 1) * 8 + 2 * i) * 2048 + (step & 4) * 16); fb[s ^ 1][2 * i + 1] = lds8(lds + rd + 32768 + ((s ^ 1) * 8 + 2 * i + 1) * 2048 + (step & 4) * 16); }
                    else { fa[s ^ 1][2 * i - 8] = lds8(lds + rd + ((s ^ 1) * 8 + 2 * i - 8) * 2048 + (step & 4) * 16); fa[s ^ 1][2 * i - 7] = lds8(lds + rd + ((s ^ 1) * 8 + 2 * i - 7) * 2048 + (step & 4) * 16); }
                }
                if (i < NDMA) {
                    const int p = step * NDMA + i;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_AS void*)(lds + 65536 + wave * 16384 + (p & 15) * 1024), 16, voff,
                                                             __builtin_amdgcn_readfirstlane((p & 31) * 12288 + ((step >> 2) % 12) * 128), 0, 0);
                }
            }
            if (NDMA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA * 3 > 63 ? 63 : NDMA * 3) : "memory");      // ~3 steps of DMA stay in flight
            if (BAR) __builtin_amdgcn_s_barrier();
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // MFMA D -> compiler reads of the accumulators
    // accumulators leave one at a time (a reduction over all 256 values made hipcc copy them into the VGPR half together: spills inside the loop)
    f32x4* o4 = reinterpret_cast<f32x4*>(out) + (size_t)(blockIdx.x * 256 + threadIdx.x) * 64;
    if constexpr (MF == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) { o4[i * 8 + j] = acc[i][j]; __builtin_amdgcn_sched_barrier(0); }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o4[(i * 4 + j) * 4 + e] = f32x4{acc32[i][j][4 * e], acc32[i][j][4 * e + 1], acc32[i][j][4 * e + 2], acc32[i][j][4 * e + 3]};
                    __builtin_amdgcn_sched_barrier(0);
                }
    }
    if (lane == 0) clk[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int MF, int NRD, int NDMA, int BAR>
void run(const char* name, const bf16* src, float* out, long long* clk, double base_ns, double* ns_out) {
    const size_t ldsb = 131072;
    hipFuncSetAttribute((const void*)probe<MF, NRD, NDMA, BAR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((probe<MF, NRD, NDMA, BAR>), dim3(256), dim3(256), ldsb, 0, src, out, clk);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", name); return; }
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0);
        for (int i = 0; i < 4; ++i) hipLaunchKernelGGL((probe<MF, NRD, NDMA, BAR>), dim3(256), dim3(256), ldsb, 0, src, out, clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms / 4 < best) best = ms / 4;
    }
    std::vector<long long> h(1024);
    hipMemcpy(h.data(), clk, 1024 * sizeof(long long), hipMemcpyDeviceToHost);
    double cyc = 0; for (auto c : h) cyc += (double)c; cyc /= 1024.0;
    const double ns = best * 1e6 / ITERS, tf = 256.0 * 4 * 64 * 16 * 16 * 32 * 2 / (ns * 1e-9) / 1e12;
    printf("%-52s %7.1f ns / K-step  %6.0f shader clocks / K-step (1024 = matrix pipe full)  clock %.2f GHz  %5.0f TFLOP/s  %s\n", name, ns, cyc / ITERS,
           cyc / ITERS / ns, tf, base_ns > 0 ? "" : "(base)");
    if (base_ns > 0) printf("%-52s   = %.3f of the bare MFMA loop's rate\n", "", base_ns / ns);
    if (ns_out) *ns_out = ns;
}

int main() {
    std::vector<bf16> h((size_t)8 * 256 * 768);
    srand(3);
    for (auto& x : h) x = (bf16)((rand() % 2001 - 1000) / 1000.0f);
    bf16* src; float* out; long long* clk;
    hipMalloc(&src, h.size() * 2); hipMalloc(&out, (size_t)256 * 256 * 64 * 16); hipMalloc(&clk, 1024 * 8);
    hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    double b16 = 0, b32 = 0;
    printf("v_mfma_f32_16x16x32_bf16, 64 per K-step:\n");
    run<0, 0, 0, 0>("  bare MFMA loop", src, out, clk, 0, &b16);
    run<0, 16, 0, 0>("  + 16 ds_read_b128", src, out, clk, b16, nullptr);
    run<0, 0, 8, 0>("  + 8 LDS-DMA", src, out, clk, b16, nullptr);
    run<0, 0, 4, 0>("  + 4 LDS-DMA", src, out, clk, b16, nullptr);
    run<0, 0, 0, 1>("  + barrier", src, out, clk, b16, nullptr);
    run<0, 16, 8, 0>("  + 16 ds_read_b128 + 8 LDS-DMA", src, out, clk, b16, nullptr);
    run<0, 16, 8, 1>("  + 16 ds_read_b128 + 8 LDS-DMA + barrier (the K-step)", src, out, clk, b16, nullptr);
    printf("v_mfma_f32_32x32x16_bf16, 32 per K-step:\n");
    run<1, 0, 0, 0>("  bare MFMA loop", src, out, clk, 0, &b32);
    run<1, 16, 8, 1>("  + 16 ds_read_b128 + 8 LDS-DMA + barrier (the K-step)", src, out, clk, b32, nullptr);
    return 0;
}
