// Go / no-go probe for an own FF1 GEMM: C[M, N] = A[M, K] . W[N, K]^T, bf16 in / bf16 out, fp32 accumulate,
// at the encoder feed-forward's first projection, M = 163 840 rows (B 256 x S 640), N = 4096 (2 x ffi), K = 768.
// The gate: the BARE kernel (no GEGLU epilogue yet) must reach 1.1 PFLOP/s in isolation, or the fusion cannot pay back what it
// loses against the tuned library GEMM.
//
// Structure (the "step-3" structure of cdna_hip_programming.md section 5 with the staging of csrc/mha_sh.hip): 128 x 128 tile,
// BK = 64, 4 waves (each 64 x 64 = 2 x 2 MFMA 32x32x16 tiles), A and W tiles by LDS-DMA (buffer_load ... lds, 1 KiB pieces,
// source-side swizzle so that the b128 fragment reads are conflict free), two LDS buffers, ONE raw s_barrier per K-step with the
// next tile's DMA in flight under the MFMAs (counted vmcnt), two blocks per CU.  The product is computed transposed
// (W rows on the MFMA's M side) so that a lane owns one output row and 4 consecutive columns per register group.
//
//   hipcc --offload-arch=gfx950 -O3 tools/probes/gemm_ff1_probe.hip -o /tmp/gemm_probe && /tmp/gemm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#define LDS_AS __attribute__((address_space(3)))

__device__ __forceinline__ int swz(int r) { return (((r >> 1) & 1) << 2) | (((r >> 2) & 1) << 1) | ((r >> 3) & 1); }

__device__ __forceinline__ void dma(const bf16* base, int voff, int soff, bf16* lds_piece) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    bf16* ub = reinterpret_cast<bf16*>(((unsigned long long)hi << 32) | lo);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(ub, 0, 0x7fffffff, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_AS void*)lds_piece, 16, voff, __builtin_amdgcn_readfirstlane(soff), 0, 0);
}

template <int K>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const bf16* __restrict__ A, const bf16* __restrict__ W, bf16* __restrict__ C, int M, int N) {
    __shared__ __attribute__((aligned(1024))) bf16 lds[2][2][128 * 64];          // [buffer][A | W][128 rows x 64]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, hh = lane >> 5;
    // XCD-aware tile order: consecutive blocks of one XCD walk the N tiles of one M tile (A tile stays in that XCD's L2)
    const int ntn = N / 128, nblk = gridDim.x, xcd = blockIdx.x & 7, q = nblk / 8;
    const int lin = xcd * q + (blockIdx.x >> 3);
    const int tm = lin / ntn, tn = lin % ntn;
    const bf16* Ab = A + (long)tm * 128 * K;
    const bf16* Wb = W + (long)tn * 128 * K;
    const int wm = wave >> 1, wn = wave & 1;                                      // wave's 64 x 64 quadrant: rows of C (m), cols (n)
    // DMA: wave w stages rows [32 w, 32 w + 32) of both tiles: 4 pieces of 8 rows each
    const int prow = lane >> 3;
    const int voff = prow * (K * 2) + 16 * ((lane & 7) ^ swz(prow));
    auto stage = [&](int kt, int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = 32 * wave + 8 * j;
            dma(Ab + kt * 64, voff ^ (16 * (j & 1)), row * K * 2, &lds[buf][0][row * 64]);
            dma(Wb + kt * 64, voff ^ (16 * (j & 1)), row * K * 2, &lds[buf][1][row * 64]);
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int frag = r * 64 + 8 * (hh ^ swz(r));                                 // row r of a 32-row block, k-step ks: frag ^ (16 ks)
    constexpr int NK = K / 64;
    stage(0, 0);
    for (int kt = 0; kt < NK; ++kt) {
        const int buf = kt & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 1 < NK) stage(kt + 1, buf ^ 1);
        const bf16* As = lds[buf][0] + 64 * 64 * wm;
        const bf16* Ws = lds[buf][1] + 64 * 64 * wn;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 af[2], wf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[i] = *reinterpret_cast<const bf16x8*>(As + 2048 * i + (frag ^ (16 * ks)));
                wf[i] = *reinterpret_cast<const bf16x8*>(Ws + 2048 * i + (frag ^ (16 * ks)));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);   // D[n][m]
        }
    }
    // epilogue: lane (m = r, hh) of block (i, j) holds C[m][n = 8 (e >> 2) + 4 hh + (e & 3)]
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        bf16* crow = C + ((long)tm * 128 + 64 * wm + 32 * i + r) * N + tn * 128 + 64 * wn + 4 * hh;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16 o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (bf16)acc[i][j][4 * g + e];
                *reinterpret_cast<u32x2*>(crow + 32 * j + 8 * g) = *reinterpret_cast<u32x2*>(o);
            }
    }
}

int main() {
    const int M = 163840, N = 4096, K = 768;
    std::vector<bf16> hA((size_t)M * K), hW((size_t)N * K);
    srand(1);
    for (auto& x : hA) x = (bf16)((rand() % 2001 - 1000) / 1000.0f);
    for (auto& x : hW) x = (bf16)((rand() % 2001 - 1000) / 1000.0f);
    bf16 *A, *W, *C;
    hipMalloc(&A, hA.size() * 2); hipMalloc(&W, hW.size() * 2); hipMalloc(&C, (size_t)M * N * 2);
    hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(W, hW.data(), hW.size() * 2, hipMemcpyHostToDevice);
    const dim3 grid((M / 128) * (N / 128)), blk(256);
    gemm_kernel<K><<<grid, blk>>>(A, W, C, M, N);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
    std::vector<bf16> hC((size_t)4096 * 0 + 1);
    // spot check 64 entries
    double worst = 0;
    for (int t = 0; t < 64; ++t) {
        const long m = (long)(rand() % M), n = rand() % N;
        bf16 got;
        hipMemcpy(&got, C + m * N + n, 2, hipMemcpyDeviceToHost);
        double ref = 0;
        for (int k = 0; k < K; ++k) ref += (double)(float)hA[m * K + k] * (double)(float)hW[n * K + k];
        worst = fmax(worst, fabs((double)(float)got - ref) / fmax(1.0, fabs(ref)));
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) gemm_kernel<K><<<grid, blk>>>(A, W, C, M, N);
    hipEventRecord(e0);
    const int it = 20;
    for (int i = 0; i < it; ++i) gemm_kernel<K><<<grid, blk>>>(A, W, C, M, N);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double tf = 2.0 * M * N * K / (ms / it * 1e-3) / 1e12;
    printf("own FF1 GEMM probe  M %d N %d K %d : %.1f us per launch = %.0f TFLOP/s (gate 1100)   spot-check max rel err %.2e\n",
           M, N, K, ms / it * 1e3, tf, worst);
    return 0;
}
