// Go / no-go probe, second attempt: the FF1 GEMM of the encoder's feed-forward, C[M, N] = A[M, K] . W[N, K]^T,
// bf16 in / bf16 out, fp32 accumulate, M = 163 840 (B 256 x S 640), N = 4096 (2 x ffi), K = 768 -- this time on the structure
// cdna_hip_programming.md section 5 says wins ("The 256^2 8-phase template"), written from that description:
//
//   256 x 256 x 64 tile, 8 waves (2 along M x 4 along N; a wave owns 128 x 64 of C in 128 accumulator registers), v_mfma_f32_16x16x32_bf16;
//   operands by LDS-DMA (buffer_load ... lds, 16 B per lane, 1 KiB per wave-instruction) into TWO 64 KB tile buffers, never through
//   registers; swizzle on the SOURCE address + the same XOR on the fragment read (b128 reads conflict-free: tools/probes/lds_swizzle_check.py
//   model); 4 phases per K-tile (one 64 x 32 quadrant of the wave's tile x K 64 = 16 MFMAs each), every phase = { ds_read the
//   quadrant's fragments, issue ONE staging step (2 DMA instructions per lane), [counted s_waitcnt vmcnt(N) -- never 0 in the loop],
//   raw s_barrier, MFMAs under s_setprio(1), raw s_barrier }; waves 4..7 run one barrier behind waves 0..3, so on every SIMD one
//   wave issues MFMAs while its partner reads LDS and issues DMA (the two barriers per phase are that hand-over);
//   bijective XCD-aware tile order (all tiles of an A row-panel group on one XCD's L2).
//
// Staging granule = a quarter tile (64 rows x 64 k of one operand half = 8 KB = one DMA per lane), two of them per phase.  Order and
// liveness (tile t lives in buffer t & 1; "dead after q" = last ds_read of that region is in phase q of its tile):
//     A-lo quarters (rows 0..63 of both 128-row halves): read in phase 0, dead after 0 -> restaged in phase 2 with tile t+2
//     W half 0 / half 1                                : read in phases 0, 1      -> restaged in phase 3 (t+2) / phase 0 of t+1 (t+2)
//     A-hi quarters                                    : read in phase 2          -> restaged in phase 1 of t+1 (t+2)
//   i.e. a region is restaged >= 2 phases after its last read (WAR), every read is >= 1 phase after the counted wait that retires
//   its DMA (RAW: the wait sits before the phase's first barrier), and 3-4 staging steps (6-8 DMA) stay in flight across barriers.
//
//   hipcc --offload-arch=gfx950 -O3 tools/probes/gemm8p_probe.hip -o /tmp/gemm8p && /tmp/gemm8p [M N]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define LDS_AS __attribute__((address_space(3)))

#define VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

__device__ __forceinline__ unsigned pack2(float lo, float hi) {
    const unsigned a = __builtin_bit_cast(unsigned short, (bf16)lo), b = __builtin_bit_cast(unsigned short, (bf16)hi);
    return a | (b << 16);
}

struct Ctx {
    __amdgpu_buffer_rsrc_t rsA, rsW;
    char* lds;                 // 2 x 64 KB: [buffer][A tile 256 x 64 | W tile 256 x 64], 128-byte rows
    int voffA, voffW;          // per-lane source offsets of a DMA piece (8 rows x 128 B), swizzle folded in
    int sA, sW;                // scalar byte offsets of this wave's first piece row in A / W (tile origin + 8 * wave rows)
    int rdA[2], rdW[2];        // per-lane fragment read offsets (k-step 0 / 1) inside a buffer
};

template <int K>
__device__ __forceinline__ void dma(const __amdgpu_buffer_rsrc_t& rs, int voff, int soff, char* dst) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_AS void*)dst, 16, voff, __builtin_amdgcn_readfirstlane(soff), 0, 0);
}
// A quarter pair x (0: rows 0..63 of both halves, 1: rows 64..127) of K-tile kt into buffer buf
template <int K>
__device__ __forceinline__ void stageA(const Ctx& c, int kt, int buf, int hi, int wave) {
    const int row = 64 * hi + 8 * wave;
    dma<K>(c.rsA, c.voffA, c.sA + (64 * hi * K + kt * 64) * 2, c.lds + buf * 65536 + row * 128);
    dma<K>(c.rsA, c.voffA, c.sA + ((128 + 64 * hi) * K + kt * 64) * 2, c.lds + buf * 65536 + (128 + row) * 128);
}
// W half (0: tile rows 0..127, 1: rows 128..255) of K-tile kt into buffer buf
template <int K>
__device__ __forceinline__ void stageW(const Ctx& c, int kt, int buf, int half, int wave) {
    const int row = 128 * half + 8 * wave;
    dma<K>(c.rsW, c.voffW, c.sW + (128 * half * K + kt * 64) * 2, c.lds + buf * 65536 + 32768 + row * 128);
    dma<K>(c.rsW, c.voffW, c.sW + ((128 * half + 64) * K + kt * 64) * 2, c.lds + buf * 65536 + 32768 + (row + 64) * 128);
}

struct Frags {
    bf16x8 a[4][2];            // A fragments of the current 64-row group (b operand: column = m)
    bf16x8 wlo[2][2], whi[2][2];   // W fragments: n-tiles 0, 1 and 2, 3 (a operand: row = n)
};

__device__ __forceinline__ bf16x8 lds8(const char* p) { return *reinterpret_cast<const bf16x8*>(p); }

template <int Q>
__device__ __forceinline__ void load_frags(const Ctx& c, int buf, Frags& f) {
    const char* base = c.lds + buf * 65536;
    if (Q == 0) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) f.wlo[ni][ks] = lds8(base + c.rdW[ks] + ni * 512);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (Q == 1) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) f.whi[ni][ks] = lds8(base + c.rdW[ks] + (2 + ni) * 512);
    }
    if (Q == 0 || Q == 2) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) f.a[mi][ks] = lds8(base + c.rdA[ks] + ((Q == 2 ? 4 : 0) + mi) * 2048);
    }
}

// quadrant Q of the wave's tile: (m group, n pair) = (0,0) (0,1) (1,1) (1,0); acc[ni][mi] holds C^T tiles (rows n, column m)
template <int Q>
__device__ __forceinline__ void mfma_phase(const Frags& f, f32x4 (&acc)[4][8]) {
    constexpr int qa = Q >= 2, qb = (Q == 1 || Q == 2);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
                acc[2 * qb + ni][4 * qa + mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qb ? f.whi[ni][ks] : f.wlo[ni][ks], f.a[mi][ks],
                                                                                      acc[2 * qb + ni][4 * qa + mi], 0, 0, 0);
}

#define PHASE_TAIL(Q)                                                      \
    __builtin_amdgcn_sched_barrier(0);                                     \
    __builtin_amdgcn_s_barrier();                                          \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                     \
    __builtin_amdgcn_sched_barrier(0);                                     \
    __builtin_amdgcn_s_setprio(1);                                         \
    mfma_phase<Q>(f, acc);                                                 \
    __builtin_amdgcn_s_setprio(0);                                         \
    __builtin_amdgcn_sched_barrier(0);                                     \
    __builtin_amdgcn_s_barrier();                                          \
    __builtin_amdgcn_sched_barrier(0);

// one K-tile (4 phases).  N1: tile t+1 exists, N2: tile t+2 exists (compile time: the counted waits depend on them)
template <int K, bool N1, bool N2>
__device__ __forceinline__ void ktile(const Ctx& c, int t, int b, int wave, Frags& f, f32x4 (&acc)[4][8]) {
    load_frags<0>(c, b, f);
    if (N1) stageW<K>(c, t + 1, b ^ 1, 1, wave);
    PHASE_TAIL(0)
    load_frags<1>(c, b, f);
    if (N1) stageA<K>(c, t + 1, b ^ 1, 1, wave);
    if (N1) VMCNT(8); else VMCNT(0);                       // A-hi of THIS tile (issued 4 staging steps ago) has landed
    PHASE_TAIL(1)
    load_frags<2>(c, b, f);
    if (N2) stageA<K>(c, t + 2, b, 0, wave);
    PHASE_TAIL(2)
    if (N2) stageW<K>(c, t + 2, b, 0, wave);
    if (N2) VMCNT(6); else if (N1) VMCNT(2);               // A-lo, W half 0, W half 1 of tile t+1 have landed
    PHASE_TAIL(3)
}

template <int K>
__global__ __launch_bounds__(512, 2) void gemm8p_kernel(const bf16* __restrict__ A, const bf16* __restrict__ W, bf16* __restrict__ C, int M, int N) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    // bijective XCD-aware order: XCD x (= blockIdx % 8) owns M-panels [x TMX, (x+1) TMX) and walks them in groups of GN N-tiles
    const int ntn = N / 256, ntm = M / 256, nblk = ntm * ntn;
    int tm, tn;
    {
        constexpr int GN = 8;
        const int x = blockIdx.x & 7, idx = blockIdx.x >> 3;
        if (ntm % 8 == 0 && ntn % GN == 0) {
            const int tmx = ntm / 8, per = tmx * GN;
            const int half = idx / per, rem = idx % per;
            tm = x * tmx + rem / GN; tn = half * GN + rem % GN;
        } else {
            const int q = nblk / 8, r = nblk % 8;
            const int lin = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + idx;
            tm = lin / ntn; tn = lin % ntn;
        }
    }
    Ctx c;
    c.lds = lds;
    c.rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(A), 0, (unsigned)((long)M * K * 2), 0x00020000);
    c.rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(W), 0, (unsigned)((long)N * K * 2), 0x00020000);
    {
        const int pr = lane >> 3, ch = lane & 7;
        c.voffA = pr * K * 2 + 16 * (ch ^ pr);
        c.voffW = pr * K * 2 + 16 * (ch ^ ((((wave >> 1) & 3) << 1) | ((lane >> 4) & 1)));
        c.sA = (tm * 256 + 8 * wave) * K * 2;
        c.sW = (tn * 256 + 8 * wave) * K * 2;
        const int i = lane & 15, g = lane >> 4;
        const int ra = (128 * wr + i) * 128 + 16 * (g ^ (i & 7));
        const int fw = ((i >> 2) << 1) | ((i >> 1) & 1);
        const int rw = 32768 + (64 * wc + 16 * (i >> 2) + (i & 3)) * 128 + 16 * (g ^ fw);
        c.rdA[0] = ra; c.rdA[1] = ra ^ 64; c.rdW[0] = rw; c.rdW[1] = rw ^ 64;
    }
    f32x4 acc[4][8];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
    Frags f;
    constexpr int NT = K / 64;
    static_assert(NT >= 4 && NT % 2 == 0, "K must be a multiple of 128, >= 256");
    // prologue: tile 0 complete, A-lo and W half 0 of tile 1; the order of the steady state
    stageA<K>(c, 0, 0, 0, wave); stageW<K>(c, 0, 0, 0, wave); stageW<K>(c, 0, 0, 1, wave); stageA<K>(c, 0, 0, 1, wave);
    stageA<K>(c, 1, 1, 0, wave); stageW<K>(c, 1, 1, 0, wave);
    VMCNT(6);
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();             // waves 4..7 run one barrier behind (MFMA of one group over the other's loads)
    for (int t = 0; t < NT - 2; t += 2) {
        ktile<K, true, true>(c, t, 0, wave, f, acc);
        ktile<K, true, true>(c, t + 1, 1, wave, f, acc);
    }
    ktile<K, true, false>(c, NT - 2, 0, wave, f, acc);
    ktile<K, false, false>(c, NT - 1, 1, wave, f, acc);
    if (wr == 0) __builtin_amdgcn_s_barrier();
    // epilogue: lane (m = lane & 15, g = lane >> 4) of tile (ni, mi) holds C[m][16 g + 4 ni + reg]: 16 consecutive columns per row
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
        bf16* crow = C + ((long)tm * 256 + 128 * wr + 16 * mi + (lane & 15)) * N + tn * 256 + 64 * wc + 16 * (lane >> 4);
        u32x4 lo, hi;
        lo[0] = pack2(acc[0][mi][0], acc[0][mi][1]); lo[1] = pack2(acc[0][mi][2], acc[0][mi][3]);
        lo[2] = pack2(acc[1][mi][0], acc[1][mi][1]); lo[3] = pack2(acc[1][mi][2], acc[1][mi][3]);
        hi[0] = pack2(acc[2][mi][0], acc[2][mi][1]); hi[1] = pack2(acc[2][mi][2], acc[2][mi][3]);
        hi[2] = pack2(acc[3][mi][0], acc[3][mi][1]); hi[3] = pack2(acc[3][mi][2], acc[3][mi][3]);
        *reinterpret_cast<u32x4*>(crow) = lo;
        *reinterpret_cast<u32x4*>(crow + 8) = hi;
    }
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 163840, N = argc > 2 ? atoi(argv[2]) : 4096;
    constexpr int K = 768;
    if (M % 256 || N % 256) { printf("M, N must be multiples of 256\n"); return 1; }
    std::vector<bf16> hA((size_t)M * K), hW((size_t)N * K);
    srand(1);
    for (auto& x : hA) x = (bf16)((rand() % 2001 - 1000) / 1000.0f);
    for (auto& x : hW) x = (bf16)((rand() % 2001 - 1000) / 1000.0f);
    bf16 *A, *W, *C;
    hipMalloc(&A, hA.size() * 2); hipMalloc(&W, hW.size() * 2); hipMalloc(&C, (size_t)M * N * 2);
    hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(W, hW.data(), hW.size() * 2, hipMemcpyHostToDevice);
    hipMemset(C, 0xff, (size_t)M * N * 2);
    const dim3 grid((M / 256) * (N / 256)), blk(512);
    const size_t ldsb = 131072;
    if (hipFuncSetAttribute((const void*)gemm8p_kernel<K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb) != hipSuccess) { printf("LDS attribute failed\n"); return 1; }
    gemm8p_kernel<K><<<grid, blk, ldsb>>>(A, W, C, M, N);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
    // check: 3 whole tiles (first, last, one in the middle) element by element + 512 random entries
    double worst = 0; long bad = 0, checked = 0;
    auto check = [&](long m, long n, bf16 got) {
        double ref = 0;
        for (int k = 0; k < K; ++k) ref += (double)(float)hA[m * K + k] * (double)(float)hW[n * K + k];
        const double e = fabs((double)(float)got - ref) / fmax(1.0, fabs(ref));
        worst = fmax(worst, e); bad += e > 1e-2; ++checked;
    };
    std::vector<bf16> row(N);
    const long tiles[3][2] = {{0, 0}, {M / 256 - 1, N / 256 - 1}, {(M / 256) / 2 + 1, (N / 256) / 2 - 1}};
    for (auto& tl : tiles)
        for (int r = 0; r < 256; r += 1) {
            const long m = tl[0] * 256 + r;
            hipMemcpy(row.data(), C + m * N, (size_t)N * 2, hipMemcpyDeviceToHost);
            for (int cc = 0; cc < 256; ++cc) check(m, tl[1] * 256 + cc, row[tl[1] * 256 + cc]);
        }
    for (int t = 0; t < 512; ++t) {
        const long m = (long)(rand() % M), n = rand() % N;
        bf16 got; hipMemcpy(&got, C + m * N + n, 2, hipMemcpyDeviceToHost);
        check(m, n, got);
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) gemm8p_kernel<K><<<grid, blk, ldsb>>>(A, W, C, M, N);
    float best = 1e9f, tot = 0;
    const int rounds = 5, it = 10;
    for (int r = 0; r < rounds; ++r) {
        hipEventRecord(e0);
        for (int i = 0; i < it; ++i) gemm8p_kernel<K><<<grid, blk, ldsb>>>(A, W, C, M, N);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = fminf(best, ms / it); tot += ms / it;
    }
    const double fl = 2.0 * M * N * K;
    printf("own 256x256x64 8-phase GEMM  M %d N %d K %d : mean %.1f us = %.0f TFLOP/s, best round %.1f us = %.0f TFLOP/s (gate 1150; random [-1,1) operands)\n"
           "  check: %ld entries, %ld beyond 1e-2, max rel err %.2e\n",
           M, N, K, tot / rounds * 1e3, fl / (tot / rounds * 1e-3) / 1e12, best * 1e3, fl / (best * 1e-3) / 1e12, checked, bad, worst);
    return bad ? 2 : 0;
}
