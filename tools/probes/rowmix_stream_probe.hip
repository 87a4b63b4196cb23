// Does the access WIDTH of the add+LayerNorm kernels (fp32 16 B/lane, bf16 8 B/lane, one wave per 768-wide row) cost
// bandwidth against a flat mapping where every access is 16 B?  Arithmetic reduced to x_new = x + d, y = bf16(x_new).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
// A: layout of add_ln_fwd: wave = row, lane owns 4 columns per 256-column chunk
__global__ __launch_bounds__(256) void rowwise(const float* x, const bf16* d, float* xn, bf16* y, long rows, int D) {
    const int lane = threadIdx.x & 63; const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int col = 4 * (lane + 64 * c);
        f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(x + row * D + col));
        union { u32x2 q; bf16 e[4]; } u; u.q = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(d + row * D + col));
        v[0] += (float)u.e[0]; v[1] += (float)u.e[1]; v[2] += (float)u.e[2]; v[3] += (float)u.e[3];
        __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(xn + row * D + col));
        union { u32x2 q; bf16 e[4]; } o; o.e[0] = (bf16)v[0]; o.e[1] = (bf16)v[1]; o.e[2] = (bf16)v[2]; o.e[3] = (bf16)v[3];
        __builtin_nontemporal_store(o.q, reinterpret_cast<u32x2*>(y + row * D + col));
    }
}
// B: flat, 8 elements per thread: one 16 B bf16 access, two 16 B fp32 accesses (32 B contiguous per lane)
__global__ __launch_bounds__(256) void flat8(const float* x, const bf16* d, float* xn, bf16* y, long n) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i >= n) return;
    f32x4 a = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(x + i)), b = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(x + i + 4));
    union { u32x4 q; bf16 e[8]; } u; u.q = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(d + i));
#pragma unroll
    for (int j = 0; j < 4; ++j) { a[j] += (float)u.e[j]; b[j] += (float)u.e[4 + j]; }
    __builtin_nontemporal_store(a, reinterpret_cast<f32x4*>(xn + i)); __builtin_nontemporal_store(b, reinterpret_cast<f32x4*>(xn + i + 4));
    union { u32x4 q; bf16 e[8]; } o;
#pragma unroll
    for (int j = 0; j < 4; ++j) { o.e[j] = (bf16)a[j]; o.e[4 + j] = (bf16)b[j]; }
    __builtin_nontemporal_store(o.q, reinterpret_cast<u32x4*>(y + i));
}
// C: flat, fp32 halves interleaved across the wave so each instruction is dense: lane l takes fp32 [16l, 16l+16) B of the
// wave's first 1 KB with instruction 0 and of the second 1 KB with instruction 1; bf16 stays 8 B/lane x 2
__global__ __launch_bounds__(256) void flat_dense(const float* x, const bf16* d, float* xn, bf16* y, long n) {
    const int lane = threadIdx.x & 63; const long wbase = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 512;   // 512 elements per wave
    if (wbase >= n) return;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const long i = wbase + h * 256 + lane * 4;
        f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(x + i));
        union { u32x2 q; bf16 e[4]; } u; u.q = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(d + i));
        v[0] += (float)u.e[0]; v[1] += (float)u.e[1]; v[2] += (float)u.e[2]; v[3] += (float)u.e[3];
        __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(xn + i));
        union { u32x2 q; bf16 e[4]; } o; o.e[0] = (bf16)v[0]; o.e[1] = (bf16)v[1]; o.e[2] = (bf16)v[2]; o.e[3] = (bf16)v[3];
        __builtin_nontemporal_store(o.q, reinterpret_cast<u32x2*>(y + i));
    }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main() {
    const long rows = 163840; const int D = 768; const long n = rows * D;
    float *x, *xn; bf16 *d, *y;
    CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&xn, n * 4)); CK(hipMalloc(&d, n * 2)); CK(hipMalloc(&y, n * 2));
    CK(hipMemset(x, 0, n * 4)); CK(hipMemset(d, 0, n * 2));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double bytes = n * 12.0;
#define RUN(name, ...) { for (int w = 0; w < 3; ++w) { __VA_ARGS__; } CK(hipEventRecord(e0)); for (int it = 0; it < 20; ++it) { __VA_ARGS__; } CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); \
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); printf("%-44s %.1f us  %.2f TB/s\n", name, ms / 20 * 1e3, bytes / (ms / 20 * 1e-3) / 1e12); }
    RUN("A row per wave (16 B fp32, 8 B bf16)", (rowwise<<<(unsigned)(rows / 4), 256>>>(x, d, xn, y, rows, D)));
    RUN("B flat, 8 elements per thread (all 16 B)", (flat8<<<(unsigned)(n / 8 / 256), 256>>>(x, d, xn, y, n)));
    RUN("C flat, dense 4-element accesses", (flat_dense<<<(unsigned)(n / 512 / 4), 256>>>(x, d, xn, y, n)));
    return 0;
}
