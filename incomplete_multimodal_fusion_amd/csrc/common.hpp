// Shared device helpers for the mmae HIP kernels (gfx950 / CDNA4 only: 64-wide waves, MFMA).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MMAE_F32 0
#define MMAE_BF16 1

#define MMAE_OK 0
#define MMAE_ERR_ARG (-1)      // invalid argument (null pointer, bad size / alignment, unsupported dtype or head dim)
#define MMAE_ERR_LAUNCH (-2)   // hipGetLastError() after launch was not hipSuccess

typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

#define WAVE 64

template <typename T> struct Vec8;
template <> struct Vec8<float> { typedef f32x8 type; };
template <> struct Vec8<bf16> { typedef bf16x8 type; };

__device__ __forceinline__ float to_f(float x) { return x; }
__device__ __forceinline__ float to_f(bf16 x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f(float x);
template <> __device__ __forceinline__ float from_f<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16 from_f<bf16>(float x) { return (bf16)x; }

// 16x16 output tile, K = 32 contraction step.  Every lane supplies 8 k-contiguous elements
// (k = 8*(lane>>4) + j) of row/col (lane & 15) for A and B.
//   bf16: one v_mfma_f32_16x16x32_bf16.
//   f32 : eight v_mfma_f32_16x16x4_f32; instruction j contracts the k-set {8g + j : g = 0..3}.  The sum over k is
//         order independent as long as A and B use the same k-slot assignment, so the same fragments serve both.
// C/D layout (dtype independent): col = lane & 15, row = 4*(lane>>4) + reg.
__device__ __forceinline__ f32x4 mma16(const bf16x8& a, const bf16x8& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mma16(const f32x8& a, const f32x8& b, f32x4 c) {
#pragma unroll
    for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], c, 0, 0, 0);
    return c;
}

template <typename T> __device__ __forceinline__ typename Vec8<T>::type zero8() {
    typename Vec8<T>::type v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = from_f<T>(0.f);
    return v;
}

// 8 contiguous elements from global memory (16 B for bf16, 32 B for f32); p must be 16-B aligned.
template <typename T> __device__ __forceinline__ typename Vec8<T>::type ld8(const T* p) {
    return *reinterpret_cast<const typename Vec8<T>::type*>(p);
}
template <typename T> __device__ __forceinline__ void st8(T* p, const typename Vec8<T>::type& v) {
    *reinterpret_cast<typename Vec8<T>::type*>(p) = v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// exact (erf) GELU and its derivative
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
    const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752f));
    const float pdf = 0.39894228040143268f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// Launch + check.  hipGetLastError() reports (and clears) the last error of ANY earlier runtime call of this thread -- e.g. a
// benign hipErrorNotReady the host framework left behind after an event / stream query -- so the slot is cleared right before
// the launch: MMAE_CHECK_LAUNCH() then sees this launch's own status only.  The code of a failed launch is kept per thread for
// mmae_last_hip_error() (error reporting of the binding).
extern thread_local int mmae_tls_last_hip_error;
#define MMAE_LAUNCH(...)                                     \
    do {                                                     \
        (void)hipGetLastError();                             \
        hipLaunchKernelGGL(__VA_ARGS__);                     \
    } while (0)
#define MMAE_CHECK_LAUNCH()                                  \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        if (e__ != hipSuccess) { mmae_tls_last_hip_error = (int)e__; return MMAE_ERR_LAUNCH; } \
    } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
