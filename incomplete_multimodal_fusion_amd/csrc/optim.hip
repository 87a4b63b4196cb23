// Fused AdamW over flat parameter buffers (gfx950, HBM-bound: 30 B per element, one pass).
//
// Replaces the optimizer tail of the reference step (pretraining/utils/optim_factory.py:136-179 builds torch AdamW over
// all parameters; pretraining/utils/native_scaler.py:20-40 steps it, :49-62 computes the gradient norm per parameter with a
// stack of torch.norm calls).  One launch updates the fp32 master weights, both moments and the bf16 shadow copy the
// next forward's GEMMs read -- so there is no per-parameter cast kernel and no separate multi-tensor optimizer launch.
// Update rule = torch.optim.AdamW (decoupled weight decay, bias-corrected):
//   p *= 1 - lr*wd;  m = b1*m + (1-b1)*g;  v = b2*v + (1-b2)*g*g;  p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
#include "common.hpp"
#include "mmae_hip.h"

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, bf16* __restrict__ shadow, long n, float lr,
                                                    float b1, float b2, float eps, float wd, float inv_bc1,
                                                    float inv_sqrt_bc2, float grad_scale) {
    const long i0 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    const long stride = (long)gridDim.x * 256 * 4;
    for (long i = i0; i < n; i += stride) {
        f32x4 P = *reinterpret_cast<const f32x4*>(p + i);
        f32x4 G = *reinterpret_cast<const f32x4*>(g + i) * grad_scale;
        f32x4 M = *reinterpret_cast<const f32x4*>(m + i);
        f32x4 V = *reinterpret_cast<const f32x4*>(v + i);
        P *= (1.f - lr * wd);
        M = M * b1 + G * (1.f - b1);
        V = V * b2 + G * G * (1.f - b2);
#pragma unroll
        for (int j = 0; j < 4; ++j) P[j] -= (lr * inv_bc1) * M[j] / (sqrtf(V[j]) * inv_sqrt_bc2 + eps);
        *reinterpret_cast<f32x4*>(p + i) = P;
        *reinterpret_cast<f32x4*>(m + i) = M;
        *reinterpret_cast<f32x4*>(v + i) = V;
        if (shadow) {
            bf16x4 s; s[0] = (bf16)P[0]; s[1] = (bf16)P[1]; s[2] = (bf16)P[2]; s[3] = (bf16)P[3];
            *reinterpret_cast<bf16x4*>(shadow + i) = s;
        }
    }
}

// fp32 -> bf16 copy of a flat buffer (initial shadow / after loading a checkpoint)
__global__ __launch_bounds__(256) void shadow_kernel(const float* __restrict__ p, bf16* __restrict__ shadow, long n) {
    const long stride = (long)gridDim.x * 256 * 4;
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
        const f32x4 P = *reinterpret_cast<const f32x4*>(p + i);
        bf16x4 s; s[0] = (bf16)P[0]; s[1] = (bf16)P[1]; s[2] = (bf16)P[2]; s[3] = (bf16)P[3];
        *reinterpret_cast<bf16x4*>(shadow + i) = s;
    }
}

// sum of squares of a flat fp32 buffer -> out[0] (+= when accumulate): two-stage, deterministic
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, long n, float* __restrict__ part) {
    __shared__ float red[4];
    float a = 0.f;
    const long stride = (long)gridDim.x * 256 * 4;
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
        const f32x4 G = *reinterpret_cast<const f32x4*>(g + i);
        a += G[0] * G[0] + G[1] * G[1] + G[2] * G[2] + G[3] * G[3];
    }
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void sumsq_finish_kernel(const float* __restrict__ part, int nb, float* out) {
    __shared__ float red[256];
    float a = 0.f;
    for (int i = threadIdx.x; i < nb; i += 256) a += part[i];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) out[0] = sqrtf(red[0]);
}

static int grid_for(long n) { long b = (n / 4 + 255) / 256; if (b > 2048) b = 2048; if (b < 1) b = 1; return (int)b; }

extern "C" int mmae_adamw_step(long n, float* p, const float* g, float* m, float* v, void* shadow_bf16, float lr,
                               float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                               void* stream) {
    if (n < 0 || (n % 4) || !p || !g || !m || !v || step < 1) return MMAE_ERR_ARG;
    if (n == 0) return MMAE_OK;
    const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p, g, m, v,
                       reinterpret_cast<bf16*>(shadow_bf16), n, lr, beta1, beta2, eps, weight_decay, 1.f / bc1,
                       1.f / sqrtf(bc2), grad_scale);
    MMAE_CHECK_LAUNCH();
    return MMAE_OK;
}

extern "C" int mmae_shadow_bf16(long n, const float* p, void* shadow_bf16, void* stream) {
    if (n < 0 || (n % 4) || !p || !shadow_bf16) return MMAE_ERR_ARG;
    if (n == 0) return MMAE_OK;
    hipLaunchKernelGGL(shadow_kernel, dim3(grid_for(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p,
                       reinterpret_cast<bf16*>(shadow_bf16), n);
    MMAE_CHECK_LAUNCH();
    return MMAE_OK;
}

extern "C" int mmae_grad_norm(long n, const float* g, float* partial_ws_2048, float* out_norm, void* stream) {
    if (n < 0 || (n % 4) || !g || !partial_ws_2048 || !out_norm) return MMAE_ERR_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int nb = grid_for(n);
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nb), dim3(256), 0, st, g, n, partial_ws_2048);
    MMAE_CHECK_LAUNCH();
    hipLaunchKernelGGL(sumsq_finish_kernel, dim3(1), dim3(256), 0, st, partial_ws_2048, nb, out_norm);
    MMAE_CHECK_LAUNCH();
    return MMAE_OK;
}
