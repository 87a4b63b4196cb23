// Input staging: raw sensor tiles -> the normalised fp32 tiles the model consumes (gfx950, HBM-bound).
//
// Reference arithmetic replaced (pretraining/utils/multimodal_dfc2023.py; CPU numpy + cv2 per sample there):
//   load_sar :127-139   10*log10(x + 1e-7) -> clip [-25, 0] -> nan_to_num -> resize (INTER_AREA) -> (x - mean) / std  (:36-42)
//   load_rgb :114-124   nan_to_num -> resize -> per-channel (x - mean[c]) / std[c]                                   (:25-31)
//   load_dsm :99-111    nan_to_num -> resize -> per-TILE z-score (x - mean(x)) / sqrt(var(x) + 1e-6)
// cv2.INTER_AREA with an integer shrink factor f is the mean over f x f blocks; factor 1 is the identity.  For uint8
// input cv2 returns uint8, i.e. the block mean is rounded (half to even) before normalisation -- reproduced here.
#include "common.hpp"
#include "mmae_hip.h"

struct StageDesc {
    const void* raw;    // (B, C, H*f, W*f)
    float* out;         // (B, C, H, W)
    int B, C, H, W, f, kind;
    float mean[4], stdv[4];
};

template <typename TI> __device__ __forceinline__ float load_raw(const TI* p, long i);
template <> __device__ __forceinline__ float load_raw<float>(const float* p, long i) { return p[i]; }
template <> __device__ __forceinline__ float load_raw<unsigned char>(const unsigned char* p, long i) { return (float)p[i]; }

__device__ __forceinline__ float nan_to_num_f(float v) {
    if (v != v) return 0.f;
    return fminf(fmaxf(v, -3.4028234663852886e38f), 3.4028234663852886e38f);
}

// block mean of the pre-transformed raw values for output pixel (y, x) of plane `p`
template <typename TI, int KIND>
__device__ __forceinline__ float pooled(const TI* __restrict__ plane, int y, int x, int f, int Wraw) {
    float acc = 0.f;
    for (int dy = 0; dy < f; ++dy)
        for (int dx = 0; dx < f; ++dx) {
            float v = load_raw<TI>(plane, (long)(y * f + dy) * Wraw + x * f + dx);
            if (KIND == MMAE_STAGE_SAR_DB) {
                const float r = v + 1e-7f;
                // log10 of NaN / a negative is NaN -> clip keeps NaN -> nan_to_num gives 0; log10(0) = -inf clips to -25
                v = (r != r || r < 0.f) ? 0.f : fminf(fmaxf(10.f * log10f(r), -25.f), 0.f);
            } else {
                v = nan_to_num_f(v);
            }
            acc += v;
        }
    float m = f == 1 ? acc : acc / (float)(f * f);
    if (sizeof(TI) == 1 && f > 1) m = rintf(m);           // cv2 keeps uint8: saturate_cast<uchar>(cvRound(mean))
    return m;
}

template <typename TI, int KIND>
__global__ __launch_bounds__(256) void stage_affine_kernel(StageDesc d) {
    const long n = (long)d.B * d.C * d.H * d.W;
    const long hw = (long)d.H * d.W;
    const int Wraw = d.W * d.f;
    const long raw_plane = (long)d.H * d.f * Wraw;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long plane = i / hw;
        const int c = (int)(plane % d.C);
        const int rem = (int)(i - plane * hw), y = rem / d.W, x = rem - y * d.W;
        const float v = pooled<TI, KIND>(reinterpret_cast<const TI*>(d.raw) + plane * raw_plane, y, x, d.f, Wraw);
        d.out[i] = (v - d.mean[c]) / d.stdv[c];
    }
}

__device__ __forceinline__ float block_sum_1024(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += red[i];
    return t;
}

// per-tile z-score: one 1024-thread block per (b, c) plane; two-pass mean / variance (population), fixed order
template <typename TI>
__global__ __launch_bounds__(1024) void stage_zscore_kernel(StageDesc d) {
    __shared__ float red[16];
    const long plane = blockIdx.x;
    const int hw = d.H * d.W, Wraw = d.W * d.f;
    const TI* raw = reinterpret_cast<const TI*>(d.raw) + plane * (long)d.H * d.f * Wraw;
    float* out = d.out + plane * hw;
    float s = 0.f;
    for (int i = threadIdx.x; i < hw; i += 1024) {
        const int y = i / d.W, x = i - y * d.W;
        const float v = pooled<TI, MMAE_STAGE_ZSCORE>(raw, y, x, d.f, Wraw);
        out[i] = v;
        s += v;
    }
    const float mean = block_sum_1024(s, red) / (float)hw;
    float q = 0.f;
    for (int i = threadIdx.x; i < hw; i += 1024) { const float e = out[i] - mean; q += e * e; }
    const float var = block_sum_1024(q, red) / (float)hw;
    const float sd = sqrtf(var + 1e-6f);
    for (int i = threadIdx.x; i < hw; i += 1024) out[i] = (out[i] - mean) / sd;
}

extern "C" int mmae_stage_tiles(int kind, int in_dtype, int B, int C, int H, int W, int factor, const void* raw, float* out,
                                const float* mean, const float* stdv, void* stream) {
    if (kind < 0 || kind > 2 || (in_dtype != MMAE_RAW_F32 && in_dtype != MMAE_RAW_U8) || B < 0 || C < 1 || C > 4 || H < 1 ||
        W < 1 || factor < 1 || factor > 16 || !raw || !out)
        return MMAE_ERR_ARG;
    if (kind != MMAE_STAGE_ZSCORE && (!mean || !stdv)) return MMAE_ERR_ARG;
    if (B == 0) return MMAE_OK;
    StageDesc d{};
    d.raw = raw; d.out = out; d.B = B; d.C = C; d.H = H; d.W = W; d.f = factor; d.kind = kind;
    for (int c = 0; c < C && kind != MMAE_STAGE_ZSCORE; ++c) {
        if (!(stdv[c] > 0.f)) return MMAE_ERR_ARG;
        d.mean[c] = mean[c]; d.stdv[c] = stdv[c];
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const long n = (long)B * C * H * W;
    long nb = (n + 255) / 256; if (nb > 8192) nb = 8192;
    const bool u8 = in_dtype == MMAE_RAW_U8;
    if (kind == MMAE_STAGE_ZSCORE) {
        if (u8) MMAE_LAUNCH(stage_zscore_kernel<unsigned char>, dim3(B * C), dim3(1024), 0, st, d);
        else MMAE_LAUNCH(stage_zscore_kernel<float>, dim3(B * C), dim3(1024), 0, st, d);
    } else if (kind == MMAE_STAGE_SAR_DB) {
        if (u8) MMAE_LAUNCH((stage_affine_kernel<unsigned char, MMAE_STAGE_SAR_DB>), dim3((unsigned)nb), dim3(256), 0, st, d);
        else MMAE_LAUNCH((stage_affine_kernel<float, MMAE_STAGE_SAR_DB>), dim3((unsigned)nb), dim3(256), 0, st, d);
    } else {
        if (u8) MMAE_LAUNCH((stage_affine_kernel<unsigned char, MMAE_STAGE_AFFINE>), dim3((unsigned)nb), dim3(256), 0, st, d);
        else MMAE_LAUNCH((stage_affine_kernel<float, MMAE_STAGE_AFFINE>), dim3((unsigned)nb), dim3(256), 0, st, d);
    }
    MMAE_CHECK_LAUNCH();
    return MMAE_OK;
}
