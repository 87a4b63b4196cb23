// Contrastive loss heads (gfx950).  Tiny (B x D) problems: one wave per sample row, fp32 throughout.
//   dino_loss_func   (pretraining/multimae/criterion.py:328-335): L2-normalise both, log_softmax(student/0.1),
//                    softmax(teacher/0.04).detach(), mean_b sum_d -t*log s.  Gradient flows to the student only.
//   HardNegtive_loss (criterion.py:233-268, estimator 'hard'): debiased hard-negative NT-Xent over the LOCAL batch.
#include "common.hpp"
#include "mmae_hip.h"

#define LNC 4   // up to 1024 columns: 4 chunks of (64 lanes x 4)

struct DinoDesc {
    const float* s; const float* t; float* row_loss; const float* gloss; float* gs;
    int B, D; float inv_ts, inv_tt;
};

__device__ __forceinline__ void dino_row(const DinoDesc& d, int row, int lane, f32x4 (&sn)[LNC], f32x4 (&ps)[LNC],
                                         f32x4 (&pt)[LNC], float& norm_s, float& loss) {
    const int D = d.D;
    f32x4 tn[LNC];
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int c = 0; c < LNC; ++c) {
        const int col = 4 * (lane + 64 * c);
        sn[c] = f32x4{0.f, 0.f, 0.f, 0.f}; tn[c] = sn[c];
        if (col < D) {
            sn[c] = *reinterpret_cast<const f32x4*>(d.s + (long)row * D + col);
            tn[c] = *reinterpret_cast<const f32x4*>(d.t + (long)row * D + col);
#pragma unroll
            for (int j = 0; j < 4; ++j) { a += sn[c][j] * sn[c][j]; b += tn[c][j] * tn[c][j]; }
        }
    }
    norm_s = fmaxf(sqrtf(wave_sum(a)), 1e-12f);
    const float norm_t = fmaxf(sqrtf(wave_sum(b)), 1e-12f);
    float ms = -INFINITY, mt = -INFINITY;
#pragma unroll
    for (int c = 0; c < LNC; ++c) {
        const int col = 4 * (lane + 64 * c);
        if (col < D) {
            sn[c] = sn[c] / norm_s; tn[c] = tn[c] / norm_t;
#pragma unroll
            for (int j = 0; j < 4; ++j) { ms = fmaxf(ms, sn[c][j] * d.inv_ts); mt = fmaxf(mt, tn[c][j] * d.inv_tt); }
        }
    }
    ms = wave_max(ms); mt = wave_max(mt);
    float es = 0.f, et = 0.f;
#pragma unroll
    for (int c = 0; c < LNC; ++c) {
        const int col = 4 * (lane + 64 * c);
        ps[c] = f32x4{0.f, 0.f, 0.f, 0.f}; pt[c] = ps[c];
        if (col < D) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ps[c][j] = expf(sn[c][j] * d.inv_ts - ms); pt[c][j] = expf(tn[c][j] * d.inv_tt - mt);
                es += ps[c][j]; et += pt[c][j];
            }
        }
    }
    es = wave_sum(es); et = wave_sum(et);
    const float lse_s = ms + logf(es);
    float l = 0.f;
#pragma unroll
    for (int c = 0; c < LNC; ++c) {
        const int col = 4 * (lane + 64 * c);
        if (col < D) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ps[c][j] /= es; pt[c][j] /= et;
                l -= pt[c][j] * (sn[c][j] * d.inv_ts - lse_s);
            }
        }
    }
    loss = wave_sum(l);
}

__global__ __launch_bounds__(256) void dino_fwd_kernel(DinoDesc d) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= d.B) return;
    f32x4 sn[LNC], ps[LNC], pt[LNC];
    float ns, loss;
    dino_row(d, row, lane, sn, ps, pt, ns, loss);
    if (lane == 0) d.row_loss[row] = loss;
}

__global__ __launch_bounds__(256) void mean_rows_kernel(const float* v, int n, float* out) {
    __shared__ float s[256];
    float a = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) a += v[i];
    s[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) out[0] = s[0] / (float)n;
}

__global__ __launch_bounds__(256) void dino_bwd_kernel(DinoDesc d) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= d.B) return;
    f32x4 sn[LNC], ps[LNC], pt[LNC];
    float ns, loss;
    dino_row(d, row, lane, sn, ps, pt, ns, loss);
    const float gl = d.gloss[0] / (float)d.B;
    // dL/dz = gl * (softmax(z) - pt);  z = sn / ts;  sn = s / |s|
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < LNC; ++c) {
        const int col = 4 * (lane + 64 * c);
        if (col < d.D) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ps[c][j] = gl * (ps[c][j] - pt[c][j]) * d.inv_ts;      // grad wrt sn
                dot += ps[c][j] * sn[c][j];
            }
        }
    }
    dot = wave_sum(dot);
#pragma unroll
    for (int c = 0; c < LNC; ++c) {
        const int col = 4 * (lane + 64 * c);
        if (col < d.D) {
            f32x4 g;
#pragma unroll
            for (int j = 0; j < 4; ++j) g[j] = (ps[c][j] - sn[c][j] * dot) / ns;
            *reinterpret_cast<f32x4*>(d.gs + (long)row * d.D + col) = g;
        }
    }
}

extern "C" int mmae_dino_loss_fwd(int B, int D, const float* student, const float* teacher, float student_temp,
                                  float teacher_temp, float* row_loss_ws, float* loss, void* stream) {
    if (B <= 0 || D <= 0 || (D % 4) || D > 1024 || !student || !teacher || !row_loss_ws || !loss) return MMAE_ERR_ARG;
    DinoDesc d{student, teacher, row_loss_ws, nullptr, nullptr, B, D, 1.f / student_temp, 1.f / teacher_temp};
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    MMAE_LAUNCH(dino_fwd_kernel, dim3(cdiv(B, 4)), dim3(256), 0, st, d);
    MMAE_CHECK_LAUNCH();
    MMAE_LAUNCH(mean_rows_kernel, dim3(1), dim3(256), 0, st, row_loss_ws, B, loss);
    MMAE_CHECK_LAUNCH();
    return MMAE_OK;
}

extern "C" int mmae_dino_loss_bwd(int B, int D, const float* student, const float* teacher, float student_temp,
                                  float teacher_temp, const float* gloss, float* gstudent, void* stream) {
    if (B <= 0 || D <= 0 || (D % 4) || D > 1024 || !student || !teacher || !gloss || !gstudent) return MMAE_ERR_ARG;
    DinoDesc d{student, teacher, nullptr, gloss, gstudent, B, D, 1.f / student_temp, 1.f / teacher_temp};
    MMAE_LAUNCH(dino_bwd_kernel, dim3(cdiv(B, 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), d);
    MMAE_CHECK_LAUNCH();
    return MMAE_OK;
}

// ------------------------------------------------------------------------------------------ hard-negative NT-Xent
// n = 2B rows; out = cat(normalize(o1), normalize(o2)); E[i][j] = exp(<out_i, out_j> / T).  For row i:
//   neg = {E[i][j] : j mod B != i mod B},  pos_i = E[i][(i+B) mod n],
//   imp = neg^beta,  rw = sum(imp*neg) / mean(imp),  Ng = max((-tau*N*pos + rw) / (1 - tau), N*e^(-1/T)),  N = n - 2,
//   loss = mean_i -log(pos / (pos + Ng)).
// Backward: dL/dE analytically (G = dL/dS, S = <u_i,u_j>/T), then dL/du_i = sum_j (G[i][j] + G[j][i]) u_j (E is symmetric: both
// roles of a row), then through the row normalisation.
// A chain of small kernels over the whole chip, fixed summation orders (no atomics): rows -> Gram tiles -> per-row statistics
// (+ G) -> loss mean | (G + G^T) U tiles -> normalisation backward.  The first version ran the whole head in ONE workgroup:
// 3.2 ms per call at B = 64, D = 1024 (n^2 = 16 k dot products of 1024 on one CU) -- 27 % of BASELINE config 5's step.
struct HnDesc {
    const float* o1; const float* o2; float* loss; const float* gloss; float* g1; float* g2;
    float* ws;     // workspace: n*D (normalised rows) + n*n (E) + n*n (dL/dS) + n (norms) + n (row losses) floats
    int B, D; float tau_plus, beta, temperature;
};
__device__ __forceinline__ float* hn_U(const HnDesc& d) { return d.ws; }
__device__ __forceinline__ float* hn_E(const HnDesc& d) { return d.ws + 2L * d.B * d.D; }
__device__ __forceinline__ float* hn_G(const HnDesc& d) { return hn_E(d) + 4L * d.B * d.B; }
__device__ __forceinline__ float* hn_nrm(const HnDesc& d) { return hn_G(d) + 4L * d.B * d.B; }
__device__ __forceinline__ float* hn_rowl(const HnDesc& d) { return hn_nrm(d) + 2L * d.B; }

// one wave per row: u = s / max(|s|, 1e-12)
__global__ __launch_bounds__(256) void hn_normalize_kernel(HnDesc d) {
    const int n = 2 * d.B, D = d.D, lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const float* src = i < d.B ? d.o1 + (long)i * D : d.o2 + (long)(i - d.B) * D;
    float a = 0.f;
    for (int c = lane; c < D; c += 64) a += src[c] * src[c];
    const float nm = fmaxf(sqrtf(wave_sum(a)), 1e-12f);
    float* U = hn_U(d);
    for (int c = lane; c < D; c += 64) U[(long)i * D + c] = src[c] / nm;
    if (lane == 0) hn_nrm(d)[i] = nm;
}

// E tile of 32 x 32 per workgroup: 256 threads, thread (ty, tx) owns E[i0 + ty + 8 r][j0 + tx], r = 0..3; D walked in chunks of 32
// through LDS (rows padded to 33 floats: conflict-free for both operands)
__global__ __launch_bounds__(256) void hn_gram_kernel(HnDesc d) {
    __shared__ float A[32][33], Bt[32][33];
    const int n = 2 * d.B, D = d.D, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
    const float* U = hn_U(d);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c0 = 0; c0 < D; c0 += 32) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = ty + 8 * r, c = c0 + tx;
            A[row][tx] = (i0 + row < n && c < D) ? U[(long)(i0 + row) * D + c] : 0.f;
            Bt[row][tx] = (j0 + row < n && c < D) ? U[(long)(j0 + row) * D + c] : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int k = 0; k < 32; ++k) {
            const float b = Bt[tx][k];
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = fmaf(A[ty + 8 * r][k], b, acc[r]);
        }
        __syncthreads();
    }
    float* E = hn_E(d);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = i0 + ty + 8 * r, j = j0 + tx;
        if (i < n && j < n) E[(long)i * n + j] = expf(acc[r] / d.temperature);
    }
}

// one wave per row: the row's statistics, its loss and (backward) its row of G
__global__ __launch_bounds__(256) void hn_rows_kernel(HnDesc d, int backward) {
    const int n = 2 * d.B, lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const float* E = hn_E(d) + (long)i * n;
    const float Nn = (float)(n - 2);
    const float clampv = Nn * expf(-1.f / d.temperature);
    const int ib = i % d.B;
    float simp = 0.f, sin_ = 0.f;
    for (int j = lane; j < n; j += 64) {
        if (j % d.B != ib) { const float e = E[j]; const float imp = powf(e, d.beta); simp += imp; sin_ += imp * e; }
    }
    simp = wave_sum(simp); sin_ = wave_sum(sin_);
    const float pos = E[(i + d.B) % n];
    const float meanimp = simp / Nn;
    const float rw = sin_ / meanimp;
    const float raw = (-d.tau_plus * Nn * pos + rw) / (1.f - d.tau_plus);
    const bool clamped = raw < clampv;
    const float Ng = clamped ? clampv : raw;
    if (lane == 0) hn_rowl(d)[i] = -logf(pos / (pos + Ng));
    if (!backward) return;
    // loss_i = log(pos + Ng) - log(pos)
    const float gi = d.gloss[0] / (float)n;
    const float dNg = clamped ? 0.f : gi / (pos + Ng);
    const float dpos = gi * (1.f / (pos + Ng) - 1.f / pos) + dNg * (-d.tau_plus * Nn) / (1.f - d.tau_plus);
    const float drw = dNg / (1.f - d.tau_plus);
    float* G = hn_G(d) + (long)i * n;
    // rw = Nn * sin_ / simp ; imp = e^beta
    for (int j = lane; j < n; j += 64) {
        float ge = 0.f;
        const float e = E[j];
        if (j % d.B != ib) {
            const float imp = powf(e, d.beta);
            const float dimp_de = d.beta * imp / e;
            // d(sin_)/de = dimp_de*e + imp ; d(simp)/de = dimp_de
            ge = drw * Nn * ((dimp_de * e + imp) / simp - sin_ * dimp_de / (simp * simp));
        }
        if (j == (i + d.B) % n) ge += dpos;
        G[j] = ge * e / d.temperature;                 // dL/d<u_i,u_j> contribution of row i
    }
}

__global__ __launch_bounds__(64) void hn_loss_kernel(HnDesc d) {
    const int n = 2 * d.B;
    if (threadIdx.x == 0) { float a = 0.f; for (int i = 0; i < n; ++i) a += hn_rowl(d)[i]; d.loss[0] = a / (float)n; }   // fixed order
}

// du tile: 16 rows x 256 columns per workgroup (thread = one column): du_i[c] = sum_j (G[i][j] + G[j][i]) u_j[c], j in ascending
// order; the 16 x n block of G + G^T is staged in LDS in chunks of 128 columns of j
__global__ __launch_bounds__(256) void hn_du_kernel(HnDesc d) {
    __shared__ float H[16][129];
    const int n = 2 * d.B, D = d.D, tid = threadIdx.x;
    const int i0 = blockIdx.y * 16, c = blockIdx.x * 256 + tid;
    const float* G = hn_G(d);
    const float* U = hn_U(d);
    float acc[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int jb = 0; jb < n; jb += 128) {
        for (int e = tid; e < 16 * 128; e += 256) {
            const int r = e >> 7, jj = e & 127, i = i0 + r, j = jb + jj;
            H[r][jj] = (i < n && j < n) ? G[(long)i * n + j] : 0.f;
        }
        __syncthreads();
        for (int e = tid; e < 16 * 128; e += 256) {                       // + G^T: 16 consecutive floats of row j
            const int jj = e >> 4, r = e & 15, i = i0 + r, j = jb + jj;
            if (i < n && j < n) H[r][jj] += G[(long)j * n + i];
        }
        __syncthreads();
        const int jn = min(128, n - jb);
        if (c < D) {
            for (int jj = 0; jj < jn; ++jj) {
                const float u = U[(long)(jb + jj) * D + c];
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = fmaf(H[r][jj], u, acc[r]);
            }
        }
        __syncthreads();
    }
    if (c < D) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = i0 + r;
            if (i < n) (i < d.B ? d.g1 + (long)i * D : d.g2 + (long)(i - d.B) * D)[c] = acc[r];      // du, finished below
        }
    }
}

// one wave per row, through the normalisation: ds = (du - u <u, du>) / |s|
__global__ __launch_bounds__(256) void hn_finish_kernel(HnDesc d) {
    const int n = 2 * d.B, D = d.D, lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    float* dst = i < d.B ? d.g1 + (long)i * D : d.g2 + (long)(i - d.B) * D;
    const float* u = hn_U(d) + (long)i * D;
    float dot = 0.f;
    for (int c = lane; c < D; c += 64) dot += dst[c] * u[c];
    dot = wave_sum(dot);
    const float nm = hn_nrm(d)[i];
    for (int c = lane; c < D; c += 64) dst[c] = (dst[c] - u[c] * dot) / nm;
}

static int hardneg_launch(const HnDesc& d, int backward, hipStream_t st) {
    const int n = 2 * d.B;
    MMAE_LAUNCH(hn_normalize_kernel, dim3(cdiv(n, 4)), dim3(256), 0, st, d);
    MMAE_CHECK_LAUNCH();
    MMAE_LAUNCH(hn_gram_kernel, dim3(cdiv(n, 32), cdiv(n, 32)), dim3(256), 0, st, d);
    MMAE_CHECK_LAUNCH();
    MMAE_LAUNCH(hn_rows_kernel, dim3(cdiv(n, 4)), dim3(256), 0, st, d, backward);
    MMAE_CHECK_LAUNCH();
    if (!backward) {
        MMAE_LAUNCH(hn_loss_kernel, dim3(1), dim3(64), 0, st, d);
        MMAE_CHECK_LAUNCH();
        return MMAE_OK;
    }
    MMAE_LAUNCH(hn_du_kernel, dim3(cdiv(d.D, 256), cdiv(n, 16)), dim3(256), 0, st, d);
    MMAE_CHECK_LAUNCH();
    MMAE_LAUNCH(hn_finish_kernel, dim3(cdiv(n, 4)), dim3(256), 0, st, d);
    MMAE_CHECK_LAUNCH();
    return MMAE_OK;
}

extern "C" long mmae_hardneg_ws_floats(int B, int D) { const long n = 2L * B; return n * D + 2 * n * n + 2 * n; }

extern "C" int mmae_hardneg_loss_fwd(int B, int D, const float* out_1, const float* out_2, float tau_plus, float beta,
                                     float temperature, float* ws, float* loss, void* stream) {
    if (B <= 1 || D <= 0 || !out_1 || !out_2 || !ws || !loss) return MMAE_ERR_ARG;
    HnDesc d{out_1, out_2, loss, nullptr, nullptr, nullptr, ws, B, D, tau_plus, beta, temperature};
    return hardneg_launch(d, 0, reinterpret_cast<hipStream_t>(stream));
}
extern "C" int mmae_hardneg_loss_bwd(int B, int D, const float* out_1, const float* out_2, float tau_plus, float beta,
                                     float temperature, float* ws, const float* gloss, float* g1, float* g2, void* stream) {
    if (B <= 1 || D <= 0 || !out_1 || !out_2 || !ws || !gloss || !g1 || !g2) return MMAE_ERR_ARG;
    HnDesc d{out_1, out_2, nullptr, gloss, g1, g2, ws, B, D, tau_plus, beta, temperature};
    return hardneg_launch(d, 1, reinterpret_cast<hipStream_t>(stream));
}
