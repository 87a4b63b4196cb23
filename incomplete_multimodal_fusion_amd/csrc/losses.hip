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
// One block.  n = 2B rows; out = cat(normalize(o1), normalize(o2)).  LDS holds the normalised rows' Gram matrix
// E[i][j] = exp(<out_i, out_j> / T).  For row i:  neg = {E[i][j] : j mod B != i mod B},  pos_i = E[i][(i+B) mod n],
//   imp = neg^beta,  rw = sum(imp*neg) / mean(imp),  Ng = max((-tau*N*pos + rw) / (1 - tau), N*e^(-1/T)),  N = n - 2,
//   loss = mean_i -log(pos / (pos + Ng)).
// Backward: dL/dE analytically, then dL/dout = (dE o E)/T-weighted sums of rows (E is symmetric: both roles of a row
// are accumulated), then through the row normalisation.
struct HnDesc {
    const float* o1; const float* o2; float* loss; const float* gloss; float* g1; float* g2;
    float* ws;     // workspace: n*D (normalised rows) + n*n (E) + n*n (dL/dS) + n (norms) floats
    int B, D; float tau_plus, beta, temperature;
};

__global__ __launch_bounds__(1024) void hardneg_kernel(HnDesc d, int backward) {
    const int n = 2 * d.B, D = d.D, tid = threadIdx.x, nt = blockDim.x;
    float* U = d.ws;                 // (n, D) normalised rows
    float* E = U + (long)n * D;      // (n, n)
    float* G = E + (long)n * n;      // (n, n) dL/dS where S = <u_i,u_j>/T
    float* nrm = G + (long)n * n;    // (n)
    float* rowl = nrm + n;           // (n) per-row loss
    const int lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
    for (int i = wave; i < n; i += nw) {
        const float* src = i < d.B ? d.o1 + (long)i * D : d.o2 + (long)(i - d.B) * D;
        float a = 0.f;
        for (int c = lane; c < D; c += 64) a += src[c] * src[c];
        const float nm = fmaxf(sqrtf(wave_sum(a)), 1e-12f);
        for (int c = lane; c < D; c += 64) U[(long)i * D + c] = src[c] / nm;
        if (lane == 0) nrm[i] = nm;
    }
    __syncthreads();
    for (int ij = wave; ij < n * n; ij += nw) {
        const int i = ij / n, j = ij % n;
        float a = 0.f;
        for (int c = lane; c < D; c += 64) a += U[(long)i * D + c] * U[(long)j * D + c];
        a = wave_sum(a);
        if (lane == 0) E[ij] = expf(a / d.temperature);
    }
    __syncthreads();
    const float Nn = (float)(n - 2);
    const float clampv = Nn * expf(-1.f / d.temperature);
    for (int i = wave; i < n; i += nw) {
        const int ib = i % d.B;
        float simp = 0.f, sin_ = 0.f;
        for (int j = lane; j < n; j += 64) {
            if (j % d.B != ib) { const float e = E[i * n + j]; const float imp = powf(e, d.beta); simp += imp; sin_ += imp * e; }
        }
        simp = wave_sum(simp); sin_ = wave_sum(sin_);
        const float pos = E[i * n + (i + d.B) % n];
        const float meanimp = simp / Nn;
        const float rw = sin_ / meanimp;
        const float raw = (-d.tau_plus * Nn * pos + rw) / (1.f - d.tau_plus);
        const bool clamped = raw < clampv;
        const float Ng = clamped ? clampv : raw;
        if (lane == 0) rowl[i] = -logf(pos / (pos + Ng));
        if (backward) {
            // loss_i = log(pos + Ng) - log(pos)
            const float gi = d.gloss[0] / (float)n;
            const float dNg = clamped ? 0.f : gi / (pos + Ng);
            const float dpos = gi * (1.f / (pos + Ng) - 1.f / pos) + dNg * (-d.tau_plus * Nn) / (1.f - d.tau_plus);
            const float drw = dNg / (1.f - d.tau_plus);
            // rw = Nn * sin_ / simp ; imp = e^beta
            for (int j = lane; j < n; j += 64) {
                float ge = 0.f;
                if (j % d.B != ib) {
                    const float e = E[i * n + j];
                    const float imp = powf(e, d.beta);
                    const float dimp_de = d.beta * imp / e;
                    // d(sin_)/de = dimp_de*e + imp ; d(simp)/de = dimp_de
                    ge = drw * Nn * ((dimp_de * e + imp) / simp - sin_ * dimp_de / (simp * simp));
                }
                if (j == (i + d.B) % n) ge += dpos;
                G[i * n + j] = ge * E[i * n + j] / d.temperature;      // dL/d<u_i,u_j> contribution of row i
            }
        }
    }
    __syncthreads();
    if (!backward) {
        if (tid == 0) { float a = 0.f; for (int i = 0; i < n; ++i) a += rowl[i]; d.loss[0] = a / (float)n; }
        return;
    }
    // du_i = sum_j (G[i][j] + G[j][i]) u_j ; then through normalisation: ds = (du - u <u,du>) / |s|
    for (int i = wave; i < n; i += nw) {
        float dot = 0.f;
        float* dst = i < d.B ? d.g1 + (long)i * D : d.g2 + (long)(i - d.B) * D;
        for (int c = lane; c < D; c += 64) {
            float a = 0.f;
            for (int j = 0; j < n; ++j) a += (G[i * n + j] + G[j * n + i]) * U[(long)j * D + c];
            dst[c] = a;                       // temporarily du
            dot += a * U[(long)i * D + c];
        }
        dot = wave_sum(dot);
        for (int c = lane; c < D; c += 64) dst[c] = (dst[c] - U[(long)i * D + c] * dot) / nrm[i];
    }
}

extern "C" long mmae_hardneg_ws_floats(int B, int D) { const long n = 2L * B; return n * D + 2 * n * n + 2 * n; }

extern "C" int mmae_hardneg_loss_fwd(int B, int D, const float* out_1, const float* out_2, float tau_plus, float beta,
                                     float temperature, float* ws, float* loss, void* stream) {
    if (B <= 1 || D <= 0 || !out_1 || !out_2 || !ws || !loss) return MMAE_ERR_ARG;
    HnDesc d{out_1, out_2, loss, nullptr, nullptr, nullptr, ws, B, D, tau_plus, beta, temperature};
    MMAE_LAUNCH(hardneg_kernel, dim3(1), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream), d, 0);
    MMAE_CHECK_LAUNCH();
    return MMAE_OK;
}
extern "C" int mmae_hardneg_loss_bwd(int B, int D, const float* out_1, const float* out_2, float tau_plus, float beta,
                                     float temperature, float* ws, const float* gloss, float* g1, float* g2, void* stream) {
    if (B <= 1 || D <= 0 || !out_1 || !out_2 || !ws || !gloss || !g1 || !g2) return MMAE_ERR_ARG;
    HnDesc d{out_1, out_2, nullptr, gloss, g1, g2, ws, B, D, tau_plus, beta, temperature};
    MMAE_LAUNCH(hardneg_kernel, dim3(1), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream), d, 1);
    MMAE_CHECK_LAUNCH();
    return MMAE_OK;
}
