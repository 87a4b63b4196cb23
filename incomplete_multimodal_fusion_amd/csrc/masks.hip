// Mask bookkeeping on the device (integer work, must be bit-exact with the reference given the same random draws).
//
//   mmae_masks_from_draws  : MultiMAE.generate_random_masks (pretraining/multimae/multimae_crossattn.py:233-272) with
//       the Dirichlet sample and the uniform noises injected:  samples_per_task = round(p*N) (:233), per task
//       mask[pos] = argsort(noise)[pos] < quota ? 0 : 1 (:241-246), ids_shuffle = argsort(mask_all + noise_all) (:264),
//       ids_restore = argsort(ids_shuffle) (:265), ids_keep = ids_shuffle[:N] (:266), mask_all rebuilt with exactly N
//       zeros (:269-272).  argsort is realised as a stable rank (ties -> lower index first).
//   mmae_build_descriptors : token selection + type bookkeeping (:402-447, :454-462, :489-493, :530-543) as device-side
//       int32 descriptors for the packed kernels instead of nonzero()/python ints (no host sync):
//       row space of a step:  [ B*N packed modality tokens | B*P fusion tokens | P mask-embedding rows ]
//       kept tokens of a sample are packed modality by modality, ascending patch index (= nonzero order, :402-406).
// One block per mask row / sample; O(P^2) rank counting -- P <= 1024, M <= 7.
#include "common.hpp"
#include "mmae_hip.h"

struct DrawsDesc {
    const float* dirichlet; const float* noise; const float* noise_all;
    long long* mask_all; long long* ids_keep; long long* ids_restore;
    int R, M, P, N;
};

__global__ __launch_bounds__(256) void masks_from_draws_kernel(DrawsDesc d) {
    extern __shared__ float sm[];            // key[M*P]
    const int r = blockIdx.x, M = d.M, P = d.P, T = M * P, N = d.N;
    float* key = sm;
    const float* noise = d.noise + (long)r * T;
    // stage 1: per-task masks -> key = mask + noise_all
    for (int j = threadIdx.x; j < T; j += blockDim.x) {
        const int task = j / P, e = j % P;
        const float x = noise[j];
        int rank = 0;
        for (int k = 0; k < P; ++k) {
            const float y = noise[task * P + k];
            rank += (y < x || (y == x && k < e)) ? 1 : 0;
        }
        // element e has rank `rank` in the sorted order => ids_arange_shuffle[rank] = e => mask[rank] = e < quota ? 0 : 1
        const long long quota = (long long)rintf(d.dirichlet[(long)r * M + task] * (float)N);
        const float mk = (long long)e < quota ? 0.f : 1.f;
        key[task * P + rank] = mk + d.noise_all[(long)r * T + task * P + rank];
    }
    __syncthreads();
    // stage 2: global stable rank of key
    for (int j = threadIdx.x; j < T; j += blockDim.x) {
        const float x = key[j];
        int rank = 0;
        for (int k = 0; k < T; ++k) {
            const float y = key[k];
            rank += (y < x || (y == x && k < j)) ? 1 : 0;
        }
        d.ids_restore[(long)r * T + j] = rank;                 // argsort(ids_shuffle) == rank
        if (rank < N) d.ids_keep[(long)r * N + rank] = j;      // ids_shuffle[rank] = j
        d.mask_all[(long)r * T + j] = rank < N ? 0 : 1;
    }
}

extern "C" int mmae_masks_from_draws(int R, int M, int P, int N, const float* dirichlet, const float* noise,
                                     const float* noise_all, long long* mask_all, long long* ids_keep,
                                     long long* ids_restore, void* stream) {
    if (R <= 0 || M <= 0 || M > 7 || P <= 0 || P > 4096 || N < 0 || N > M * P) return MMAE_ERR_ARG;
    if (!dirichlet || !noise || !noise_all || !mask_all || !ids_keep || !ids_restore) return MMAE_ERR_ARG;
    DrawsDesc d{dirichlet, noise, noise_all, mask_all, ids_keep, ids_restore, R, M, P, N};
    MMAE_LAUNCH(masks_from_draws_kernel, dim3(R), dim3(256), (size_t)M * P * sizeof(float),
                       reinterpret_cast<hipStream_t>(stream), d);
    MMAE_CHECK_LAUNCH();
    return MMAE_OK;
}

// Descriptor buffer layout (int32), all sections contiguous in `desc`, offsets returned by mmae_descriptor_layout:
//   0 enc_start (B, M+1)   row of each encoder segment (modality segments in the token part, fusion segment)
//   1 enc_len   (B, M+1)
//   2 tok_mod   (B*N)      modality of each packed token
//   3 tok_patch (B*N)      patch index of each packed token
//   4 tok_pe    (B*N)      tok_mod * P + tok_patch  (row in the concatenated pos-emb table)
//   5 tok_fus   (B*N)      row of the fusion token at the same patch  (B*N + b*P + patch)
//   6 slot_row  (B*P, M+1) rows read by the modality attention (token row, or B*N + B*P + patch for a masked slot; fusion last)
//   7 pool_qstart (B, M+1), 8 pool_qlen (B, M+1)        : 1 pooled return token per type, row b*(M+1) + s
//   9 ctr_qstart  (B, M+1), 10 ctr_qlen (B, M+1)        : 1 contrastive query per modality, row b*M + m; fusion segment empty
//   11 ctr_kstart (B, M+1), 12 ctr_klen (B, M+1)        : keys = rows of the (B*N) gathered fusion tokens, per modality
//   13 status (4)           [0] = number of samples whose kept count != N
struct BuildDesc { const long long* mask_all; int* desc; int B, R, M, P, N; };

__host__ __device__ inline void layout(int B, int M, int P, int N, long* off) {
    const long s = (long)B * (M + 1), t = (long)B * N;
    long o = 0;
    off[0] = o; o += s; off[1] = o; o += s;
    off[2] = o; o += t; off[3] = o; o += t; off[4] = o; o += t; off[5] = o; o += t;
    off[6] = o; o += (long)B * P * (M + 1);
    for (int i = 7; i <= 12; ++i) { off[i] = o; o += s; }
    off[13] = o; o += 4;
    off[14] = o;
}

extern "C" long mmae_descriptor_layout(int B, int M, int P, int N, long* offsets15) {
    if (!offsets15 || B < 0 || M < 0 || P < 0 || N < 0) return MMAE_ERR_ARG;
    layout(B, M, P, N, offsets15);
    return offsets15[14];
}

__global__ __launch_bounds__(256) void build_descriptors_kernel(BuildDesc d) {
    __shared__ int cnt[8];
    const int b = blockIdx.x, M = d.M, P = d.P, N = d.N, B = d.B, T = M * P;
    const long long* mask = d.mask_all + (long)(d.R == 1 ? 0 : b) * T;
    long off[15];
    layout(B, M, P, N, off);
    int* D = d.desc;
    if (threadIdx.x < 8) cnt[threadIdx.x] = 0;
    __syncthreads();
    if (threadIdx.x < M) {
        int c = 0;
        for (int e = 0; e < P; ++e) c += mask[threadIdx.x * P + e] == 0 ? 1 : 0;
        cnt[threadIdx.x] = c;
    }
    __syncthreads();
    int total = 0;
    for (int m = 0; m < M; ++m) total += cnt[m];
    const bool ok = total == N;
    if (threadIdx.x == 0 && !ok) atomicAdd(&D[off[13]], 1);
    const int fus_base = B * N, shared_base = B * N + B * P;
    if (threadIdx.x <= M) {
        const int s = threadIdx.x;
        int start = 0;
        for (int m = 0; m < s && m < M; ++m) start += cnt[m];
        const int len = s < M ? cnt[s] : P;
        D[off[0] + b * (M + 1) + s] = s < M ? b * N + start : fus_base + b * P;
        D[off[1] + b * (M + 1) + s] = ok ? len : 0;
        D[off[7] + b * (M + 1) + s] = b * (M + 1) + s;
        D[off[8] + b * (M + 1) + s] = 1;
        D[off[9] + b * (M + 1) + s] = s < M ? b * M + s : 0;
        D[off[10] + b * (M + 1) + s] = s < M ? 1 : 0;
        D[off[11] + b * (M + 1) + s] = s < M ? b * N + start : 0;
        D[off[12] + b * (M + 1) + s] = (s < M && ok) ? len : 0;
    }
    for (int j = threadIdx.x; j < T; j += blockDim.x) {
        const int m = j / P, e = j % P;
        const bool kept = ok && mask[j] == 0;
        int row = shared_base + e;
        if (kept) {
            int rank = 0;
            for (int k = 0; k < e; ++k) rank += mask[m * P + k] == 0 ? 1 : 0;
            int start = 0;
            for (int mm = 0; mm < m; ++mm) start += cnt[mm];
            const int pos = start + rank;              // position inside the sample's packed tokens
            row = b * N + pos;
            D[off[2] + row] = m;
            D[off[3] + row] = e;
            D[off[4] + row] = m * P + e;
            D[off[5] + row] = fus_base + b * P + e;
        }
        D[off[6] + ((long)b * P + e) * (M + 1) + m] = row;
    }
    for (int e = threadIdx.x; e < P; e += blockDim.x) D[off[6] + ((long)b * P + e) * (M + 1) + M] = fus_base + b * P + e;
    if (!ok) {
        // inconsistent sample: keep the descriptors memory-safe (all tokens point at row 0 of their tables)
        for (int i = threadIdx.x; i < N; i += blockDim.x) {
            D[off[2] + b * N + i] = 0; D[off[3] + b * N + i] = 0; D[off[4] + b * N + i] = 0; D[off[5] + b * N + i] = fus_base + b * P;
        }
    }
}

extern "C" int mmae_build_descriptors(int B, int R, int M, int P, int N, const long long* mask_all, int* desc, void* stream) {
    if (B <= 0 || (R != 1 && R != B) || M <= 0 || M > 7 || P <= 0 || N < 0 || N > M * P || !mask_all || !desc) return MMAE_ERR_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    long off[15];
    layout(B, M, P, N, off);
    if (hipMemsetAsync(desc + off[13], 0, 4 * sizeof(int), st) != hipSuccess) return MMAE_ERR_LAUNCH;
    BuildDesc d{mask_all, desc, B, R, M, P, N};
    MMAE_LAUNCH(build_descriptors_kernel, dim3(B), dim3(256), 0, st, d);
    MMAE_CHECK_LAUNCH();
    return MMAE_OK;
}
