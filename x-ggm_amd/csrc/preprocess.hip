// Offline preprocessing that feeds the hot path.
//
//   xggm_cosine_adjacency_f32   the 36 x 36 "object attribute-class cosine similarity" adjacency every training
//                               sample carries (adj_true): data/preprocess/vqa/compute_adjacency.py:38-45
//                               (compute_cosin_sim_v2: a Python double loop of 666 torch.cosine_similarity calls
//                               per image) followed by matrix / matrix.max() (:90).
// One workgroup per image: the class and attribute embeddings [N, D] stream through LDS once in 64-wide chunks,
// every thread accumulates its share of the N x N dot products and of the 2N squared norms in registers, then the
// matrix is masked to j >= i, mirrored (the diagonal counts twice, as in the reference: adj + adj^T) and divided
// by its maximum.  HBM traffic: 2 N D floats in, N N floats out -- the kernel is a stream.
#include "common.h"
#include "xggm.h"

namespace {

constexpr int NT = 256, DC = 64;

template <int NP>
__global__ __launch_bounds__(NT) void cosine_adj_kernel(const float* __restrict__ C, const float* __restrict__ A, float* out,
                                                        int N, int D, float eps) {
    constexpr int PP = (NP * NP + NT - 1) / NT;  // (i, j) pairs per thread
    __shared__ float cs[NP * (DC + 1)];
    __shared__ float as[NP * (DC + 1)];
    __shared__ float S[NP * (NP + 1)];
    __shared__ float nrm[2 * NP];
    __shared__ float red[NT / 64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const float* Cb = C + (int64_t)blockIdx.x * N * D;
    const float* Ab = A + (int64_t)blockIdx.x * N * D;
    float dot[PP];
#pragma unroll
    for (int p = 0; p < PP; ++p) dot[p] = 0.f;
    float n2 = 0.f;  // squared norm of row tid (class rows first, then attribute rows)
    for (int k0 = 0; k0 < D; k0 += DC) {
        for (int e = tid; e < N * (DC / 4); e += NT) {
            const int r = e / (DC / 4), c = (e % (DC / 4)) * 4;
            float4 vc = make_float4(0.f, 0.f, 0.f, 0.f), va = vc;
            if (k0 + c < D) {  // D % 4 == 0
                vc = *reinterpret_cast<const float4*>(Cb + (int64_t)r * D + k0 + c);
                va = *reinterpret_cast<const float4*>(Ab + (int64_t)r * D + k0 + c);
            }
            float* pc = cs + r * (DC + 1) + c;
            float* pa = as + r * (DC + 1) + c;
            pc[0] = vc.x; pc[1] = vc.y; pc[2] = vc.z; pc[3] = vc.w;
            pa[0] = va.x; pa[1] = va.y; pa[2] = va.z; pa[3] = va.w;
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < PP; ++p) {
            const int e = tid + p * NT;
            if (e < N * N) {
                const float* pc = cs + (e / N) * (DC + 1);
                const float* pa = as + (e % N) * (DC + 1);
                float s = dot[p];
#pragma unroll 16
                for (int k = 0; k < DC; ++k) s = fmaf(pc[k], pa[k], s);
                dot[p] = s;
            }
        }
        if (tid < 2 * N) {
            const float* pr = (tid < N ? cs + tid * (DC + 1) : as + (tid - N) * (DC + 1));
            float s = n2;
#pragma unroll 16
            for (int k = 0; k < DC; ++k) s = fmaf(pr[k], pr[k], s);
            n2 = s;
        }
        __syncthreads();
    }
    if (tid < 2 * N) nrm[tid] = fmaxf(sqrtf(n2), eps);  // torch.cosine_similarity: each norm clamped from below
    __syncthreads();
#pragma unroll
    for (int p = 0; p < PP; ++p) {
        const int e = tid + p * NT;
        if (e < N * N) {
            const int i = e / N, j = e % N;
            S[i * (NP + 1) + j] = j >= i ? dot[p] / (nrm[i] * nrm[N + j]) : 0.f;
        }
    }
    __syncthreads();
    float v[PP], mx = -INFINITY;
#pragma unroll
    for (int p = 0; p < PP; ++p) {
        const int e = tid + p * NT;
        v[p] = 0.f;
        if (e < N * N) {
            const int i = e / N, j = e % N;
            v[p] = S[i * (NP + 1) + j] + S[j * (NP + 1) + i];
            mx = fmaxf(mx, v[p]);
        }
    }
    mx = wave_max(mx);
    if (lane == 0) red[wid] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float* ob = out + (int64_t)blockIdx.x * N * N;
#pragma unroll
    for (int p = 0; p < PP; ++p) {
        const int e = tid + p * NT;
        if (e < N * N) ob[e] = v[p] / mx;
    }
}

}  // namespace

extern "C" int xggm_cosine_adjacency_f32(const float* cls, const float* attr, float* adj, int n_img, int N, int D, float eps,
                                         hipStream_t st) {
    XGGM_REQUIRE(cls && attr && adj && n_img > 0, "xggm_cosine_adjacency_f32: bad arguments");
    XGGM_REQUIRE(N >= 1 && N <= 64, "xggm_cosine_adjacency_f32: N = %d objects (1..64)", N);
    XGGM_REQUIRE(D > 0 && D % 4 == 0, "xggm_cosine_adjacency_f32: embedding width %d must be a multiple of 4", D);
    XGGM_REQUIRE(reinterpret_cast<uintptr_t>(cls) % 16 == 0 && reinterpret_cast<uintptr_t>(attr) % 16 == 0,
                 "xggm_cosine_adjacency_f32: embeddings must be 16-byte aligned");
    if (N <= 36) hipLaunchKernelGGL((cosine_adj_kernel<36>), dim3(n_img), dim3(NT), 0, st, cls, attr, adj, N, D, eps);
    else hipLaunchKernelGGL((cosine_adj_kernel<64>), dim3(n_img), dim3(NT), 0, st, cls, attr, adj, N, D, eps);
    return xggm_check_launch("xggm_cosine_adjacency_f32");
}
