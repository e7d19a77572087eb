// Graph-attention pieces of GATConv (src/module/gat.py:25-49):
//   e_ij = LeakyReLU_alpha(a1.h_i + a2.h_j);  e_ij = -9e15 where adj_ij == 0;  att = softmax_j(e)
// The pairwise tensor [B,N,N,2D] the reference materialises (gat.py:33-37) never exists here:
// a^T [h_i || h_j] splits into two per-node scalars s1 = h a1, s2 = h a2 (one skinny GEMM), and
// the N x N attention tile is built in registers, one wave64 per row with shuffle reductions.
// xggm_elu_* / xggm_dropout_* are the element-wise ends of the block (F.elu, F.dropout(x, .5)).
#include "common.h"
#include "xggm.h"

namespace {
constexpr int NT = 256;
inline int grid1d(int64_t n) { return (int)std::min<int64_t>(ceil_div64(n, NT), 2048); }

// s: [B*N, 2] fp32 (s1 = col 0, s2 = col 1); adj fp32 [B,N,N]; att out fp32 [B,N,N]
__global__ __launch_bounds__(NT) void gat_att_fwd_kernel(const float* __restrict__ s, const float* __restrict__ adj, float* att,
                                                         int B, int N, float alpha) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int rows = B * N;
    for (int r = blockIdx.x * 4 + wid; r < rows; r += gridDim.x * 4) {
        const int b = r / N;
        const float s1 = s[2 * r];
        float v = -INFINITY;
        if (lane < N) {
            const float raw = s1 + s[2 * (b * N + lane) + 1];
            v = raw > 0.f ? raw : alpha * raw;
            if (adj[(int64_t)r * N + lane] == 0.f) v = -9e15f;
        }
        const float m = wave_max(v);
        const float e = lane < N ? __expf(v - m) : 0.f;
        const float sum = wave_sum(e);
        if (lane < N) att[(int64_t)r * N + lane] = e / sum;
    }
}

// d_att -> ds [B*N, 2] (T).  One workgroup per sample: rows by waves, column sums through LDS.
template <typename T>
__global__ __launch_bounds__(NT) void gat_att_bwd_kernel(const float* __restrict__ d_att, const float* __restrict__ att,
                                                         const float* __restrict__ s, const float* __restrict__ adj, T* ds,
                                                         int N, float alpha) {
    __shared__ float De[64 * 65];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int i = wid; i < N; i += NT / 64) {
        const int r = b * N + i;
        float a = 0.f, da = 0.f;
        if (lane < N) {
            a = att[(int64_t)r * N + lane];
            da = d_att[(int64_t)r * N + lane];
        }
        const float dot = wave_sum(a * da);
        float de = 0.f;
        if (lane < N && adj[(int64_t)r * N + lane] != 0.f) {
            const float raw = s[2 * r] + s[2 * (b * N + lane) + 1];
            de = a * (da - dot) * (raw > 0.f ? 1.f : alpha);
        }
        if (lane < N) De[i * 65 + lane] = de;
        const float rs = wave_sum(de);
        if (lane == 0) ds[2 * (int64_t)r] = from_f32<T>(rs);
    }
    __syncthreads();
    if ((int)threadIdx.x < N) {
        float cs = 0.f;
        for (int i = 0; i < N; ++i) cs += De[i * 65 + threadIdx.x];
        ds[2 * ((int64_t)b * N + threadIdx.x) + 1] = from_f32<T>(cs);
    }
}

template <typename T>
__global__ __launch_bounds__(NT) void elu_fwd_kernel(const T* __restrict__ x, T* out, int M, int D, int64_t ld_out) {
    const int64_t n = (int64_t)M * D;
    for (int64_t t = (int64_t)blockIdx.x * NT + threadIdx.x; t < n; t += (int64_t)gridDim.x * NT) {
        const int m = (int)(t / D), c = (int)(t % D);
        const float v = to_f32(x[t]);
        out[(int64_t)m * ld_out + c] = from_f32<T>(v > 0.f ? v : expm1f(v));
    }
}
template <typename T>
__global__ __launch_bounds__(NT) void elu_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y, T* dx, int M, int D,
                                                     int64_t ld) {
    const int64_t n = (int64_t)M * D;
    for (int64_t t = (int64_t)blockIdx.x * NT + threadIdx.x; t < n; t += (int64_t)gridDim.x * NT) {
        const int m = (int)(t / D), c = (int)(t % D);
        const float yy = to_f32(y[(int64_t)m * ld + c]);
        const float g = to_f32(dy[(int64_t)m * ld + c]);
        dx[t] = from_f32<T>(yy > 0.f ? g : g * (yy + 1.f));
    }
}
template <typename T>
__global__ __launch_bounds__(NT) void dropout_kernel(const T* __restrict__ x, T* out, int64_t n, float p, const uint64_t* rng,
                                                     uint32_t sid) {
    const float ik = 1.f / (1.f - p);
    const uint64_t seed = rng[0], off = rng[1];
    for (int64_t t = (int64_t)blockIdx.x * NT + threadIdx.x; t < n; t += (int64_t)gridDim.x * NT)
        out[t] = from_f32<T>(to_f32(x[t]) * dropout_scale(p, ik, seed, off, sid, (uint64_t)t));
}
}  // namespace

extern "C" int xggm_gat_att_fwd(const float* s, const float* adj, float* att, int B, int N, float alpha, hipStream_t st) {
    XGGM_REQUIRE(s && adj && att && B > 0 && N > 0 && N <= 64, "xggm_gat_att_fwd: bad arguments B=%d N=%d", B, N);
    hipLaunchKernelGGL(gat_att_fwd_kernel, dim3(std::min(ceil_div(B * N, 4), 2048)), dim3(NT), 0, st, s, adj, att, B, N,
                       alpha);
    return xggm_check_launch("xggm_gat_att_fwd");
}

#define GAT_API(SUF, T)                                                                                                  \
    extern "C" int xggm_gat_att_bwd_##SUF(const float* d_att, const float* att, const float* s, const float* adj,       \
                                          void* ds, int B, int N, float alpha, hipStream_t st) {                        \
        XGGM_REQUIRE(d_att && att && s && adj && ds && B > 0 && N > 0 && N <= 64, "xggm_gat_att_bwd: bad arguments");   \
        hipLaunchKernelGGL((gat_att_bwd_kernel<T>), dim3(B), dim3(NT), 0, st, d_att, att, s, adj, (T*)ds, N, alpha);    \
        return xggm_check_launch("xggm_gat_att_bwd");                                                                   \
    }                                                                                                                    \
    extern "C" int xggm_elu_fwd_##SUF(const void* x, void* out, int M, int D, int64_t ld_out, hipStream_t st) {         \
        XGGM_REQUIRE(x && out && M > 0 && D > 0 && ld_out >= D, "xggm_elu_fwd: bad arguments");                          \
        hipLaunchKernelGGL((elu_fwd_kernel<T>), dim3(grid1d((int64_t)M * D)), dim3(NT), 0, st, (const T*)x, (T*)out, M,  \
                           D, ld_out);                                                                                  \
        return xggm_check_launch("xggm_elu_fwd");                                                                       \
    }                                                                                                                    \
    extern "C" int xggm_elu_bwd_##SUF(const void* dy, const void* y, void* dx, int M, int D, int64_t ld,                \
                                      hipStream_t st) {                                                                 \
        XGGM_REQUIRE(dy && y && dx && M > 0 && D > 0 && ld >= D, "xggm_elu_bwd: bad arguments");                         \
        hipLaunchKernelGGL((elu_bwd_kernel<T>), dim3(grid1d((int64_t)M * D)), dim3(NT), 0, st, (const T*)dy,            \
                           (const T*)y, (T*)dx, M, D, ld);                                                              \
        return xggm_check_launch("xggm_elu_bwd");                                                                       \
    }                                                                                                                    \
    extern "C" int xggm_dropout_##SUF(const void* x, void* out, int64_t n, float p, const uint64_t* rng, uint32_t sid,  \
                                      hipStream_t st) {                                                                 \
        XGGM_REQUIRE(x && out && n > 0 && p > 0.f && p < 1.f && rng, "xggm_dropout: bad arguments");                     \
        hipLaunchKernelGGL((dropout_kernel<T>), dim3(grid1d(n)), dim3(NT), 0, st, (const T*)x, (T*)out, n, p, rng,      \
                           sid);                                                                                        \
        return xggm_check_launch("xggm_dropout");                                                                       \
    }

GAT_API(f32, float)
GAT_API(bf16, bf16)
