// Loss kernels and the fused optimiser.
//   xggm_dsm_loss_fwd/bwd   denoising score matching  (src/vqa/vqacpv2.py:48-51)
//   xggm_symkl_fwd/bwd      symmetric KL of row softmaxes (src/vqa/vqacpv2.py:54-61)
//   xggm_bce_fwd/bwd        BCEWithLogits(mean) * A   (src/vqa/vqacpv2.py:131,173)
//   xggm_sqnorm_f32         sum of squares of a flat fp32 gradient range (clip_grad_norm_, vqacpv2.py:175)
//   xggm_bertadam_f32       clip-scale + BertAdam update + bf16 shadow weight, ONE pass over
//                           p/g/m/v (src/lxrt/optimization.py:159-193): 16 B read + 12(+2) B
//                           written per parameter, the HBM floor of the update.
//   xggm_sched_step         warmup_linear(step/t_total) on the device-resident step counter
//                           (optimization.py:42-48,177-181) so a captured graph replays correctly
// Scalars (losses, norms, schedule values, upstream gradients) live in device memory: nothing
// here synchronises with the host.
#include "common.h"
#include "xggm.h"

namespace {

constexpr int NT = 256;

__device__ __forceinline__ float block_sum(float v) {
    __shared__ float red[NT / 64];
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

inline int grid1d(int64_t n, int cap = 1024) { return (int)std::min<int64_t>(ceil_div64(n, NT), cap); }
// workgroups of a kernel that ends in ordered_grid_sum: every one pays a ticket, and the last one adds them all.  While
// the ticket sat behind a device-scope release fence (a write-back of the XCD's L2), 1024 workgroups made the DSM loss
// 36 us instead of 15 and the grid was held at 128 -- which left the symmetric KL's forward with nine rows per workgroup
// (29 us against its backward's 12).  The fence is gone (common.h): one workgroup per four rows again, up to 512.
constexpr int LOSS_GRID = 512;

// ------------------------------------------------------------------------------- DSM
template <typename T>
__global__ __launch_bounds__(NT) void dsm_fwd_kernel(const T* __restrict__ s, const float* __restrict__ g, float* loss,
                                                     int64_t n, float coef, float* ws) {
    float acc = 0.f;
    const int64_t n4 = ((reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(g)) & 15) == 0 ? n >> 2 : 0;
    for (int64_t t = (int64_t)blockIdx.x * NT + threadIdx.x; t < n4; t += (int64_t)gridDim.x * NT) {  // four elements per load
        float sv[4], gv[4];
        load4(s + 4 * t, sv);
        load4(g + 4 * t, gv);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float d = sv[i] - gv[i];
            acc += d * d;
        }
    }
    for (int64_t t = 4 * n4 + (int64_t)blockIdx.x * NT + threadIdx.x; t < n; t += (int64_t)gridDim.x * NT) {
        const float d = to_f32(s[t]) - g[t];
        acc += d * d;
    }
    acc = block_sum(acc);
    float total;
    if (ordered_grid_sum(acc, ws, gridDim.x, blockIdx.x, total)) *loss += total * coef;
}
template <typename T>
__global__ __launch_bounds__(NT) void dsm_bwd_kernel(const T* __restrict__ s, const float* __restrict__ g,
                                                     const float* __restrict__ gout, T* ds, int64_t n, float coef) {
    const float k = 2.f * coef * (gout ? *gout : 1.f);
    for (int64_t t = (int64_t)blockIdx.x * NT + threadIdx.x; t < n; t += (int64_t)gridDim.x * NT)
        ds[t] = from_f32<T>(k * (to_f32(s[t]) - g[t]));
}

// ------------------------------------------------------------------------------- symmetric KL
// per row: f = sum_c (px - py)(lpx - lpy);  df/dx_k = px_k (u_k - <u>_px) + w_k,
// df/dy_k = -py_k (u_k - <u>_py) - w_k  with u = lpx - lpy, w = px - py.
template <typename T, int NV>
__global__ __launch_bounds__(NT) void symkl_kernel(const T* __restrict__ x, const T* __restrict__ y, float* loss,
                                                   const float* __restrict__ gout, T* dx, T* dy, int rows, int W, float coef,
                                                   int accumulate, float* ws) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const float gk = coef * ((dx || dy) && gout ? *gout : 1.f);
    float total = 0.f;
    for (int row = blockIdx.x * 4 + wid; row < rows; row += gridDim.x * 4) {
        const int64_t rb = (int64_t)row * W;
        float xv[NV], yv[NV];
        float mx = -INFINITY, my = -INFINITY;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int c = v * 64 + lane;
            xv[v] = c < W ? to_f32(x[rb + c]) : -INFINITY;
            yv[v] = c < W ? to_f32(y[rb + c]) : -INFINITY;
            mx = fmaxf(mx, xv[v]);
            my = fmaxf(my, yv[v]);
        }
        mx = wave_max(mx);
        my = wave_max(my);
        float sx = 0.f, sy = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int c = v * 64 + lane;
            if (c < W) {
                sx += __expf(xv[v] - mx);
                sy += __expf(yv[v] - my);
            }
        }
        const float lx = mx + __logf(wave_sum(sx)), ly = my + __logf(wave_sum(sy));
        float f = 0.f, upx = 0.f, upy = 0.f;
        float u[NV], px[NV], py[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int c = v * 64 + lane;
            u[v] = px[v] = py[v] = 0.f;
            if (c < W) {
                const float lpx = xv[v] - lx, lpy = yv[v] - ly;
                px[v] = __expf(lpx);
                py[v] = __expf(lpy);
                u[v] = lpx - lpy;
                f += (px[v] - py[v]) * u[v];
                upx += px[v] * u[v];
                upy += py[v] * u[v];
            }
        }
        total += f;  // lane partial; reduced once per block below
        if (dx || dy) {
            upx = wave_sum(upx);
            upy = wave_sum(upy);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int c = v * 64 + lane;
                if (c < W) {
                    const float w = px[v] - py[v];
                    if (dx) {
                        float gx = gk * (px[v] * (u[v] - upx) + w);
                        if (accumulate) gx += to_f32(dx[rb + c]);
                        dx[rb + c] = from_f32<T>(gx);
                    }
                    if (dy) {
                        float gy = gk * (-py[v] * (u[v] - upy) - w);
                        if (accumulate) gy += to_f32(dy[rb + c]);
                        dy[rb + c] = from_f32<T>(gy);
                    }
                }
            }
        }
    }
    if (loss) {
        total = block_sum(total);
        float sum;
        if (ordered_grid_sum(total, ws, gridDim.x, blockIdx.x, sum)) *loss += sum * coef;
    }
}

// ------------------------------------------------------------------------------- BCE with logits
__global__ __launch_bounds__(NT) void bce_fwd_kernel(const float* __restrict__ l, const float* __restrict__ t, float* loss,
                                                     int64_t n, float coef, float* ws) {
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const float z = l[i];
        acc += fmaxf(z, 0.f) - z * t[i] + log1pf(__expf(-fabsf(z)));
    }
    acc = block_sum(acc);
    float total;
    if (ordered_grid_sum(acc, ws, gridDim.x, blockIdx.x, total)) *loss += total * coef;
}
template <typename T>
__global__ __launch_bounds__(NT) void bce_bwd_kernel(const float* __restrict__ l, const float* __restrict__ t,
                                                     const float* __restrict__ gout, T* dl, int64_t n, float coef) {
    const float k = coef * (gout ? *gout : 1.f);
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT)
        dl[i] = from_f32<T>(k * (sigmoid_f(l[i]) - t[i]));
}

// ------------------------------------------------------------------------------- optimiser
__global__ __launch_bounds__(NT) void sqnorm_kernel(const float* __restrict__ g, int64_t n, float* out, float* ws) {
    float acc = 0.f;
    const int64_t n4 = n >> 2;
    typedef float __attribute__((ext_vector_type(4))) f4;
    const f4* g4 = reinterpret_cast<const f4*>(g);
    const int64_t stride = (int64_t)gridDim.x * NT;
    int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x;
    // four independent 16-byte loads in flight per thread
    for (; i + 3 * stride < n4; i += 4 * stride) {
        const f4 a = __builtin_nontemporal_load(g4 + i), b = __builtin_nontemporal_load(g4 + i + stride),
                 c = __builtin_nontemporal_load(g4 + i + 2 * stride), d = __builtin_nontemporal_load(g4 + i + 3 * stride);
        acc += (a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w) + (b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w) +
               (c.x * c.x + c.y * c.y + c.z * c.z + c.w * c.w) + (d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w);
    }
    for (; i < n4; i += stride) {
        const f4 v = g4[i];
        acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const float v = g[(n4 << 2) + threadIdx.x];
        acc += v * v;
    }
    acc = block_sum(acc);
    if (threadIdx.x == 0) ws[blockIdx.x] = acc;  // per-workgroup partial; summed in index order by sqnorm_finish_kernel
}

// second stage: one workgroup adds the partials in a fixed order (no floating-point atomics: replicas that hold
// the same gradients get the same bits).  A separate launch instead of a last-arriver finish in the first
// kernel: the device-scope fence that needs costs an L2 write-back per workgroup (3.5 -> 1.6 TB/s measured).
__global__ __launch_bounds__(NT) void sqnorm_finish_kernel(const float* __restrict__ ws, int nblk, float* out) {
    float t = 0.f;
    for (int b = threadIdx.x; b < nblk; b += NT) t += ws[b];
    t = block_sum(t);
    if (threadIdx.x == 0) *out += t;
}

// clip_grad_norm_ over SEVERAL ranges of one fp32 buffer in two launches: every workgroup owns one slice of one
// range (ranges get workgroups in proportion to their length), partials land in ws in (range, slice) order and one
// workgroup adds them in that order -- the same bits whatever the scheduling.  The finish also seeds / extends the
// running sum and writes the norm, so the host issues no fill, add or sqrt kernels around it.
constexpr int MAX_SPANS = 16;
struct Spans {
    int64_t off[MAX_SPANS], len[MAX_SPANS];
    int blk0[MAX_SPANS + 1];  // first workgroup of each range
    int n;
};
// one workgroup's share of one range: slice b of nb, four 16-byte loads in flight per thread
template <bool SQ>
__device__ __forceinline__ float span_partial(const float* __restrict__ g, int64_t n, int b, int nb) {
    const int64_t n4 = n >> 2;
    typedef float __attribute__((ext_vector_type(4))) f4;
    const f4* g4 = reinterpret_cast<const f4*>(g);
    const int64_t stride = (int64_t)nb * NT;
    float acc = 0.f;
    int64_t i = (int64_t)b * NT + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        const f4 a = __builtin_nontemporal_load(g4 + i), bb = __builtin_nontemporal_load(g4 + i + stride),
                 c = __builtin_nontemporal_load(g4 + i + 2 * stride), d = __builtin_nontemporal_load(g4 + i + 3 * stride);
        if (SQ)
            acc += (a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w) + (bb.x * bb.x + bb.y * bb.y + bb.z * bb.z + bb.w * bb.w) +
                   (c.x * c.x + c.y * c.y + c.z * c.z + c.w * c.w) + (d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w);
        else
            acc += ((a.x + a.y) + (a.z + a.w)) + ((bb.x + bb.y) + (bb.z + bb.w)) + ((c.x + c.y) + (c.z + c.w)) +
                   ((d.x + d.y) + (d.z + d.w));
    }
    for (; i < n4; i += stride) {
        const f4 v = g4[i];
        acc += SQ ? v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w : (v.x + v.y) + (v.z + v.w);
    }
    if (b == 0 && threadIdx.x < (n & 3)) {
        const float v = g[(n4 << 2) + threadIdx.x];
        acc += SQ ? v * v : v;
    }
    return block_sum(acc);
}

template <bool SQ>
__global__ __launch_bounds__(NT) void sqnorm_multi_kernel(const float* __restrict__ base, Spans sp, float* ws) {
    int k = 0;
#pragma unroll
    for (int j = 1; j < MAX_SPANS; ++j)
        if (j < sp.n && (int)blockIdx.x >= sp.blk0[j]) k = j;
    const float acc = span_partial<SQ>(base + sp.off[k], sp.len[k], blockIdx.x - sp.blk0[k], sp.blk0[k + 1] - sp.blk0[k]);
    if (threadIdx.x == 0) ws[blockIdx.x] = acc;
}

// The norm of a whole pass in ONE pair of launches (xggm_clip_norm_f32): ranges of the gradient buffer (squared) and
// ranges of the slot table the weight-gradient products filled (already sums of squares: added as they are) side by
// side in one grid, partials in (range, slice) order; the finishing workgroup also takes the schedule step and the
// RNG advance of the pass along (three one-workgroup launches of their own before).
constexpr int MAX_NORM_SPANS = 24;
struct NormSpans {
    const float* ptr[MAX_NORM_SPANS];
    int64_t len[MAX_NORM_SPANS];
    int blk0[MAX_NORM_SPANS + 1];
    int n, n_sq;  // ranges [0, n_sq) are squared, [n_sq, n) summed as they are
};
__global__ __launch_bounds__(NT) void clip_norm_kernel(NormSpans sp, float* ws) {
    int k = 0;
#pragma unroll
    for (int j = 1; j < MAX_NORM_SPANS; ++j)
        if (j < sp.n && (int)blockIdx.x >= sp.blk0[j]) k = j;
    const int b = blockIdx.x - sp.blk0[k], nb = sp.blk0[k + 1] - sp.blk0[k];
    const float acc = k < sp.n_sq ? span_partial<true>(sp.ptr[k], sp.len[k], b, nb) : span_partial<false>(sp.ptr[k], sp.len[k], b, nb);
    if (threadIdx.x == 0) ws[blockIdx.x] = acc;
}
// the same grid over ranges of a bf16 buffer (the data-parallel wire arena): 8 elements per 16-byte load, four in flight
__device__ __forceinline__ float span_partial_bf16(const bf16* __restrict__ g, int64_t n, int b, int nb) {
    typedef short __attribute__((ext_vector_type(8))) s8;
    const s8* g8 = reinterpret_cast<const s8*>(g);
    const int64_t n8 = n >> 3, stride = (int64_t)nb * NT;
    float acc = 0.f;
    auto sq8 = [](const s8 v) {
        float t = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float f = __bfloat162float(__builtin_bit_cast(bf16, (short)v[e]));
            t += f * f;
        }
        return t;
    };
    int64_t i = (int64_t)b * NT + threadIdx.x;
    for (; i + 3 * stride < n8; i += 4 * stride) {
        const s8 a = __builtin_nontemporal_load(g8 + i), bb = __builtin_nontemporal_load(g8 + i + stride),
                 c = __builtin_nontemporal_load(g8 + i + 2 * stride), d = __builtin_nontemporal_load(g8 + i + 3 * stride);
        acc += (sq8(a) + sq8(bb)) + (sq8(c) + sq8(d));
    }
    for (; i < n8; i += stride) acc += sq8(g8[i]);
    if (b == 0 && threadIdx.x < (n & 7)) {
        const float f = __bfloat162float(g[(n8 << 3) + threadIdx.x]);
        acc += f * f;
    }
    return block_sum(acc);
}
struct NormSpans16 {
    const bf16* ptr[MAX_NORM_SPANS];
    int64_t len[MAX_NORM_SPANS];
    int blk0[MAX_NORM_SPANS + 1];
    int n;
};
__global__ __launch_bounds__(NT) void clip_norm_bf16_kernel(NormSpans16 sp, float* ws) {
    int k = 0;
#pragma unroll
    for (int j = 1; j < MAX_NORM_SPANS; ++j)
        if (j < sp.n && (int)blockIdx.x >= sp.blk0[j]) k = j;
    const float acc = span_partial_bf16(sp.ptr[k], sp.len[k], blockIdx.x - sp.blk0[k], sp.blk0[k + 1] - sp.blk0[k]);
    if (threadIdx.x == 0) ws[blockIdx.x] = acc;
}

__global__ __launch_bounds__(NT) void sqnorm_multi_finish_kernel(const float* __restrict__ ws, int nblk, float* out, float* norm,
                                                                 int overwrite, float mul) {
    float t = 0.f;
    for (int b = threadIdx.x; b < nblk; b += NT) t += ws[b];
    t = block_sum(t);
    if (threadIdx.x == 0) {
        const float s = ((overwrite ? 0.f : *out) + t) * mul;
        *out = s;
        if (norm) *norm = sqrtf(s);
    }
}

// the same sum over a bf16 buffer (data-parallel wire arena): 8 elements per 16-byte load
__global__ __launch_bounds__(NT) void sqnorm_bf16_kernel(const bf16* __restrict__ g, int64_t n, float* ws) {
    float acc = 0.f;
    const int64_t n8 = n >> 3;
    typedef short __attribute__((ext_vector_type(8))) s8;
    const s8* g8 = reinterpret_cast<const s8*>(g);
    const int64_t stride = (int64_t)gridDim.x * NT;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n8; i += stride) {
        const s8 v = __builtin_nontemporal_load(g8 + i);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float f = __bfloat162float(__builtin_bit_cast(bf16, (short)v[e]));
            acc += f * f;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
        const float f = __bfloat162float(g[(n8 << 3) + threadIdx.x]);
        acc += f * f;
    }
    acc = block_sum(acc);
    if (threadIdx.x == 0) ws[blockIdx.x] = acc;
}

struct AdamArgs {
    float* p;
    const void* g;          // fp32, or bf16 when GB16
    float* m;
    float* v;
    bf16* shadow;
    int64_t n;
    const float* sqnorm;    // device scalar: sum of squares over ALL grads of the step (or null)
    float max_norm;
    float lr;
    const float* lr_dev;    // device scalar replacing lr (or null)
    const float* lr_scale;  // device scalar from xggm_sched_step (or null = 1)
    float b1, b2, eps, wd;
    unsigned char* shadow8;     // e4m3 copy of the updated weights (or null)
    const uint16_t* w8_id;      // per 256-element arena chunk: entry of the scale table, 0 = no e4m3 copy
    const float* w8_qscale;
    float* w8_amax;
    int64_t elem0;              // arena offset of p[0] (multiple of 256 with shadow8)
    float g_scale;              // gradients are SUMS over data-parallel ranks: 1 / world (1 otherwise)
    int w8_slots;               // floats every w8_amax entry is spread over (>= 1)
};

// One element of src/lxrt/optimization.py:159-193 (no bias correction, decoupled weight decay) with the shape of every
// multiply-add PINNED: which products fuse into an fma is otherwise the compiler's choice per instantiation under
// -ffp-contract=fast (the backend fuses whatever the source says), and the fp32- and the bf16-gradient instantiation of
// this template then differed in the last bit of v and p.  __fmul_rn / __fadd_rn are rounded on their own and never
// fused; the three fmaf are the fusions wanted.
__device__ __forceinline__ void adam_update(float& p, float& m, float& v, float gk, float b1, float b2, float eps, float wd, float lr) {
    m = __fmaf_rn(m, b1, __fmul_rn(1.f - b1, gk));
    v = __fmaf_rn(v, b2, __fmul_rn(__fmul_rn(1.f - b2, gk), gk));
    const float upd = __fadd_rn(m / __fadd_rn(sqrtf(v), eps), __fmul_rn(wd, p));
    p = __fmaf_rn(-lr, upd, p);
}

// One float4 of each of p, g, m, v per step; UNR independent float4 quadruples per thread and
// iteration (loads issued before any use).  g is read once and m, v, shadow are not re-read before the
// next step: non-temporal accesses keep them from displacing the weights' bf16 shadow in L2/MALL.
template <int UNR, bool NTMP, bool GB16>
__device__ __forceinline__ void bertadam_body(const AdamArgs& a, const int blk, const int nblk) {
    float coef = 1.f;
    if (a.sqnorm) {
        const float total = sqrtf(*a.sqnorm);
        coef = fminf(a.max_norm / (total + 1e-6f), 1.f);  // torch.nn.utils.clip_grad_norm_
    }
    coef *= a.g_scale;
    const float lr = (a.lr_dev ? *a.lr_dev : a.lr) * (a.lr_scale ? *a.lr_scale : 1.f);
    const int64_t n4 = a.n >> 2;
    typedef float __attribute__((ext_vector_type(4))) f4;
    typedef short __attribute__((ext_vector_type(4))) s4;
    const int64_t stride = (int64_t)nblk * NT;
    const int lane = threadIdx.x & 63;
    // every wave handles 64 consecutive float4 = one 256-element chunk of the arena per unrolled step, so the
    // chunk's scale-table entry is wave-uniform.  The trip count is made block-uniform (i0 of thread 0) so that
    // the wave reduction below runs with all lanes present; lanes beyond n4 only idle.
    for (int64_t b0 = (int64_t)blk * NT; b0 < n4; b0 += stride * UNR) {
        const int64_t i0 = b0 + threadIdx.x;
        f4 p[UNR], g[UNR], m[UNR], v[UNR];
        // e4m3 copy: the chunk's scale-table entry and its scale are fetched FIRST -- the oldest loads of the iteration,
        // so waiting for them (to issue the dependent scale load) leaves the streams below in flight, and both are long
        // back when the packed bytes are stored (they sat behind the update's arithmetic before: a dependent L2 round
        // trip or two at the end of every step)
        unsigned w8id[UNR];
        if (a.shadow8) {
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int64_t iw = (b0 + (threadIdx.x & ~63)) + u * stride;
                w8id[u] = iw < n4 ? a.w8_id[(a.elem0 + 4 * iw) >> 8] : 0u;
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int64_t i = i0 + u * stride;
            if (i < n4) {
                p[u] = NTMP ? __builtin_nontemporal_load(reinterpret_cast<const f4*>(a.p) + i) : reinterpret_cast<const f4*>(a.p)[i];
                if (GB16) {
                    const s4 gb = __builtin_nontemporal_load(reinterpret_cast<const s4*>(a.g) + i);
#pragma unroll
                    for (int k = 0; k < 4; ++k) g[u][k] = __bfloat162float(__builtin_bit_cast(bf16, (short)gb[k]));
                } else if (NTMP) {
                    g[u] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(a.g) + i);
                } else {
                    g[u] = reinterpret_cast<const f4*>(a.g)[i];
                }
                if (NTMP) {
                    m[u] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(a.m) + i);
                    v[u] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(a.v) + i);
                } else {
                    m[u] = reinterpret_cast<const f4*>(a.m)[i];
                    v[u] = reinterpret_cast<const f4*>(a.v)[i];
                }
            }
        }
        float w8q[UNR];
        if (a.shadow8) {
#pragma unroll
            for (int u = 0; u < UNR; ++u) w8q[u] = w8id[u] ? a.w8_qscale[w8id[u]] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int64_t i = i0 + u * stride;
            const bool on = i < n4;
            if (on) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float pk = p[u][k], mk = m[u][k], vk = v[u][k];
                    adam_update(pk, mk, vk, __fmul_rn(g[u][k], coef), a.b1, a.b2, a.eps, a.wd, lr);
                    p[u][k] = pk;
                    m[u][k] = mk;
                    v[u][k] = vk;
                }
                if (NTMP) __builtin_nontemporal_store(p[u], reinterpret_cast<f4*>(a.p) + i);
                else reinterpret_cast<f4*>(a.p)[i] = p[u];
                if (NTMP) {
                    __builtin_nontemporal_store(m[u], reinterpret_cast<f4*>(a.m) + i);
                    __builtin_nontemporal_store(v[u], reinterpret_cast<f4*>(a.v) + i);
                } else {
                    reinterpret_cast<f4*>(a.m)[i] = m[u];
                    reinterpret_cast<f4*>(a.v)[i] = v[u];
                }
                if (a.shadow) {
                    s4 sh;
#pragma unroll
                    for (int k = 0; k < 4; ++k) sh[k] = __builtin_bit_cast(short, __float2bfloat16(p[u][k]));
                    reinterpret_cast<s4*>(a.shadow)[i] = sh;
                }
            }
            if (a.shadow8) {
                // chunk of lane 0 == chunk of every lane of the wave; lanes past the end contribute nothing
                const unsigned id = w8id[u];
                if (id) {
                    const float q = w8q[u];
                    float mx = 0.f;
                    if (on) {
                        mx = fmaxf(fmaxf(fabsf(p[u][0]), fabsf(p[u][1])), fmaxf(fabsf(p[u][2]), fabsf(p[u][3])));
                        reinterpret_cast<int*>(a.shadow8)[i] = pack4_e4m3(p[u][0], p[u][1], p[u][2], p[u][3], q);
                    }
                    // Only maxima near or beyond the representable range matter to the next scale (e4m3 is a
                    // floating-point format: a stale smaller range costs nothing until values shrink by orders of
                    // magnitude, see xggm_fp8_scale_update): a few per cent of the waves issue the atomic.
                    mx = wave_max(mx);
                    if (lane == 0 && mx > 0.75f * 448.f / q) amax_record(a.w8_amax + (int64_t)id * a.w8_slots, a.w8_slots, blk, mx);
                }
            }
        }
    }
    if (blk == 0 && threadIdx.x < (a.n & 3)) {  // (ranges with an e4m3 copy are multiples of 256: no tail there)
        const int64_t i = (n4 << 2) + threadIdx.x;
        const float gi = GB16 ? __bfloat162float(reinterpret_cast<const bf16*>(a.g)[i]) : reinterpret_cast<const float*>(a.g)[i];
        float p = a.p[i], m = a.m[i], v = a.v[i];
        adam_update(p, m, v, __fmul_rn(gi, coef), a.b1, a.b2, a.eps, a.wd, lr);
        a.m[i] = m;
        a.v[i] = v;
        a.p[i] = p;
        if (a.shadow) a.shadow[i] = __float2bfloat16(p);
    }
}

template <int UNR, bool NTMP, bool GB16>
__global__ __launch_bounds__(NT) void bertadam_kernel(AdamArgs a) {
    bertadam_body<UNR, NTMP, GB16>(a, (int)blockIdx.x, (int)gridDim.x);
}

// several spans (the arena groups a pass updates, or a rank's slices of them) in ONE launch: a workgroup finds its span
// in the prefix table and runs the same body on its share of that span's grid.  Four launches per pass before; every
// launch boundary leaves the chip draining one update's tail while the next one's first loads have not been issued.
constexpr int ADAM_MULTI = 8;
struct AdamMulti {
    AdamArgs a[ADAM_MULTI];
    int blk0[ADAM_MULTI + 1];
    int n;
};
template <int UNR, bool NTMP, bool GB16>
__global__ __launch_bounds__(NT) void bertadam_multi_kernel(AdamMulti am) {
    int j = 0;
#pragma unroll
    for (int k = 1; k < ADAM_MULTI; ++k)
        if (k < am.n && (int)blockIdx.x >= am.blk0[k]) j = k;
    bertadam_body<UNR, NTMP, GB16>(am.a[j], (int)blockIdx.x - am.blk0[j], am.blk0[j + 1] - am.blk0[j]);
}

// Delayed scaling of the e4m3 operands (weights and activation sites share one table).  Producers record the
// maximum of a step only when it comes near the representable range 448 / q (> 1/2 of it; the weight update
// > 3/4), so per entry: a recorded maximum m within the history -> range = margin * m; nothing recorded for a whole
// history -> the range halves when `shrink` (values have shrunk a lot; still no overflow, finer subnormals) or stays
// (weights).  qscale <= 0 marks an entry that has not been calibrated: its producers quantise with 1 and record
// every maximum, and it stays uncalibrated until something was recorded.
// One workgroup of 16 waves; a wave takes FOUR entries per round with all eight loads in flight together (lane s reads
// slot s of an entry's maximum, lane j element j of its history: coalesced, reduced with wave_max).  The first form --
// one thread per entry walking its 64 slots -- took 12.8 us per launch, five launches per iteration of the fp8 step.
__global__ __launch_bounds__(1024) void fp8_scale_update_kernel(float* amax, float* hist, float* qscale, float* dscale, int64_t* pos,
                                                                int n, int hist_len, float margin, int shrink, int bump, int slots) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int slot = (int)(*pos % hist_len);
    constexpr int E = 4;
    for (int i0 = wid * E; i0 < n; i0 += 16 * E) {
        float am[E], h[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int i = i0 + e;
            am[e] = (i < n && lane < slots) ? amax[(int64_t)i * slots + lane] : 0.f;
            h[e] = (i < n && lane < hist_len) ? hist[(int64_t)i * hist_len + lane] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int i = i0 + e;
            if (i < n && lane < slots) amax[(int64_t)i * slots + lane] = 0.f;
            const float a = wave_max(am[e]);
            const float m = wave_max(lane == slot ? a : h[e]);  // the history with this step's maximum in its slot
            if (i < n && lane == 0) {
                hist[(int64_t)i * hist_len + slot] = a;
                const float q0 = qscale[i];
                float q = q0;
                if (m > 0.f) q = 448.f / (m * margin);
                else if (q0 > 0.f && shrink && slot == hist_len - 1) q = fminf(q0 * 2.f, 1.0995e12f);
                if (q > 0.f) {
                    qscale[i] = q;
                    dscale[i] = 1.f / q;
                }
            }
        }
    }
    __syncthreads();  // every wave has read *pos
    if (bump && threadIdx.x == 0) *pos += 1;
}

// lr_scale = warmup_linear(step / t_total, warmup); step += 1   (t_total <= 0: scale = 1)
__global__ void sched_kernel(int64_t* step, float* lr_scale, int64_t t_total, float warmup) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int64_t s = *step;
    float sc = 1.f;
    if (t_total > 0) {
        const float x = (float)((double)s / (double)t_total);
        sc = x < warmup ? x / warmup : fmaxf((x - 1.f) / (warmup - 1.f), 0.f);
    }
    *lr_scale = sc;
    *step = s + 1;
}

// the same for several counters of one table in ONE launch (all parameter groups of an optimiser step)
constexpr int MAX_SCHED = 16;
struct SchedArgs {
    int index[MAX_SCHED];
    int64_t t_total[MAX_SCHED];
    float warmup[MAX_SCHED];
    int n;
};
__global__ void sched_multi_kernel(int64_t* steps, float* lr_scale, SchedArgs a) {
    const int i = threadIdx.x;
    if (blockIdx.x != 0 || i >= a.n) return;
    const int k = a.index[i];
    const int64_t s = steps[k];
    float sc = 1.f;
    if (a.t_total[i] > 0) {
        const float x = (float)((double)s / (double)a.t_total[i]);
        sc = x < a.warmup[i] ? x / a.warmup[i] : fmaxf((x - 1.f) / (a.warmup[i] - 1.f), 0.f);
    }
    lr_scale[k] = sc;
    steps[k] = s + 1;
}

__global__ void rng_advance_kernel(uint64_t* rng, uint64_t by) {
    if (threadIdx.x == 0 && blockIdx.x == 0) rng[1] += by;
}

__global__ __launch_bounds__(NT) void cast_bf16_kernel(const float* __restrict__ x, bf16* out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT)
        out[i] = __float2bfloat16(x[i]);
}

__global__ __launch_bounds__(NT) void dropout_mask_kernel(float* out, int64_t n, float p, const uint64_t* rng, uint32_t sid) {
    const float ik = p > 0.f ? 1.f / (1.f - p) : 1.f;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT)
        out[i] = p > 0.f ? dropout_scale(p, ik, rng[0], rng[1], sid, (uint64_t)i) : 1.f;
}

__global__ __launch_bounds__(NT) void normal_kernel(float* out, int64_t n, const uint64_t* rng, uint32_t sid) {
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT)
        out[i] = philox_normal(rng[0], rng[1], sid, (uint64_t)i);
}

template <typename T>
int symkl(const void* x, const void* y, float* loss, const float* gout, void* dx, void* dy, int rows, int W, float coef,
          int accumulate, float* ws, hipStream_t st) {
    XGGM_REQUIRE(x && y && rows > 0 && W > 0, "xggm_symkl: bad arguments");
    XGGM_REQUIRE(!loss || ws, "xggm_symkl: the loss sum needs its workspace (XGGM_SUM_WS_FLOATS floats, ws[0] == 0)");
    XGGM_REQUIRE(W <= 64 * 16, "xggm_symkl: row width %d > 1024", W);
    XGGM_REQUIRE(loss || dx || dy, "xggm_symkl: nothing to compute");
    // the loss sum ends in ordered_grid_sum (a release fence and a ticket per workgroup): few, fat workgroups there
    const int grid = std::min(ceil_div(rows, 4), loss ? LOSS_GRID : 1024);
    const int nv = ceil_div(W, 64);
#define SYMKL_LAUNCH(NV)                                                                                               \
    hipLaunchKernelGGL((symkl_kernel<T, NV>), dim3(grid), dim3(NT), 0, st, (const T*)x, (const T*)y, loss, gout, (T*)dx, \
                       (T*)dy, rows, W, coef, accumulate, ws)
    if (nv <= 1) SYMKL_LAUNCH(1);
    else if (nv <= 2) SYMKL_LAUNCH(2);
    else if (nv <= 4) SYMKL_LAUNCH(4);
    else if (nv <= 8) SYMKL_LAUNCH(8);
    else if (nv <= 12) SYMKL_LAUNCH(12);
    else SYMKL_LAUNCH(16);
#undef SYMKL_LAUNCH
    return xggm_check_launch("xggm_symkl");
}

}  // namespace

#define LOSS_API(SUF, T)                                                                                                   \
    extern "C" int xggm_dsm_loss_fwd_##SUF(const void* s, const float* g, float* loss, int64_t n, float coef, float* ws,  \
                                           hipStream_t st) {                                                              \
        XGGM_REQUIRE(s && g && loss && ws && n > 0, "xggm_dsm_loss_fwd: bad arguments");                                   \
        hipLaunchKernelGGL((dsm_fwd_kernel<T>), dim3(grid1d(n / 4 + 1, LOSS_GRID)), dim3(NT), 0, st, (const T*)s, g, loss, n, coef, ws); \
        return xggm_check_launch("xggm_dsm_loss_fwd");                                                                    \
    }                                                                                                                      \
    extern "C" int xggm_dsm_loss_bwd_##SUF(const void* s, const float* g, const float* gout, void* ds, int64_t n,         \
                                           float coef, hipStream_t st) {                                                  \
        XGGM_REQUIRE(s && g && ds && n > 0, "xggm_dsm_loss_bwd: bad arguments");                                           \
        hipLaunchKernelGGL((dsm_bwd_kernel<T>), dim3(grid1d(n)), dim3(NT), 0, st, (const T*)s, g, gout, (T*)ds, n, coef); \
        return xggm_check_launch("xggm_dsm_loss_bwd");                                                                    \
    }                                                                                                                      \
    extern "C" int xggm_symkl_##SUF(const void* x, const void* y, float* loss, const float* gout, void* dx, void* dy,     \
                                    int rows, int W, float coef, int accumulate, float* ws, hipStream_t st) {             \
        return symkl<T>(x, y, loss, gout, dx, dy, rows, W, coef, accumulate, ws, st);                                     \
    }                                                                                                                      \
    extern "C" int xggm_bce_bwd_##SUF(const float* l, const float* t, const float* gout, void* dl, int64_t n, float coef, \
                                      hipStream_t st) {                                                                   \
        XGGM_REQUIRE(l && t && dl && n > 0, "xggm_bce_bwd: bad arguments");                                                \
        hipLaunchKernelGGL((bce_bwd_kernel<T>), dim3(grid1d(n)), dim3(NT), 0, st, l, t, gout, (T*)dl, n, coef);           \
        return xggm_check_launch("xggm_bce_bwd");                                                                         \
    }

LOSS_API(f32, float)
LOSS_API(bf16, bf16)

extern "C" int xggm_bce_fwd(const float* l, const float* t, float* loss, int64_t n, float coef, float* ws, hipStream_t st) {
    XGGM_REQUIRE(l && t && loss && ws && n > 0, "xggm_bce_fwd: bad arguments");
    hipLaunchKernelGGL(bce_fwd_kernel, dim3(grid1d(n, LOSS_GRID)), dim3(NT), 0, st, l, t, loss, n, coef, ws);
    return xggm_check_launch("xggm_bce_fwd");
}

extern "C" int xggm_sqnorm_f32(const float* g, int64_t n, float* out, float* ws, hipStream_t st) {
    XGGM_REQUIRE(g && out && ws && n > 0, "xggm_sqnorm_f32: bad arguments");
    XGGM_REQUIRE(reinterpret_cast<uintptr_t>(g) % 16 == 0, "xggm_sqnorm_f32: pointer must be 16-byte aligned");
    // at least 8 float4 per thread: the per-workgroup atomics all hit ONE address and serialise
    const int nblk = grid1d(n / 32 + 1, 4096);
    hipLaunchKernelGGL(sqnorm_kernel, dim3(nblk), dim3(NT), 0, st, g, n, out, ws);
    hipLaunchKernelGGL(sqnorm_finish_kernel, dim3(1), dim3(NT), 0, st, ws, nblk, out);
    return xggm_check_launch("xggm_sqnorm_f32");
}

namespace {
int launch_adam(const AdamArgs& a, bool g_bf16, hipStream_t st) {
    // measured on MI355X (tools/bench_adam.py, 110 M parameters): 4.7 TB/s with plain accesses and a 4096-block
    // grid-stride loop, 6.3 TB/s with non-temporal g/m/v and one or two float4 quadruples per thread
    const dim3 grid(grid1d(a.n / 8 + 1, 65536));
    if (g_bf16) hipLaunchKernelGGL((bertadam_kernel<2, true, true>), grid, dim3(NT), 0, st, a);
    else hipLaunchKernelGGL((bertadam_kernel<2, true, false>), grid, dim3(NT), 0, st, a);
    return xggm_check_launch("xggm_bertadam");
}
}  // namespace

extern "C" int xggm_bertadam_f32(float* p, const float* g, float* m, float* v, void* shadow_bf16, int64_t n,
                                 const float* sqnorm, float max_norm, float lr, const float* lr_scale, float b1, float b2,
                                 float eps, float weight_decay, hipStream_t st) {
    XGGM_REQUIRE(p && g && m && v && n > 0, "xggm_bertadam_f32: bad arguments");
    XGGM_REQUIRE((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                  reinterpret_cast<uintptr_t>(v)) % 16 == 0,
                 "xggm_bertadam_f32: pointers must be 16-byte aligned");
    XGGM_REQUIRE(!shadow_bf16 || reinterpret_cast<uintptr_t>(shadow_bf16) % 8 == 0, "xggm_bertadam_f32: shadow misaligned");
    AdamArgs a{p, g, m, v, (bf16*)shadow_bf16, n, sqnorm, max_norm, lr, nullptr, lr_scale, b1, b2, eps, weight_decay,
               nullptr, nullptr, nullptr, nullptr, 0, 1.f, 1};
    return launch_adam(a, false, st);
}

extern "C" int xggm_sqnorm_multi_f32(const float* base, const int64_t* offsets, const int64_t* lengths, int n, float* out,
                                     float* norm, float* ws, int overwrite, int square, float mul, hipStream_t st) {
    XGGM_REQUIRE(base && offsets && lengths && out && ws && n >= 0 && n <= MAX_SPANS,
                 "xggm_sqnorm_multi_f32: bad arguments (n = %d, at most %d ranges)", n, MAX_SPANS);
    XGGM_REQUIRE(reinterpret_cast<uintptr_t>(base) % 16 == 0, "xggm_sqnorm_multi_f32: base must be 16-byte aligned");
    Spans sp;
    sp.n = n;
    int64_t total = 0;
    for (int i = 0; i < n; ++i) {
        XGGM_REQUIRE(offsets[i] >= 0 && lengths[i] > 0 && offsets[i] % 4 == 0, "xggm_sqnorm_multi_f32: range %d (offset %lld, "
                     "length %lld) must be non-empty and start on a multiple of 4", i, (long long)offsets[i], (long long)lengths[i]);
        sp.off[i] = offsets[i];
        sp.len[i] = lengths[i];
        total += lengths[i];
    }
    // 4096 workgroups in all (the partials fit XGGM_SQNORM_WS_FLOATS), at least 8 float4 per thread, at least one each
    int nblk = 0;
    for (int i = 0; i < n; ++i) {
        sp.blk0[i] = nblk;
        const int64_t want = std::max<int64_t>(1, std::min<int64_t>(ceil_div64(lengths[i], (int64_t)NT * 32), (4096 - n) * lengths[i] / total + 1));
        nblk += (int)want;
    }
    sp.blk0[n] = nblk;
    XGGM_REQUIRE(nblk <= 4100, "xggm_sqnorm_multi_f32: internal: %d partials", nblk);
    if (nblk > 0) {
        if (square) hipLaunchKernelGGL(sqnorm_multi_kernel<true>, dim3(nblk), dim3(NT), 0, st, base, sp, ws);
        else hipLaunchKernelGGL(sqnorm_multi_kernel<false>, dim3(nblk), dim3(NT), 0, st, base, sp, ws);
    }
    hipLaunchKernelGGL(sqnorm_multi_finish_kernel, dim3(1), dim3(NT), 0, st, ws, nblk, out, norm, overwrite, mul);
    return xggm_check_launch("xggm_sqnorm_multi_f32");
}

extern "C" int xggm_bertadam_ex(const xggm_adam_args* x, hipStream_t st) {
    XGGM_REQUIRE(x && x->p && x->g && x->m && x->v && x->n > 0, "xggm_bertadam_ex: bad arguments");
    XGGM_REQUIRE((reinterpret_cast<uintptr_t>(x->p) | reinterpret_cast<uintptr_t>(x->m) | reinterpret_cast<uintptr_t>(x->v)) % 16 == 0 &&
                     reinterpret_cast<uintptr_t>(x->g) % (x->g_bf16 ? 8 : 16) == 0,
                 "xggm_bertadam_ex: pointers must be 16-byte aligned (bf16 gradients: 8)");
    XGGM_REQUIRE(!x->shadow_bf16 || reinterpret_cast<uintptr_t>(x->shadow_bf16) % 8 == 0, "xggm_bertadam_ex: shadow misaligned");
    XGGM_REQUIRE(!x->shadow8 || (x->w8_id && x->w8_qscale && x->w8_amax && x->elem0 % 256 == 0 && x->n % 4 == 0 &&
                                 reinterpret_cast<uintptr_t>(x->shadow8) % 4 == 0),
                 "xggm_bertadam_ex: the e4m3 copy needs the chunk table, the scale table and a range that starts on a "
                 "256-element chunk of the arena");
    AdamArgs a{x->p, x->g, x->m, x->v, (bf16*)x->shadow_bf16, x->n, x->sqnorm, x->max_norm, x->lr, x->lr_dev, x->lr_scale,
               x->b1, x->b2, x->eps, x->weight_decay, (unsigned char*)x->shadow8, x->w8_id, x->w8_qscale, x->w8_amax, x->elem0,
               x->g_scale > 0.f ? x->g_scale : 1.f, x->w8_amax_slots > 1 ? x->w8_amax_slots : 1};
    return launch_adam(a, x->g_bf16 != 0, st);
}

namespace {
int adam_args_of(const xggm_adam_args* x, AdamArgs& a) {
    XGGM_REQUIRE(x && x->p && x->g && x->m && x->v && x->n > 0, "xggm_bertadam: bad arguments");
    XGGM_REQUIRE((reinterpret_cast<uintptr_t>(x->p) | reinterpret_cast<uintptr_t>(x->m) | reinterpret_cast<uintptr_t>(x->v)) % 16 == 0 &&
                     reinterpret_cast<uintptr_t>(x->g) % (x->g_bf16 ? 8 : 16) == 0,
                 "xggm_bertadam: pointers must be 16-byte aligned (bf16 gradients: 8)");
    XGGM_REQUIRE(!x->shadow_bf16 || reinterpret_cast<uintptr_t>(x->shadow_bf16) % 8 == 0, "xggm_bertadam: shadow misaligned");
    XGGM_REQUIRE(!x->shadow8 || (x->w8_id && x->w8_qscale && x->w8_amax && x->elem0 % 256 == 0 && x->n % 4 == 0 &&
                                 reinterpret_cast<uintptr_t>(x->shadow8) % 4 == 0),
                 "xggm_bertadam: the e4m3 copy needs the chunk table, the scale table and a range that starts on a "
                 "256-element chunk of the arena");
    a = AdamArgs{x->p, x->g, x->m, x->v, (bf16*)x->shadow_bf16, x->n, x->sqnorm, x->max_norm, x->lr, x->lr_dev, x->lr_scale,
                 x->b1, x->b2, x->eps, x->weight_decay, (unsigned char*)x->shadow8, x->w8_id, x->w8_qscale, x->w8_amax, x->elem0,
                 x->g_scale > 0.f ? x->g_scale : 1.f, x->w8_amax_slots > 1 ? x->w8_amax_slots : 1};
    return XGGM_OK;
}
}  // namespace

extern "C" int xggm_bertadam_multi(const xggm_adam_args* args, int n, hipStream_t st) {
    XGGM_REQUIRE(args && n > 0, "xggm_bertadam_multi: no spans");
    for (int i0 = 0; i0 < n; i0 += ADAM_MULTI) {
        AdamMulti am;
        am.n = std::min(ADAM_MULTI, n - i0);
        int nblk = 0;
        for (int i = 0; i < am.n; ++i) {
            if (int e = adam_args_of(args + i0 + i, am.a[i])) return e;
            XGGM_REQUIRE(args[i0 + i].g_bf16 == args[i0].g_bf16, "xggm_bertadam_multi: the spans of one call share the gradient type");
            am.blk0[i] = nblk;
            nblk += grid1d(am.a[i].n / 8 + 1, 65536);
        }
        am.blk0[am.n] = nblk;
        if (args[i0].g_bf16) hipLaunchKernelGGL((bertadam_multi_kernel<2, true, true>), dim3(nblk), dim3(NT), 0, st, am);
        else hipLaunchKernelGGL((bertadam_multi_kernel<2, true, false>), dim3(nblk), dim3(NT), 0, st, am);
        if (int e = xggm_check_launch("xggm_bertadam_multi")) return e;
    }
    return XGGM_OK;
}

extern "C" int xggm_sqnorm_bf16(const void* g, int64_t n, float* out, float* ws, hipStream_t st) {
    XGGM_REQUIRE(g && out && ws && n > 0, "xggm_sqnorm_bf16: bad arguments");
    XGGM_REQUIRE(reinterpret_cast<uintptr_t>(g) % 16 == 0, "xggm_sqnorm_bf16: pointer must be 16-byte aligned");
    const int nblk = grid1d(n / 64 + 1, 4096);
    hipLaunchKernelGGL(sqnorm_bf16_kernel, dim3(nblk), dim3(NT), 0, st, (const bf16*)g, n, ws);
    hipLaunchKernelGGL(sqnorm_finish_kernel, dim3(1), dim3(NT), 0, st, ws, nblk, out);
    return xggm_check_launch("xggm_sqnorm_bf16");
}

extern "C" int xggm_fp8_scale_update(float* amax, float* hist, float* qscale, float* dscale, int64_t* pos, int n,
                                     int hist_len, float margin, int shrink, int bump, int amax_slots, hipStream_t st) {
    const int slots = amax_slots > 1 ? amax_slots : 1;
    XGGM_REQUIRE(slots <= 64 && (slots & (slots - 1)) == 0, "xggm_fp8_scale_update: amax_slots %d must be a power of two <= 64", slots);
    XGGM_REQUIRE(amax && hist && qscale && dscale && pos && n > 0 && n <= 1024 && hist_len > 0 && hist_len <= 64 && margin >= 1.f,
                 "xggm_fp8_scale_update: bad arguments (n = %d <= 1024 entries, history %d <= 64, margin %g >= 1)", n,
                 hist_len, (double)margin);
    hipLaunchKernelGGL(fp8_scale_update_kernel, dim3(1), dim3(1024), 0, st, amax, hist, qscale, dscale, pos, n, hist_len,
                       margin, shrink, bump, slots);
    return xggm_check_launch("xggm_fp8_scale_update");
}

extern "C" int xggm_sched_step(int64_t* step, float* lr_scale, int64_t t_total, float warmup, hipStream_t st) {
    XGGM_REQUIRE(step && lr_scale, "xggm_sched_step: null pointer");
    hipLaunchKernelGGL(sched_kernel, dim3(1), dim3(64), 0, st, step, lr_scale, t_total, warmup);
    return xggm_check_launch("xggm_sched_step");
}

extern "C" int xggm_sched_step_multi(int64_t* steps, float* lr_scale, const int* index, const int64_t* t_total,
                                     const float* warmup, int n, hipStream_t st) {
    XGGM_REQUIRE(steps && lr_scale && index && t_total && warmup && n > 0 && n <= MAX_SCHED,
                 "xggm_sched_step_multi: bad arguments (n = %d, at most %d counters per call)", n, MAX_SCHED);
    SchedArgs a;
    a.n = n;
    for (int i = 0; i < n; ++i) {
        XGGM_REQUIRE(index[i] >= 0, "xggm_sched_step_multi: negative index");
        for (int j = 0; j < i; ++j) XGGM_REQUIRE(index[j] != index[i], "xggm_sched_step_multi: counter %d listed twice", index[i]);
        a.index[i] = index[i];
        a.t_total[i] = t_total[i];
        a.warmup[i] = warmup[i];
    }
    hipLaunchKernelGGL(sched_multi_kernel, dim3(1), dim3(64), 0, st, steps, lr_scale, a);
    return xggm_check_launch("xggm_sched_step_multi");
}

namespace {
__global__ __launch_bounds__(NT) void clip_norm_finish_kernel(const float* __restrict__ ws, int nblk, float* out, float* norm, float mul,
                                                              int64_t* steps, float* lr_scale, SchedArgs sa, uint64_t* rng,
                                                              uint64_t rng_by, int accumulate) {
    float t = 0.f;
    for (int b = threadIdx.x; b < nblk; b += NT) t += ws[b];
    t = block_sum(t);
    if (threadIdx.x == 0) {
        const float s = ((accumulate ? *out : 0.f) + t) * mul;
        *out = s;
        if (norm) *norm = sqrtf(s);
        if (rng) rng[1] += rng_by;
    }
    const int i = threadIdx.x;
    if (i < sa.n) {  // sched_multi_kernel's step
        const int k = sa.index[i];
        const int64_t s = steps[k];
        float sc = 1.f;
        if (sa.t_total[i] > 0) {
            const float x = (float)((double)s / (double)sa.t_total[i]);
            sc = x < sa.warmup[i] ? x / sa.warmup[i] : fmaxf((x - 1.f) / (sa.warmup[i] - 1.f), 0.f);
        }
        lr_scale[k] = sc;
        steps[k] = s + 1;
    }
}
}  // namespace

namespace {
struct TailArgs {
    SchedArgs sa;
    int64_t* steps = nullptr;
    float* lr_scale = nullptr;
    uint64_t* rng = nullptr;
    uint64_t rng_by = 0;
};
int tail_args_of(const xggm_pass_tail* tail, TailArgs& ta) {
    ta.sa.n = 0;
    if (!tail) return XGGM_OK;
    XGGM_REQUIRE(tail->n >= 0 && tail->n <= MAX_SCHED && (tail->n == 0 || (tail->steps && tail->lr_scale && tail->index &&
                                                                         tail->t_total && tail->warmup)),
                 "xggm_clip_norm: bad schedule entries (n = %d, at most %d)", tail->n, MAX_SCHED);
    ta.sa.n = tail->n;
    for (int i = 0; i < tail->n; ++i) {
        XGGM_REQUIRE(tail->index[i] >= 0, "xggm_clip_norm: negative schedule index");
        for (int j = 0; j < i; ++j)
            XGGM_REQUIRE(tail->index[j] != tail->index[i], "xggm_clip_norm: counter %d listed twice", tail->index[i]);
        ta.sa.index[i] = tail->index[i];
        ta.sa.t_total[i] = tail->t_total[i];
        ta.sa.warmup[i] = tail->warmup[i];
    }
    ta.steps = tail->steps;
    ta.lr_scale = tail->lr_scale;
    ta.rng = tail->rng;
    ta.rng_by = tail->rng_by;
    return XGGM_OK;
}
}  // namespace

extern "C" int xggm_clip_norm_f32(const float* g, const int64_t* offsets, const int64_t* lengths, int n, const float* slots,
                                  const int64_t* slot_offsets, const int64_t* slot_lengths, int n_slots, float* out, float* norm,
                                  float* ws, float mul, const xggm_pass_tail* tail, hipStream_t st) {
    XGGM_REQUIRE(out && ws && n >= 0 && n_slots >= 0 && n + n_slots <= MAX_NORM_SPANS && (n == 0 || (g && offsets && lengths)) &&
                     (n_slots == 0 || (slots && slot_offsets && slot_lengths)),
                 "xggm_clip_norm_f32: bad arguments (%d + %d ranges, at most %d in all)", n, n_slots, MAX_NORM_SPANS);
    XGGM_REQUIRE((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(slots)) % 16 == 0,
                 "xggm_clip_norm_f32: buffers must be 16-byte aligned");
    NormSpans sp;
    sp.n = n + n_slots;
    sp.n_sq = n;
    int64_t total = 0;
    for (int i = 0; i < sp.n; ++i) {
        const int64_t o = i < n ? offsets[i] : slot_offsets[i - n], l = i < n ? lengths[i] : slot_lengths[i - n];
        XGGM_REQUIRE(o >= 0 && l > 0 && o % 4 == 0, "xggm_clip_norm_f32: range %d (offset %lld, length %lld) must be non-empty and "
                     "start on a multiple of 4", i, (long long)o, (long long)l);
        sp.ptr[i] = (i < n ? g : slots) + o;
        sp.len[i] = l;
        total += l;
    }
    int nblk = 0;  // as xggm_sqnorm_multi_f32: <= 4096 + n partials, at least 8 float4 per thread where a range is long enough
    for (int i = 0; i < sp.n; ++i) {
        sp.blk0[i] = nblk;
        nblk += (int)std::max<int64_t>(1, std::min<int64_t>(ceil_div64(sp.len[i], (int64_t)NT * 32), (4096 - sp.n) * sp.len[i] / total + 1));
    }
    sp.blk0[sp.n] = nblk;
    XGGM_REQUIRE(nblk <= 4100, "xggm_clip_norm_f32: internal: %d partials", nblk);
    TailArgs ta;
    if (int e = tail_args_of(tail, ta)) return e;
    if (nblk > 0) hipLaunchKernelGGL(clip_norm_kernel, dim3(nblk), dim3(NT), 0, st, sp, ws);
    hipLaunchKernelGGL(clip_norm_finish_kernel, dim3(1), dim3(NT), 0, st, ws, nblk, out, norm, mul, ta.steps, ta.lr_scale, ta.sa, ta.rng, ta.rng_by, 0);
    return xggm_check_launch("xggm_clip_norm_f32");
}

extern "C" int xggm_clip_norm_bf16(const void* g, const int64_t* offsets, const int64_t* lengths, int n, float* out, float* norm,
                                   float* ws, int accumulate, float mul, const xggm_pass_tail* tail, hipStream_t st) {
    XGGM_REQUIRE(out && ws && n >= 0 && n <= MAX_NORM_SPANS && (n == 0 || (g && offsets && lengths)),
                 "xggm_clip_norm_bf16: bad arguments (%d ranges, at most %d)", n, MAX_NORM_SPANS);
    XGGM_REQUIRE(reinterpret_cast<uintptr_t>(g) % 16 == 0, "xggm_clip_norm_bf16: the buffer must be 16-byte aligned");
    NormSpans16 sp;
    sp.n = n;
    int64_t total = 0;
    for (int i = 0; i < n; ++i) {
        XGGM_REQUIRE(offsets[i] >= 0 && lengths[i] > 0 && offsets[i] % 8 == 0, "xggm_clip_norm_bf16: range %d (offset %lld, length "
                     "%lld) must be non-empty and start on a multiple of 8", i, (long long)offsets[i], (long long)lengths[i]);
        sp.ptr[i] = reinterpret_cast<const bf16*>(g) + offsets[i];
        sp.len[i] = lengths[i];
        total += lengths[i];
    }
    int nblk = 0;
    for (int i = 0; i < n; ++i) {
        sp.blk0[i] = nblk;
        nblk += (int)std::max<int64_t>(1, std::min<int64_t>(ceil_div64(sp.len[i], (int64_t)NT * 64), (4096 - n) * sp.len[i] / total + 1));
    }
    sp.blk0[n] = nblk;
    XGGM_REQUIRE(nblk <= 4100, "xggm_clip_norm_bf16: internal: %d partials", nblk);
    TailArgs ta;
    if (int e = tail_args_of(tail, ta)) return e;
    if (nblk > 0) hipLaunchKernelGGL(clip_norm_bf16_kernel, dim3(nblk), dim3(NT), 0, st, sp, ws);
    hipLaunchKernelGGL(clip_norm_finish_kernel, dim3(1), dim3(NT), 0, st, ws, nblk, out, norm, mul, ta.steps, ta.lr_scale, ta.sa, ta.rng, ta.rng_by,
                       accumulate);
    return xggm_check_launch("xggm_clip_norm_bf16");
}

extern "C" int xggm_rng_advance(uint64_t* rng, uint64_t by, hipStream_t st) {
    XGGM_REQUIRE(rng, "xggm_rng_advance: null pointer");
    hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(64), 0, st, rng, by);
    return xggm_check_launch("xggm_rng_advance");
}

extern "C" int xggm_cast_f32_to_bf16(const float* x, void* out, int64_t n, hipStream_t st) {
    XGGM_REQUIRE(x && out && n > 0, "xggm_cast_f32_to_bf16: bad arguments");
    hipLaunchKernelGGL(cast_bf16_kernel, dim3(grid1d(n, 4096)), dim3(NT), 0, st, x, (bf16*)out, n);
    return xggm_check_launch("xggm_cast_f32_to_bf16");
}

extern "C" int xggm_dropout_mask(float* out, int64_t n, float p, const uint64_t* rng, uint32_t sid, hipStream_t st) {
    XGGM_REQUIRE(out && n > 0 && p >= 0.f && p < 1.f && (p == 0.f || rng), "xggm_dropout_mask: bad arguments");
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid1d(n)), dim3(NT), 0, st, out, n, p, rng, sid);
    return xggm_check_launch("xggm_dropout_mask");
}

extern "C" int xggm_normal(float* out, int64_t n, const uint64_t* rng, uint32_t sid, hipStream_t st) {
    XGGM_REQUIRE(out && n > 0 && rng, "xggm_normal: bad arguments");
    hipLaunchKernelGGL(normal_kernel, dim3(grid1d(n)), dim3(NT), 0, st, out, n, rng, sid);
    return xggm_check_launch("xggm_normal");
}
