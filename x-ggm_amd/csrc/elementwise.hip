// Small element-wise kernels that sit between the fused ops (HBM-bound, 16-byte accesses
// where the element count allows, grid-stride).
#include "common.h"
#include "xggm.h"

namespace {
constexpr int NT = 256;
inline int grid1d(int64_t n) { return (int)std::min<int64_t>(ceil_div64(n, NT), 2048); }

template <typename T>
__global__ __launch_bounds__(NT) void scale_kernel(const T* __restrict__ x, T* out, int64_t n, float scale,
                                                   const float* scale_ptr) {
    if (scale_ptr) scale *= (1.0f + *scale_ptr);
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT)
        out[i] = from_f32<T>(scale * to_f32(x[i]));
}
template <typename T>
__global__ __launch_bounds__(NT) void sigmoid_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, T* out,
                                                         int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT)
        out[i] = from_f32<T>(dy[i] * y[i] * (1.0f - y[i]));
}
template <typename T>
__global__ __launch_bounds__(NT) void tanh_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y, T* out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const float t = to_f32(y[i]);
        out[i] = from_f32<T>(to_f32(dy[i]) * (1.0f - t * t));
    }
}
template <typename T> __global__ __launch_bounds__(NT) void cast_kernel(const float* __restrict__ x, T* out, int64_t n) {
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n4; i += (int64_t)gridDim.x * NT) {
        float v[4];
        load4(x + 4 * i, v);
        store4(out + 4 * i, v);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) out[(n4 << 2) + threadIdx.x] = from_f32<T>(x[(n4 << 2) + threadIdx.x]);
}
}  // namespace

#define EW_API(SUF, T)                                                                                                   \
    extern "C" int xggm_scale_##SUF(const void* x, void* out, int64_t n, float scale, const float* scale_ptr,          \
                                    hipStream_t st) {                                                                   \
        XGGM_REQUIRE(x && out && n > 0, "xggm_scale: bad arguments");                                                    \
        hipLaunchKernelGGL((scale_kernel<T>), dim3(grid1d(n)), dim3(NT), 0, st, (const T*)x, (T*)out, n, scale,         \
                           scale_ptr);                                                                                  \
        return xggm_check_launch("xggm_scale");                                                                         \
    }                                                                                                                    \
    extern "C" int xggm_sigmoid_bwd_##SUF(const float* dy, const float* y, void* out, int64_t n, hipStream_t st) {      \
        XGGM_REQUIRE(dy && y && out && n > 0, "xggm_sigmoid_bwd: bad arguments");                                        \
        hipLaunchKernelGGL((sigmoid_bwd_kernel<T>), dim3(grid1d(n)), dim3(NT), 0, st, dy, y, (T*)out, n);               \
        return xggm_check_launch("xggm_sigmoid_bwd");                                                                   \
    }                                                                                                                    \
    extern "C" int xggm_tanh_bwd_##SUF(const void* dy, const void* y, void* out, int64_t n, hipStream_t st) {           \
        XGGM_REQUIRE(dy && y && out && n > 0, "xggm_tanh_bwd: bad arguments");                                           \
        hipLaunchKernelGGL((tanh_bwd_kernel<T>), dim3(grid1d(n)), dim3(NT), 0, st, (const T*)dy, (const T*)y, (T*)out,  \
                           n);                                                                                          \
        return xggm_check_launch("xggm_tanh_bwd");                                                                      \
    }                                                                                                                    \
    extern "C" int xggm_cast_from_f32_##SUF(const float* x, void* out, int64_t n, hipStream_t st) {                     \
        XGGM_REQUIRE(x && out && n > 0, "xggm_cast_from_f32: bad arguments");                                            \
        XGGM_REQUIRE(reinterpret_cast<uintptr_t>(x) % 16 == 0 && reinterpret_cast<uintptr_t>(out) % 8 == 0,             \
                     "xggm_cast_from_f32: misaligned pointers");                                                        \
        hipLaunchKernelGGL((cast_kernel<T>), dim3(grid1d(n / 4 + 1)), dim3(NT), 0, st, x, (T*)out, n);                  \
        return xggm_check_launch("xggm_cast_from_f32");                                                                 \
    }

namespace {
constexpr int MAX_RANGES = 16;
struct Ranges {
    int64_t off[MAX_RANGES], len4[MAX_RANGES];  // element offset, length in float4
    int64_t start[MAX_RANGES + 1];              // prefix sum of len4
    int n;
};
__global__ __launch_bounds__(NT) void zero_ranges_kernel(float* base, Ranges r) {
    const int64_t total = r.start[r.n];
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        int k = 0;
#pragma unroll
        for (int j = 1; j < MAX_RANGES; ++j)
            if (j < r.n && i >= r.start[j]) k = j;
        typedef float __attribute__((ext_vector_type(4))) f4;
        __builtin_nontemporal_store((f4){0.f, 0.f, 0.f, 0.f}, reinterpret_cast<f4*>(base + r.off[k]) + (i - r.start[k]));
    }
}
}  // namespace

extern "C" int xggm_zero_ranges_f32(float* base, const int64_t* offsets, const int64_t* lengths, int n, hipStream_t st) {
    XGGM_REQUIRE(base && offsets && lengths && n > 0 && n <= MAX_RANGES, "xggm_zero_ranges_f32: bad arguments (n = %d)", n);
    XGGM_REQUIRE(reinterpret_cast<uintptr_t>(base) % 16 == 0, "xggm_zero_ranges_f32: base must be 16-byte aligned");
    Ranges r;
    r.n = n;
    int64_t tot = 0;
    for (int i = 0; i < n; ++i) {
        XGGM_REQUIRE(offsets[i] >= 0 && lengths[i] >= 0 && offsets[i] % 4 == 0 && lengths[i] % 4 == 0,
                     "xggm_zero_ranges_f32: range %d is not float4-aligned", i);
        r.off[i] = offsets[i];
        r.len4[i] = lengths[i] / 4;
        r.start[i] = tot;
        tot += r.len4[i];
    }
    r.start[n] = tot;
    if (tot == 0) return XGGM_OK;
    hipLaunchKernelGGL(zero_ranges_kernel, dim3((int)std::min<int64_t>(ceil_div64(tot, NT), 4096)), dim3(NT), 0, st, base, r);
    return xggm_check_launch("xggm_zero_ranges_f32");
}

namespace {
// the same fill plus the LISTED rows of one [R, H] table (row ids and their count live on the device: the list the last
// xggm_embed_bwd_listed_* left) -- the word table's gradient is 94 MB of which a pass makes <= B * T rows non-zero
__global__ __launch_bounds__(NT) void zero_ranges_rows_kernel(float* base, Ranges r, int range_blocks, float* table,
                                                              const int64_t* __restrict__ ids, const int* __restrict__ n_ptr, int cap,
                                                              int64_t R, int H) {
    typedef float __attribute__((ext_vector_type(4))) f4;
    if ((int)blockIdx.x < range_blocks) {
        const int64_t total = r.start[r.n];
        for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)range_blocks * NT) {
            int k = 0;
#pragma unroll
            for (int j = 1; j < MAX_RANGES; ++j)
                if (j < r.n && i >= r.start[j]) k = j;
            __builtin_nontemporal_store((f4){0.f, 0.f, 0.f, 0.f}, reinterpret_cast<f4*>(base + r.off[k]) + (i - r.start[k]));
        }
        return;
    }
    const int n = min(*n_ptr, cap), nb = gridDim.x - range_blocks;
    for (int i = blockIdx.x - range_blocks; i < n; i += nb) {
        const int64_t id = ids[i];
        if (id < 0 || id >= R) continue;  // (never: the forward has read this row)
        f4* row = reinterpret_cast<f4*>(table + id * H);
        for (int c = threadIdx.x; c < H / 4; c += NT) row[c] = (f4){0.f, 0.f, 0.f, 0.f};
    }
}
}  // namespace

extern "C" int xggm_zero_ranges_rows_f32(float* base, const int64_t* offsets, const int64_t* lengths, int n, float* table,
                                         const int64_t* row_ids, const int* row_n, int row_cap, int64_t R, int H, hipStream_t st) {
    XGGM_REQUIRE(base && n >= 0 && n <= MAX_RANGES && (n == 0 || (offsets && lengths)), "xggm_zero_ranges_rows_f32: bad ranges (n = %d)", n);
    XGGM_REQUIRE(table && row_ids && row_n && row_cap > 0 && R > 0 && H > 0 && H % 4 == 0,
                 "xggm_zero_ranges_rows_f32: bad row list (cap %d, table %lld x %d)", row_cap, (long long)R, H);
    XGGM_REQUIRE((reinterpret_cast<uintptr_t>(base) | reinterpret_cast<uintptr_t>(table)) % 16 == 0,
                 "xggm_zero_ranges_rows_f32: buffers must be 16-byte aligned");
    Ranges r;
    r.n = n;
    int64_t tot = 0;
    for (int i = 0; i < n; ++i) {
        XGGM_REQUIRE(offsets[i] >= 0 && lengths[i] >= 0 && offsets[i] % 4 == 0 && lengths[i] % 4 == 0,
                     "xggm_zero_ranges_rows_f32: range %d is not float4-aligned", i);
        r.off[i] = offsets[i];
        r.len4[i] = lengths[i] / 4;
        r.start[i] = tot;
        tot += r.len4[i];
    }
    r.start[n] = tot;
    const int range_blocks = (int)std::min<int64_t>(ceil_div64(tot, NT), 4096);
    const int row_blocks = std::min(row_cap, 1024);
    hipLaunchKernelGGL(zero_ranges_rows_kernel, dim3(range_blocks + row_blocks), dim3(NT), 0, st, base, r, range_blocks, table,
                       row_ids, row_n, row_cap, R, H);
    return xggm_check_launch("xggm_zero_ranges_rows_f32");
}

EW_API(f32, float)
EW_API(bf16, bf16)

// ---- scalar / input glue of the training pass, so that a captured pass holds no framework kernels
namespace {
// adj.triu(1) + adj.tril(-1): the diagonal of the input adjacency removed (src/vqa/vqacpv2.py:188)
__global__ __launch_bounds__(NT) void zero_diag_kernel(const float* __restrict__ in, float* out, int64_t total, int N) {
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int e = (int)(i % ((int64_t)N * N));
        out[i] = (e / N == e % N) ? 0.f : in[i];
    }
}
// (1 - mask) * -10000: the additive attention mask of LXRTModel.forward (src/lxrt/modeling.py:919-928)
__global__ __launch_bounds__(NT) void additive_mask_kernel(const int64_t* __restrict__ m, float* out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT)
        out[i] = (1.0f - (float)m[i]) * -10000.0f;
}
__global__ void add_scalars_kernel(const float* a, const float* b, const float* c, const float* d, float* out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *out = (a ? *a : 0.f) + (b ? *b : 0.f) + (c ? *c : 0.f) + (d ? *d : 0.f);
}
}  // namespace

// out = a + b (+ c + d): the sum autograd forms where a tensor feeds several consumers (xggm_amd.functional.FanOutFn),
// in fp32, rounded once; `out` may alias any input.
template <typename T>
__global__ __launch_bounds__(NT) void add_n_kernel(const T* a, const T* b, const T* c, const T* d, T* out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        float v = to_f32(a[i]) + to_f32(b[i]);
        if (c) v += to_f32(c[i]);
        if (d) v += to_f32(d[i]);
        out[i] = from_f32<T>(v);
    }
}

// dst [rows, ld] (bf16) = cast(src [rows, n]) with zeros in columns n .. ld - 1: a row stride the tuned GEMM path accepts
// for outputs whose width is not a multiple of 8 (2274 answers, 630 edges)
template <typename S>
__global__ __launch_bounds__(NT) void pad_rows_kernel(const S* __restrict__ src, bf16* __restrict__ dst, int64_t total, int n,
                                                      int ld) {
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int64_t r = i / ld;
        const int c = (int)(i - r * ld);
        dst[i] = c < n ? from_f32<bf16>(to_f32(src[r * n + c])) : from_f32<bf16>(0.f);
    }
}

// dst[i, :] = bf16(src[idx[i], :]) / dst[idx[i], :] = src[i, :]: the rows of the word-embedding gradient the ranks of a
// data-parallel group actually touched (xggm_amd.dist: the table is 23.4 M parameters, a step touches <= 640 rows per
// rank -- the exchange moves world x 640 rows instead of 30522).  One wave per row, 8 bytes per lane and step.
__global__ __launch_bounds__(NT) void gather_rows_kernel(const float* __restrict__ src, const int64_t* __restrict__ idx,
                                                         bf16* __restrict__ dst, int n, int H, int64_t V) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int i = blockIdx.x * (NT / 64) + wid; i < n; i += gridDim.x * (NT / 64)) {
        const int64_t r = idx[i];
        if (r < 0 || r >= V) continue;
        for (int c = lane * 4; c < H; c += 256) {
            float v[4];
            load4(src + r * H + c, v);
            store4(dst + (int64_t)i * H + c, v);
        }
    }
}
__global__ __launch_bounds__(NT) void scatter_rows_kernel(const bf16* __restrict__ src, const int64_t* __restrict__ idx,
                                                          bf16* __restrict__ dst, int n, int H, int64_t V) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int i = blockIdx.x * (NT / 64) + wid; i < n; i += gridDim.x * (NT / 64)) {
        const int64_t r = idx[i];
        if (r < 0 || r >= V) continue;
        for (int c = lane * 4; c < H; c += 256)  // duplicates of an index carry identical rows: plain stores
            *reinterpret_cast<bf16x4*>(dst + r * H + c) = *reinterpret_cast<const bf16x4*>(src + (int64_t)i * H + c);
    }
}

extern "C" int xggm_zero_diag_f32(const float* in, float* out, int B, int N, hipStream_t st) {
    XGGM_REQUIRE(in && out && B > 0 && N > 0, "xggm_zero_diag_f32: bad arguments");
    const int64_t total = (int64_t)B * N * N;
    hipLaunchKernelGGL(zero_diag_kernel, dim3(grid1d(total)), dim3(NT), 0, st, in, out, total, N);
    return xggm_check_launch("xggm_zero_diag_f32");
}

extern "C" int xggm_add_n_f32(const float* a, const float* b, const float* c, const float* d, float* out, int64_t n, hipStream_t st) {
    XGGM_REQUIRE(a && b && out && n > 0 && (c || !d), "xggm_add_n_f32: bad arguments");
    hipLaunchKernelGGL(add_n_kernel<float>, dim3(grid1d(n)), dim3(NT), 0, st, a, b, c, d, out, n);
    return xggm_check_launch("xggm_add_n_f32");
}
extern "C" int xggm_add_n_bf16(const void* a, const void* b, const void* c, const void* d, void* out, int64_t n, hipStream_t st) {
    XGGM_REQUIRE(a && b && out && n > 0 && (c || !d), "xggm_add_n_bf16: bad arguments");
    hipLaunchKernelGGL(add_n_kernel<bf16>, dim3(grid1d(n)), dim3(NT), 0, st, (const bf16*)a, (const bf16*)b, (const bf16*)c,
                       (const bf16*)d, (bf16*)out, n);
    return xggm_check_launch("xggm_add_n_bf16");
}
extern "C" int xggm_pad_rows_bf16(const void* src, int src_f32, void* dst, int rows, int n, int ld, hipStream_t st) {
    XGGM_REQUIRE(src && dst && rows > 0 && n > 0 && ld >= n, "xggm_pad_rows_bf16: bad arguments (rows %d, n %d, ld %d)", rows, n, ld);
    const int64_t total = (int64_t)rows * ld;
    if (src_f32) hipLaunchKernelGGL(pad_rows_kernel<float>, dim3(grid1d(total)), dim3(NT), 0, st, (const float*)src, (bf16*)dst, total, n, ld);
    else hipLaunchKernelGGL(pad_rows_kernel<bf16>, dim3(grid1d(total)), dim3(NT), 0, st, (const bf16*)src, (bf16*)dst, total, n, ld);
    return xggm_check_launch("xggm_pad_rows_bf16");
}

extern "C" int xggm_gather_rows_bf16(const float* src, const int64_t* idx, void* dst, int n, int H, int64_t V, hipStream_t st) {
    XGGM_REQUIRE(src && idx && dst && n > 0 && H > 0 && H % 4 == 0 && V > 0, "xggm_gather_rows_bf16: bad arguments");
    hipLaunchKernelGGL(gather_rows_kernel, dim3(std::min(ceil_div(n, NT / 64), 2048)), dim3(NT), 0, st, src, idx, (bf16*)dst, n, H, V);
    return xggm_check_launch("xggm_gather_rows_bf16");
}
extern "C" int xggm_scatter_rows_bf16(const void* src, const int64_t* idx, void* dst, int n, int H, int64_t V, hipStream_t st) {
    XGGM_REQUIRE(src && idx && dst && n > 0 && H > 0 && H % 4 == 0 && V > 0, "xggm_scatter_rows_bf16: bad arguments");
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(std::min(ceil_div(n, NT / 64), 2048)), dim3(NT), 0, st, (const bf16*)src, idx, (bf16*)dst, n, H, V);
    return xggm_check_launch("xggm_scatter_rows_bf16");
}

extern "C" int xggm_additive_mask(const int64_t* mask, float* out, int64_t n, hipStream_t st) {
    XGGM_REQUIRE(mask && out && n > 0, "xggm_additive_mask: bad arguments");
    hipLaunchKernelGGL(additive_mask_kernel, dim3(grid1d(n)), dim3(NT), 0, st, mask, out, n);
    return xggm_check_launch("xggm_additive_mask");
}

extern "C" int xggm_add_scalars_f32(const float* a, const float* b, const float* c, const float* d, float* out, hipStream_t st) {
    XGGM_REQUIRE(out, "xggm_add_scalars_f32: null output");
    hipLaunchKernelGGL(add_scalars_kernel, dim3(1), dim3(64), 0, st, a, b, c, d, out);
    return xggm_check_launch("xggm_add_scalars_f32");
}


// ---- fp8 operands of the mixed-precision forward (BASELINE config C5): y = e4m3(clamp(x * qscale, +-448)),
// OCP e4m3fn as gfx950's matrix cores read it, round-to-nearest-even (v_cvt_pk_fp8_f32).  Per-tensor scaling:
// `qscale` is a device scalar (the GEMM multiplies by its reciprocal), `amax` (optional) collects max |x| for
// the next step's scale ("delayed scaling": no extra pass over x).  8 elements per thread: one 16/32-byte load,
// one 8-byte store.
namespace {
template <typename T>
__global__ __launch_bounds__(NT) void quantize_fp8_kernel(const T* __restrict__ x, unsigned char* __restrict__ y, int64_t n,
                                                          const float* __restrict__ qscale, float* amax) {
    const float s = Q8(qscale).q;  // an uncalibrated entry (*qscale <= 0) quantises with 1
    float mx = 0.f;
    const int64_t n8 = n >> 3;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n8; i += (int64_t)gridDim.x * NT) {
        float lo[4], hi[4];
        load4(x + 8 * i, lo);
        load4(x + 8 * i + 4, hi);
        const float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        float q[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            mx = fmaxf(mx, fabsf(v[e]));
            q[e] = fminf(fmaxf(v[e] * s, -448.f), 448.f);
        }
        int w0 = __builtin_amdgcn_cvt_pk_fp8_f32(q[0], q[1], 0, false);
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(q[2], q[3], w0, true);
        int w1 = __builtin_amdgcn_cvt_pk_fp8_f32(q[4], q[5], 0, false);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(q[6], q[7], w1, true);
        *reinterpret_cast<int2*>(y + 8 * i) = make_int2(w0, w1);
    }
    if (amax) {
        mx = wave_max(mx);
        // |x| >= 0: the IEEE bit pattern orders like the value
        if ((threadIdx.x & 63) == 0 && mx > 0.f) atomicMax(reinterpret_cast<unsigned int*>(amax), __float_as_uint(mx));
    }
}
}  // namespace

#define QUANT_API(SUF, T)                                                                                               \
    extern "C" int xggm_quantize_fp8e4m3_##SUF(const void* x, void* y, int64_t n, const float* qscale, float* amax,    \
                                               hipStream_t st) {                                                        \
        XGGM_REQUIRE(x && y && n > 0 && n % 8 == 0, "xggm_quantize_fp8e4m3: bad arguments (n = %lld must be a multiple of 8)", \
                     (long long)n);                                                                                     \
        XGGM_REQUIRE(reinterpret_cast<uintptr_t>(x) % 16 == 0 && reinterpret_cast<uintptr_t>(y) % 8 == 0,               \
                     "xggm_quantize_fp8e4m3: misaligned pointers");                                                     \
        hipLaunchKernelGGL((quantize_fp8_kernel<T>), dim3(grid1d(n / 8)), dim3(NT), 0, st, (const T*)x,                 \
                           (unsigned char*)y, n, qscale, amax);                                                         \
        return xggm_check_launch("xggm_quantize_fp8e4m3");                                                              \
    }
QUANT_API(f32, float)
QUANT_API(bf16, bf16)
