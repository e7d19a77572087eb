// Shared device/host helpers for the xggm HIP kernels (gfx950 / CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <math.h>
#include <algorithm>
#include <type_traits>

#define XGGM_OK 0
#define XGGM_ERR_ARG 1
#define XGGM_ERR_LAUNCH 2

typedef __hip_bfloat16 bf16;

// ---------------------------------------------------------------- error reporting
void xggm_set_error(const char* fmt, ...);
int xggm_check_launch(const char* what);

#define XGGM_REQUIRE(cond, ...)            \
    do {                                   \
        if (!(cond)) {                     \
            xggm_set_error(__VA_ARGS__);   \
            return XGGM_ERR_ARG;           \
        }                                  \
    } while (0)

// ---------------------------------------------------------------- dtype helpers
__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16 x) { return __bfloat162float(x); }
template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float x) { return __float2bfloat16(x); }

// value as the storage type would hold it (bf16 round trip), used where forward and
// backward must see the same rounded activation
template <typename T> __device__ __forceinline__ float round_to(float x) { return to_f32(from_f32<T>(x)); }

// vector of 4 elements of T <-> 4 floats
struct __attribute__((aligned(8))) bf16x4 { bf16 v[4]; };
__device__ __forceinline__ void load4(const float* p, float (&o)[4]) {
    float4 t = *reinterpret_cast<const float4*>(p);
    o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
}
__device__ __forceinline__ void load4(const bf16* p, float (&o)[4]) {
    bf16x4 t = *reinterpret_cast<const bf16x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = __bfloat162float(t.v[i]);
}
// raw 4-element loads: the conversion to fp32 is a USE of the loaded registers (the compiler puts the s_waitcnt in
// front of it), so kernels that want several loads in flight fetch raw first and convert after the last one is issued
template <typename T> struct Raw4;
template <> struct Raw4<float> { typedef float4 type; };
template <> struct Raw4<bf16> { typedef bf16x4 type; };
template <typename T> __device__ __forceinline__ typename Raw4<T>::type load_raw4(const T* p) {
    return *reinterpret_cast<const typename Raw4<T>::type*>(p);
}
__device__ __forceinline__ void cvt4(const float4& t, float (&o)[4]) { o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w; }
__device__ __forceinline__ void cvt4(const bf16x4& t, float (&o)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = __bfloat162float(t.v[i]);
}
__device__ __forceinline__ void store4(float* p, const float (&o)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
}
__device__ __forceinline__ void store4(bf16* p, const float (&o)[4]) {
    bf16x4 t;
#pragma unroll
    for (int i = 0; i < 4; ++i) t.v[i] = __float2bfloat16(o[i]);
    *reinterpret_cast<bf16x4*>(p) = t;
}

// ---------------------------------------------------------------- wave64 reductions
// Every lane ends with the result.  Four DPP moves finish a row of 16 lanes (quad permutes, row_half_mirror,
// row_mirror: pure VALU, no LDS round trip), four v_readlane + three uniform operations join the four rows.  The
// butterfly of six ds_bpermute this replaces cost ~100 cycles of latency per step -- in the one-row-per-wave
// kernels (LayerNorm, losses) that chain WAS the kernel.  Call with the whole wave active.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row_lane(float v, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_move<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_move<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_move<0x141>(v);  // row_half_mirror
    v += dpp_move<0x140>(v);  // row_mirror
    return (row_lane(v, 0) + row_lane(v, 16)) + (row_lane(v, 32) + row_lane(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_move<0xB1>(v));
    v = fmaxf(v, dpp_move<0x4E>(v));
    v = fmaxf(v, dpp_move<0x141>(v));
    v = fmaxf(v, dpp_move<0x140>(v));
    return fmaxf(fmaxf(row_lane(v, 0), row_lane(v, 16)), fmaxf(row_lane(v, 32), row_lane(v, 48)));
}

// ---------------------------------------------------------------- math
// erf by Abramowitz & Stegun 7.1.26 (|abs error| <= 1.5e-7), branch-free: one rcp, one exp and a
// degree-5 Horner chain instead of libm's range-split erff.  Inside GELU the error is 7.5e-8*|x|,
// far below both the fp32 parity tolerance and bf16 rounding.
__device__ __forceinline__ float erf_fast(float x) {
    const float ax = fabsf(x);
    const float t = __frcp_rn(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float r = 1.0f - p * t * __expf(-ax * ax);
    return copysignf(r, x);
}
__device__ __forceinline__ float gelu_f(float x) {
    return x * 0.5f * (1.0f + erf_fast(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float gelu_grad_f(float x) {
    // d/dx [x Phi(x)] = Phi(x) + x phi(x)
    const float cdf = 0.5f * (1.0f + erf_fast(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + __expf(-x)); }

// ---------------------------------------------------------------- Philox4x32-10
// Counter-based RNG: the dropout mask / Gaussian draw of element `idx` of random
// stream `stream_id` at step `offset` is a pure function of (seed, offset, stream_id,
// idx), so forward and backward regenerate the same mask without storing it.
// (Seven rounds -- the smallest Crush-resistant count -- were measured against the standard ten, same box, alternating
// runs: 11.55 / 11.60 / 11.53 vs 11.53 / 11.47 / 11.51 ms per iteration.  The generator's integer multiplies run
// under the memory waits of the kernels that call it; nothing to gain, so the standard count stays.)
struct Philox {
    uint32_t k0, k1;
    __host__ __device__ Philox(uint64_t seed) : k0((uint32_t)seed), k1((uint32_t)(seed >> 32)) {}
    __host__ __device__ static inline void mulhilo(uint32_t a, uint32_t b, uint32_t& hi, uint32_t& lo) {
        uint64_t p = (uint64_t)a * b;
        hi = (uint32_t)(p >> 32);
        lo = (uint32_t)p;
    }
    __host__ __device__ inline void gen(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t (&out)[4]) const {
        uint32_t a = k0, b = k1;
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            uint32_t h0, l0, h1, l1;
            mulhilo(0xD2511F53u, c0, h0, l0);
            mulhilo(0xCD9E8D57u, c2, h1, l1);
            uint32_t n0 = h1 ^ c1 ^ a, n1 = l1, n2 = h0 ^ c3 ^ b, n3 = l0;
            c0 = n0; c1 = n1; c2 = n2; c3 = n3;
            a += 0x9E3779B9u;
            b += 0xBB67AE85u;
        }
        out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
    }
};

// RNG state lives in device memory so a captured hipGraph replays with fresh numbers:
// rng[0] = seed, rng[1] = offset (advanced by xggm_rng_advance once per pass).
struct RngRef {
    const uint64_t* state;  // may be null when p == 0
    uint32_t stream_id;
};

// uniform in [0,1) for element idx (4 elements share one Philox call: idx>>2 counter)
__device__ __forceinline__ float philox_uniform(uint64_t seed, uint64_t offset, uint32_t stream_id, uint64_t idx) {
    Philox ph(seed);
    uint32_t o[4];
    ph.gen((uint32_t)(idx >> 2), (uint32_t)(idx >> 34), stream_id, (uint32_t)offset ^ (uint32_t)(offset >> 32) * 0x9E3779B9u, o);
    return (float)(o[idx & 3] >> 8) * (1.0f / 16777216.0f);
}
// keep-scale of dropout: 0 if dropped, 1/(1-p) if kept
__device__ __forceinline__ float dropout_scale(float p, float inv_keep, uint64_t seed, uint64_t offset, uint32_t stream_id,
                                               uint64_t idx) {
    return philox_uniform(seed, offset, stream_id, idx) >= p ? inv_keep : 0.0f;
}
// 4 consecutive elements (idx multiple of 4) in one Philox call
__device__ __forceinline__ void dropout_scale4(float p, float inv_keep, uint64_t seed, uint64_t offset, uint32_t stream_id,
                                               uint64_t idx4, float (&s)[4]) {
    Philox ph(seed);
    uint32_t o[4];
    ph.gen((uint32_t)(idx4 >> 2), (uint32_t)(idx4 >> 34), stream_id, (uint32_t)offset ^ (uint32_t)(offset >> 32) * 0x9E3779B9u, o);
#pragma unroll
    for (int i = 0; i < 4; ++i) s[i] = ((float)(o[i] >> 8) * (1.0f / 16777216.0f)) >= p ? inv_keep : 0.0f;
}
// standard normal for element idx (Box-Muller on two of the four words)
__device__ __forceinline__ float philox_normal(uint64_t seed, uint64_t offset, uint32_t stream_id, uint64_t idx) {
    Philox ph(seed);
    uint32_t o[4];
    ph.gen((uint32_t)(idx >> 1), (uint32_t)(idx >> 33), stream_id ^ 0x5bd1e995u,
           (uint32_t)offset ^ (uint32_t)(offset >> 32) * 0x9E3779B9u, o);
    const uint32_t a = o[(idx & 1) * 2], b = o[(idx & 1) * 2 + 1];
    const float u1 = ((float)(a >> 8) + 1.0f) * (1.0f / 16777216.0f);  // (0,1]
    const float u2 = (float)(b >> 8) * (1.0f / 16777216.0f);
    return sqrtf(-2.0f * __logf(u1)) * __cosf(6.28318530717958647692f * u2);
}

// ---------------------------------------------------------------- OCP e4m3 packing (fp8 forward operands)
// four floats -> one dword of e4m3fn bytes: scaled, saturated to +-448 (the largest e4m3 value), round to nearest even
__device__ __forceinline__ int pack4_e4m3(float a, float b, float c, float d, float q) {
    a = fminf(fmaxf(a * q, -448.f), 448.f);
    b = fminf(fmaxf(b * q, -448.f), 448.f);
    c = fminf(fmaxf(c * q, -448.f), 448.f);
    d = fminf(fmaxf(d * q, -448.f), 448.f);
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    return __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
}
// |x| >= 0: the IEEE bit pattern orders like the value, so the running maximum is one integer atomic
__device__ __forceinline__ void atomic_max_nonneg(float* dst, float v) {
    if (v > 0.f) atomicMax(reinterpret_cast<unsigned int*>(dst), __float_as_uint(v));
}

// raise entry `amax` (spread over `slots` floats, see xggm_gemm_problem.amax_slots) from workgroup `wg`
__device__ __forceinline__ void amax_record(float* amax, int slots, int wg, float v) {
    atomic_max_nonneg(amax + (slots > 1 ? (wg & (slots - 1)) : 0), v);
}

// Scale-table protocol of the e4m3 producers (xggm_fp8_scale_update): *qscale <= 0 marks an entry that has not been
// calibrated -- quantise with 1 and record every maximum; otherwise only maxima beyond half the representable range
// 448 / q are recorded (e4m3 is a floating-point format: a smaller stale range costs no precision, only a larger
// one saturates), which keeps the same-address atomics of a launch to the rare workgroups that matter.
struct Q8 {
    float q, thr;
    __device__ __forceinline__ explicit Q8(const float* qscale) {
        const float s = qscale ? *qscale : 1.f;
        q = s > 0.f ? s : 1.f;
        thr = (qscale && s > 0.f) ? 0.5f * 448.f / s : 0.f;
    }
};

// ---------------------------------------------------------------- grid-wide sums without floating-point atomics
// A scalar that many workgroups contribute to (a loss, GIN's eps gradient) used to be one atomicAdd per workgroup: the
// order of the adds, and with it the last bits of the sum, depended on which workgroup finished first.  Instead every
// workgroup leaves its partial in ws[XGGM_SUM_WS_HEAD + flat block id] and takes a ticket from the counter in ws[0];
// the workgroup that draws the LAST ticket adds the partials in index order.  Same bits whatever the scheduling, one
// launch.  (The release fence per workgroup is what DESIGN.md section 4.2 found too expensive for the gigabyte-sized
// norm pass -- that one keeps its two launches -- but these kernels move a few megabytes at most.)
// `ws`: XGGM_SUM_WS_FLOATS floats (xggm.h), ws[0] == 0 at launch; the kernel leaves it 0.
// Call from EVERY thread of EVERY workgroup (barriers inside); `part` is read from thread 0.  Returns true on thread 0
// of the finishing workgroup with `total` set.
constexpr int SUM_WS_HEAD = 8;
__device__ __forceinline__ bool ordered_grid_sum(float part, float* ws, int nblk, int blk, float& total) {
    __shared__ int s_last;
    __shared__ float s_red[16];
    unsigned* counter = reinterpret_cast<unsigned*>(ws);
    if (threadIdx.x == 0) {
        // The partial goes out as a device-scope atomic store (write-through, past this XCD's L2) and the ticket is a
        // device-scope atomic as well; all the ticket has to wait for is THAT store's acknowledgement (s_waitcnt vmcnt(0):
        // a wave's memory operations are acknowledged in order on this path), not a release fence -- which on gfx950 is
        // a write-back of the XCD's whole L2 (buffer_wbl2) and cost the loss kernels more than their arithmetic:
        // symkl forward 29 us against its backward's 12 (rocprofv3, round 4).  Nothing else is published here.
#ifdef XGGM_SUM_FENCE
        __hip_atomic_store(ws + SUM_WS_HEAD + blk, part, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        const unsigned t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
#else
        __hip_atomic_store(ws + SUM_WS_HEAD + blk, part, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        s_last = (t == (unsigned)nblk - 1u) ? 1 : 0;
    }
    __syncthreads();
    if (!s_last) return false;
#ifdef XGGM_SUM_FENCE
    __threadfence();  // acquire: every other workgroup's partial
#endif
    // (the partials are read with device-scope atomic loads below: they do not come from this XCD's L2)
    float t = 0.f;
    for (int b = threadIdx.x; b < nblk; b += blockDim.x)  // thread-strided, then waves in index order: fixed
        t += __hip_atomic_load(ws + SUM_WS_HEAD + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    t = wave_sum(t);
    const int nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x != 0) return false;
    float r = s_red[0];
    for (int w = 1; w < nw; ++w) r += s_red[w];
    __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
    total = r;
    return true;
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------- weight prefetch beside latency-bound row kernels
// The step's products wait for weights nobody has touched since the previous pass: tools/exp_weight_prefetch.py puts a
// launch whose weights already sit in the Infinity Cache 0.3 ... 1.5 us ahead of one that fetches them from HBM.  The
// kernels between those products -- attention cores, LayerNorm forward / backward -- leave most workgroup slots of the
// chip empty for microseconds of latency: xggm_prefetch_next() queues up to four byte ranges on the host, the next such
// launch takes them, appends `blocks` workgroups to its grid, and those read the ranges with plain loads (which
// allocate on the way) and discard them.  No result depends on it; XGGM_PREFETCH=0 turns the queue into a no-op.
// What a carrier can take without becoming longer itself was measured (profiles/r04_experiments/prefetch.txt): 19 MB
// in one LayerNorm launch made the launch 1.8-2.3 us longer and gave back what the products had gained; the same reads
// on a side stream cost 0.9 ms per iteration in graph branches.  Hence <= ~9 MB per LayerNorm launch, the rest on the
// (longer) attention launches.
struct PrefetchArgs {
    const void* p[4];
    unsigned long long n[4];  // bytes (whole 16-byte chunks are read)
    int k;                    // ranges
    int blocks;               // workgroups appended to the grid (0: nothing queued)
    int* sink;                // never written in practice: keeps the loads alive
};
PrefetchArgs xggm_take_prefetch();  // host: the queued ranges (the queue is cleared)

__device__ __forceinline__ void prefetch_role(const PrefetchArgs& pf, int b) {
    typedef int i4 __attribute__((ext_vector_type(4)));
    const int nt = blockDim.x, tid = threadIdx.x;
    int acc = 0;
    for (int r = 0; r < pf.k; ++r) {
        const i4* q = reinterpret_cast<const i4*>(pf.p[r]);
        const long long n16 = (long long)(pf.n[r] >> 4), step = (long long)pf.blocks * nt;
        for (long long i = (long long)b * nt + tid; i < n16; i += 8 * step) {  // eight loads in flight per thread
            i4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = i + u * step < n16 ? q[i + u * step] : (i4){0, 0, 0, 0};
#pragma unroll
            for (int u = 0; u < 8; ++u) acc ^= v[u][0] ^ v[u][3];
        }
    }
    if (acc == 0x5a17c3d9 && pf.sink) *pf.sink = acc;
}
