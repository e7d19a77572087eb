// xggm_attn_fwd / xggm_attn_bwd: the small-sequence attention core of BertAttention
// (src/lxrt/modeling.py:355-373): softmax(Q K^T / sqrt(d) + mask) -> dropout -> P V, and its
// backward.  Sequences on this path are 20 tokens / 36 objects (<= 64), d = 64, so a whole
// (batch, head) problem -- Q, K, V tiles, the score matrix and, in backward, dO and dS --
// lives in the LDS of one 256-thread workgroup: HBM traffic is exactly one read of
// Q/K/V(/dO) and one write of O (dQ/dK/dV).  Nothing but the inputs is saved for backward:
// P and the Philox dropout mask are recomputed.  Per-row softmax uses wave64 shuffle
// reductions; fp32 math throughout.  These scalar kernels serve fp32 storage (exact parity
// mode); bf16 storage runs the matrix-core versions of attention_mfma.hip.
#include "common.h"
#include "xggm.h"

// bf16 matrix-core versions (attention_mfma.hip): one or two validated problems per launch
int xggm_attn_fwd_mfma_group(const xggm_attn_problem* probs, int n, const uint64_t* rng, hipStream_t st);
int xggm_attn_bwd_mfma_group(const xggm_attn_problem* probs, int n, const uint64_t* rng, hipStream_t st);

namespace {

constexpr int NT = 256;
constexpr int D = 64;
constexpr int LD = D + 4;  // float4-aligned rows, conflict-free 16-byte reads
bool g_attn_scalar = false;  // test hook (xggm_attn_set_scalar): bf16 on the scalar kernels

inline bool mfma_ok(const void* a, const void* b, const void* c, const void* d, int64_t s0, int64_t s1, int64_t s2, int64_t s3) {
    return !g_attn_scalar &&
           ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c) |
             reinterpret_cast<uintptr_t>(d)) % 16 == 0) &&
           s0 % 8 == 0 && s1 % 8 == 0 && s2 % 8 == 0 && s3 % 8 == 0;
}

struct AttnArgs {
    const void *q, *k, *v;
    const float* mask;  // additive [B, Sk] or null
    int64_t q_rs, k_rs, v_rs, o_rs;  // row strides in elements
    int B, heads, Sq, Sk;
    float scale, p;
    const uint64_t* rng;
    uint32_t sid;
};

template <typename T>
__device__ __forceinline__ void load_tile(float* lds, const T* base, int64_t rs, int rows, int tid) {
    // rows x 64 elements, 16 chunks of 4 per row
    for (int c = tid; c < rows * 16; c += NT) {
        const int r = c >> 4, cc = (c & 15) * 4;
        float t[4];
        load4(base + (int64_t)r * rs + cc, t);
        *reinterpret_cast<float4*>(lds + r * LD + cc) = make_float4(t[0], t[1], t[2], t[3]);
    }
}

__device__ __forceinline__ float dot64(const float* a, const float* b) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < D; c += 4) {
        const float4 x = *reinterpret_cast<const float4*>(a + c);
        const float4 y = *reinterpret_cast<const float4*>(b + c);
        s += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
    }
    return s;
}

// scores + softmax (+ dropout scale matrix) into LDS.  P: probabilities, Dm: dropout scale
// (only written when p > 0).  Wave w owns rows w, w+4, ...
__device__ __forceinline__ void scores_softmax(const AttnArgs& a, const float* Qs, const float* Ks, float* P, float* Dm, int b,
                                               int h, int tid) {
    const int Sq = a.Sq, Sk = a.Sk, ldp = Sk + 1;
    for (int s = tid; s < Sq * Sk; s += NT) {
        const int i = s / Sk, j = s % Sk;
        float v = dot64(Qs + i * LD, Ks + j * LD) * a.scale;
        if (a.mask) v += a.mask[(int64_t)b * Sk + j];
        P[i * ldp + j] = v;
    }
    __syncthreads();
    const int lane = tid & 63, wid = tid >> 6;
    uint64_t seed = 0, off = 0;
    if (a.p > 0.f) {
        seed = a.rng[0];
        off = a.rng[1];
    }
    const float ik = a.p > 0.f ? 1.f / (1.f - a.p) : 1.f;
    for (int i = wid; i < Sq; i += NT / 64) {
        const float v = lane < Sk ? P[i * ldp + lane] : -INFINITY;
        const float m = wave_max(v);
        const float e = lane < Sk ? __expf(v - m) : 0.f;
        const float sum = wave_sum(e);
        if (lane < Sk) {
            P[i * ldp + lane] = e / sum;
            if (a.p > 0.f) {
                const uint64_t idx = (((uint64_t)b * a.heads + h) * Sq + i) * Sk + lane;
                Dm[i * ldp + lane] = dropout_scale(a.p, ik, seed, off, a.sid, idx);
            }
        }
    }
    __syncthreads();
}

template <typename T> __global__ __launch_bounds__(NT) void attn_fwd_kernel(AttnArgs a, T* out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int b = blockIdx.x / a.heads, h = blockIdx.x % a.heads;
    const int Sq = a.Sq, Sk = a.Sk, ldp = Sk + 1;
    float* Qs = smem;
    float* Ks = Qs + Sq * LD;
    float* Vs = Ks + Sk * LD;
    float* P = Vs + Sk * LD;
    float* Dm = P + Sq * ldp;
    load_tile<T>(Qs, (const T*)a.q + (int64_t)b * Sq * a.q_rs + h * D, a.q_rs, Sq, tid);
    load_tile<T>(Ks, (const T*)a.k + (int64_t)b * Sk * a.k_rs + h * D, a.k_rs, Sk, tid);
    load_tile<T>(Vs, (const T*)a.v + (int64_t)b * Sk * a.v_rs + h * D, a.v_rs, Sk, tid);
    __syncthreads();
    scores_softmax(a, Qs, Ks, P, Dm, b, h, tid);
    // O[i][c..c+3] = sum_j P[i][j] * D[i][j] * V[j][c..c+3]
    for (int w = tid; w < Sq * 16; w += NT) {
        const int i = w >> 4, c = (w & 15) * 4;
        float o[4] = {0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < Sk; ++j) {
            float pj = P[i * ldp + j];
            if (a.p > 0.f) pj *= Dm[i * ldp + j];
            const float4 vv = *reinterpret_cast<const float4*>(Vs + j * LD + c);
            o[0] += pj * vv.x;
            o[1] += pj * vv.y;
            o[2] += pj * vv.z;
            o[3] += pj * vv.w;
        }
        store4(out + ((int64_t)b * Sq + i) * a.o_rs + h * D + c, o);
    }
}

template <typename T>
__global__ __launch_bounds__(NT) void attn_bwd_kernel(AttnArgs a, const T* d_out, T* dq, T* dk, T* dv, int64_t dq_rs,
                                                      int64_t dk_rs, int64_t dv_rs, float* dbq, float* dbk, float* dbv,
                                                      int64_t db_bs) {
    // column sums of dQ / dK / dV over this sample's rows (bias gradients): every thread keeps the sums of ITS column
    // group over its rows (NT % 16 == 0: a thread's columns never change), the threads of a column group are added in
    // index order through `red`, and the result is a partial row of this (sample, head) nobody else writes
    __shared__ float red[NT / 16][D];
    float cq[4] = {0.f, 0.f, 0.f, 0.f}, ck[4] = {0.f, 0.f, 0.f, 0.f}, cv[4] = {0.f, 0.f, 0.f, 0.f};
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int b = blockIdx.x / a.heads, h = blockIdx.x % a.heads;
    const int Sq = a.Sq, Sk = a.Sk, ldp = Sk + 1;
    float* Qs = smem;
    float* Ks = Qs + Sq * LD;
    float* Vs = Ks + Sk * LD;
    float* dOs = Vs + Sk * LD;
    float* P = dOs + Sq * LD;
    float* Dm = P + Sq * ldp;
    float* dS = Dm + Sq * ldp;
    load_tile<T>(Qs, (const T*)a.q + (int64_t)b * Sq * a.q_rs + h * D, a.q_rs, Sq, tid);
    load_tile<T>(Ks, (const T*)a.k + (int64_t)b * Sk * a.k_rs + h * D, a.k_rs, Sk, tid);
    load_tile<T>(Vs, (const T*)a.v + (int64_t)b * Sk * a.v_rs + h * D, a.v_rs, Sk, tid);
    load_tile<T>(dOs, d_out + (int64_t)b * Sq * a.o_rs + h * D, a.o_rs, Sq, tid);
    __syncthreads();
    scores_softmax(a, Qs, Ks, P, Dm, b, h, tid);
    // dP[i][j] = D[i][j] * sum_c dO[i][c] V[j][c]
    for (int s = tid; s < Sq * Sk; s += NT) {
        const int i = s / Sk, j = s % Sk;
        float v = dot64(dOs + i * LD, Vs + j * LD);
        if (a.p > 0.f) v *= Dm[i * ldp + j];
        dS[i * ldp + j] = v;
    }
    __syncthreads();
    // dS = P * (dP - rowsum(dP * P)) * scale
    {
        const int lane = tid & 63, wid = tid >> 6;
        for (int i = wid; i < Sq; i += NT / 64) {
            const float pv = lane < Sk ? P[i * ldp + lane] : 0.f;
            const float dp = lane < Sk ? dS[i * ldp + lane] : 0.f;
            const float rs = wave_sum(pv * dp);
            if (lane < Sk) dS[i * ldp + lane] = pv * (dp - rs) * a.scale;
        }
    }
    __syncthreads();
    // dQ[i][c] = sum_j dS[i][j] K[j][c]
    // whole waves iterate (the shuffles below need every lane); lanes past the end are predicated
    for (int w = tid; w < ((Sq * 16 + 63) & ~63); w += NT) {
        const bool valid = w < Sq * 16;
        const int i = valid ? (w >> 4) : 0, c = (w & 15) * 4;
        float o[4] = {0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < (valid ? Sk : 0); ++j) {
            const float s = dS[i * ldp + j];
            const float4 kk = *reinterpret_cast<const float4*>(Ks + j * LD + c);
            o[0] += s * kk.x;
            o[1] += s * kk.y;
            o[2] += s * kk.z;
            o[3] += s * kk.w;
        }
        if (valid) store4(dq + ((int64_t)b * Sq + i) * dq_rs + h * D + c, o);
#pragma unroll
        for (int e = 0; e < 4; ++e) cq[e] += o[e];  // rows past the end contributed zeros
    }
    // dK[j][c] = sum_i dS[i][j] Q[i][c];  dV[j][c] = sum_i P[i][j] D[i][j] dO[i][c]
    for (int w = tid; w < ((Sk * 16 + 63) & ~63); w += NT) {
        const bool valid = w < Sk * 16;
        const int j = valid ? (w >> 4) : 0, c = (w & 15) * 4;
        float ok[4] = {0.f, 0.f, 0.f, 0.f}, ov[4] = {0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < (valid ? Sq : 0); ++i) {
            const float s = dS[i * ldp + j];
            float pj = P[i * ldp + j];
            if (a.p > 0.f) pj *= Dm[i * ldp + j];
            const float4 qq = *reinterpret_cast<const float4*>(Qs + i * LD + c);
            const float4 gg = *reinterpret_cast<const float4*>(dOs + i * LD + c);
            ok[0] += s * qq.x;
            ok[1] += s * qq.y;
            ok[2] += s * qq.z;
            ok[3] += s * qq.w;
            ov[0] += pj * gg.x;
            ov[1] += pj * gg.y;
            ov[2] += pj * gg.z;
            ov[3] += pj * gg.w;
        }
        if (valid) {
            store4(dk + ((int64_t)b * Sk + j) * dk_rs + h * D + c, ok);
            store4(dv + ((int64_t)b * Sk + j) * dv_rs + h * D + c, ov);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            ck[e] += ok[e];
            cv[e] += ov[e];
        }
    }
    auto fold = [&](const float (&c4)[4], float* dst) {  // uniform: every thread of the workgroup calls it
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) red[tid >> 4][(tid & 15) * 4 + e] = c4[e];
        __syncthreads();
        if (tid < D) {
            float s = 0.f;
            for (int g = 0; g < NT / 16; ++g) s += red[g][tid];  // fixed order
            dst[(int64_t)b * db_bs + h * D + tid] = s;
        }
    };
    if (dbq) fold(cq, dbq);
    if (dbk) fold(ck, dbk);
    if (dbv) fold(cv, dbv);
}

int check(const char* who, const AttnArgs& a, int head_dim) {
    XGGM_REQUIRE(head_dim == D, "%s: head_dim %d != 64", who, head_dim);
    XGGM_REQUIRE(a.B > 0 && a.heads > 0 && a.Sq > 0 && a.Sk > 0, "%s: empty problem", who);
    XGGM_REQUIRE(a.Sq <= 64 && a.Sk <= 64, "%s: Sq=%d Sk=%d exceed the 64-row LDS tile", who, a.Sq, a.Sk);
    XGGM_REQUIRE(a.q && a.k && a.v, "%s: null pointer", who);
    XGGM_REQUIRE(a.q_rs % 4 == 0 && a.k_rs % 4 == 0 && a.v_rs % 4 == 0 && a.o_rs % 4 == 0, "%s: row strides must be multiples of 4",
                 who);
    XGGM_REQUIRE(a.p >= 0.f && a.p < 1.f && (a.p == 0.f || a.rng), "%s: bad dropout arguments", who);
    XGGM_REQUIRE((int64_t)a.B * a.heads < (1 << 30), "%s: grid too large", who);
    return XGGM_OK;
}

inline AttnArgs args_of(const xggm_attn_problem& q, const uint64_t* rng) {
    return AttnArgs{q.q, q.k, q.v, q.mask, q.q_rs, q.k_rs, q.v_rs, q.o_rs, q.B, q.heads, q.Sq, q.Sk, q.scale, q.p, rng, q.sid};
}

// problems that can run on the matrix-core kernels are launched in pairs, the rest one by one
template <typename T>
int attn_fwd_grouped(const xggm_attn_problem* probs, int n, int head_dim, const uint64_t* rng, hipStream_t st) {
    XGGM_REQUIRE(probs && n > 0, "xggm_attn_fwd: no problems");
    xggm_attn_problem pend[2];
    int np = 0;
    for (int i = 0; i < n; ++i) {
        const xggm_attn_problem& q = probs[i];
        const AttnArgs a = args_of(q, rng);
        if (int e = check("xggm_attn_fwd", a, head_dim)) return e;
        XGGM_REQUIRE(q.out, "xggm_attn_fwd: null output");
        XGGM_REQUIRE(!q.out8 || (sizeof(T) == 2 && reinterpret_cast<uintptr_t>(q.out8) % 8 == 0),
                     "xggm_attn_fwd: the e4m3 output copy needs bf16 storage and an 8-byte aligned buffer");
        if (sizeof(T) == 2 && mfma_ok(q.q, q.k, q.v, q.out, q.q_rs, q.k_rs, q.v_rs, q.o_rs)) {
            pend[np++] = q;
            if (np == 2 || i == n - 1) {
                if (int e = xggm_attn_fwd_mfma_group(pend, np, rng, st)) return e;
                np = 0;
            }
            continue;
        }
        XGGM_REQUIRE(!q.out8, "xggm_attn_fwd: the e4m3 output copy is written by the matrix-core kernel only (16-byte "
                              "aligned operands, strides multiples of 8)");
        const size_t lds = sizeof(float) * ((size_t)(q.Sq + 2 * q.Sk) * LD + 2 * (size_t)q.Sq * (q.Sk + 1));
        hipLaunchKernelGGL((attn_fwd_kernel<T>), dim3(q.B * q.heads), dim3(NT), lds, st, a, (T*)q.out);
        if (int e = xggm_check_launch("xggm_attn_fwd")) return e;
    }
    if (np) return xggm_attn_fwd_mfma_group(pend, np, rng, st);
    return XGGM_OK;
}

template <typename T>
int attn_bwd_grouped(const xggm_attn_problem* probs, int n, int head_dim, const uint64_t* rng, hipStream_t st) {
    XGGM_REQUIRE(probs && n > 0, "xggm_attn_bwd: no problems");
    xggm_attn_problem pend[2];
    int np = 0;
    for (int i = 0; i < n; ++i) {
        const xggm_attn_problem& q = probs[i];
        const AttnArgs a = args_of(q, rng);
        if (int e = check("xggm_attn_bwd", a, head_dim)) return e;
        XGGM_REQUIRE(q.d_out && q.dq && q.dk && q.dv, "xggm_attn_bwd: null pointer");
        XGGM_REQUIRE(q.dq_rs % 4 == 0 && q.dk_rs % 4 == 0 && q.dv_rs % 4 == 0, "xggm_attn_bwd: row strides must be multiples of 4");
        XGGM_REQUIRE((q.dbk == nullptr) == (q.dbv == nullptr), "xggm_attn_bwd: dbk and dbv go together");
        XGGM_REQUIRE(!(q.dbq || q.dbk) || q.B == 1 || q.db_bs >= (int64_t)q.heads * D,
                     "xggm_attn_bwd: the bias-gradient partial rows need a batch stride db_bs >= heads * 64 (got %lld)",
                     (long long)q.db_bs);
        const bool grads8 = (reinterpret_cast<uintptr_t>(q.dq) | reinterpret_cast<uintptr_t>(q.dk) | reinterpret_cast<uintptr_t>(q.dv)) % 8 == 0;
        if (sizeof(T) == 2 && grads8 && mfma_ok(q.q, q.k, q.v, q.d_out, q.q_rs, q.k_rs, q.v_rs, q.o_rs)) {  // 8-byte gradient stores
            pend[np++] = q;
            if (np == 2 || i == n - 1) {
                if (int e = xggm_attn_bwd_mfma_group(pend, np, rng, st)) return e;
                np = 0;
            }
            continue;
        }
        const size_t lds = sizeof(float) * ((size_t)(2 * q.Sq + 2 * q.Sk) * LD + 3 * (size_t)q.Sq * (q.Sk + 1));
        hipLaunchKernelGGL((attn_bwd_kernel<T>), dim3(q.B * q.heads), dim3(NT), lds, st, a, (const T*)q.d_out, (T*)q.dq, (T*)q.dk,
                           (T*)q.dv, q.dq_rs, q.dk_rs, q.dv_rs, q.dbq, q.dbk, q.dbv, q.db_bs);
        if (int e = xggm_check_launch("xggm_attn_bwd")) return e;
    }
    if (np) return xggm_attn_bwd_mfma_group(pend, np, rng, st);
    return XGGM_OK;
}

template <typename T>
int attn_fwd(const void* q, const void* k, const void* v, const float* mask, void* out, int B, int heads, int Sq, int Sk,
             int head_dim, int64_t q_rs, int64_t k_rs, int64_t v_rs, int64_t o_rs, float scale, float p, const uint64_t* rng,
             uint32_t sid, hipStream_t st) {
    xggm_attn_problem pr{};
    pr.q = q; pr.k = k; pr.v = v; pr.mask = mask; pr.out = out;
    pr.B = B; pr.heads = heads; pr.Sq = Sq; pr.Sk = Sk;
    pr.q_rs = q_rs; pr.k_rs = k_rs; pr.v_rs = v_rs; pr.o_rs = o_rs;
    pr.scale = scale; pr.p = p; pr.sid = sid;
    return attn_fwd_grouped<T>(&pr, 1, head_dim, rng, st);
}

template <typename T>
int attn_bwd(const void* q, const void* k, const void* v, const float* mask, const void* d_out, void* dq, void* dk, void* dv,
             int B, int heads, int Sq, int Sk, int head_dim, int64_t q_rs, int64_t k_rs, int64_t v_rs, int64_t o_rs,
             int64_t dq_rs, int64_t dk_rs, int64_t dv_rs, float scale, float p, const uint64_t* rng, uint32_t sid,
             float* dbq, float* dbk, float* dbv, int64_t db_bs, hipStream_t st) {
    xggm_attn_problem pr{};
    pr.q = q; pr.k = k; pr.v = v; pr.mask = mask;
    pr.B = B; pr.heads = heads; pr.Sq = Sq; pr.Sk = Sk;
    pr.q_rs = q_rs; pr.k_rs = k_rs; pr.v_rs = v_rs; pr.o_rs = o_rs;
    pr.scale = scale; pr.p = p; pr.sid = sid;
    pr.d_out = d_out; pr.dq = dq; pr.dk = dk; pr.dv = dv;
    pr.dq_rs = dq_rs; pr.dk_rs = dk_rs; pr.dv_rs = dv_rs;
    pr.dbq = dbq; pr.dbk = dbk; pr.dbv = dbv; pr.db_bs = db_bs;
    return attn_bwd_grouped<T>(&pr, 1, head_dim, rng, st);
}

}  // namespace

#define ATTN_API(SUF, T)                                                                                                   \
    extern "C" int xggm_attn_fwd_##SUF(const void* q, const void* k, const void* v, const float* mask, void* out, int B,  \
                                       int heads, int Sq, int Sk, int head_dim, int64_t q_rs, int64_t k_rs, int64_t v_rs, \
                                       int64_t o_rs, float scale, float p, const uint64_t* rng, uint32_t sid,            \
                                       hipStream_t st) {                                                                  \
        return attn_fwd<T>(q, k, v, mask, out, B, heads, Sq, Sk, head_dim, q_rs, k_rs, v_rs, o_rs, scale, p, rng, sid, st);\
    }                                                                                                                      \
    extern "C" int xggm_attn_bwd_##SUF(const void* q, const void* k, const void* v, const float* mask, const void* d_out,  \
                                       void* dq, void* dk, void* dv, int B, int heads, int Sq, int Sk, int head_dim,      \
                                       int64_t q_rs, int64_t k_rs, int64_t v_rs, int64_t o_rs, int64_t dq_rs,             \
                                       int64_t dk_rs, int64_t dv_rs, float scale, float p, const uint64_t* rng,           \
                                       uint32_t sid, float* dbq, float* dbk, float* dbv, int64_t db_bs,                  \
                                       hipStream_t st) {                                                                  \
        return attn_bwd<T>(q, k, v, mask, d_out, dq, dk, dv, B, heads, Sq, Sk, head_dim, q_rs, k_rs, v_rs, o_rs, dq_rs,    \
                           dk_rs, dv_rs, scale, p, rng, sid, dbq, dbk, dbv, db_bs, st);                                    \
    }                                                                                                                      \
    extern "C" int xggm_attn_fwd_grouped_##SUF(const xggm_attn_problem* probs, int n, int head_dim, const uint64_t* rng,   \
                                               hipStream_t st) {                                                          \
        return attn_fwd_grouped<T>(probs, n, head_dim, rng, st);                                                           \
    }                                                                                                                      \
    extern "C" int xggm_attn_bwd_grouped_##SUF(const xggm_attn_problem* probs, int n, int head_dim, const uint64_t* rng,   \
                                               hipStream_t st) {                                                          \
        return attn_bwd_grouped<T>(probs, n, head_dim, rng, st);                                                           \
    }

ATTN_API(f32, float)
ATTN_API(bf16, bf16)

// test / A-B hook: 1 = bf16 attention on the scalar fp32-math kernels, 0 = matrix-core kernels
extern "C" int xggm_attn_set_scalar(int on) {
    g_attn_scalar = on != 0;
    return XGGM_OK;
}
