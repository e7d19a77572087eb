// xggm_gemm_{f32,bf16}: C[m,n] = epilogue( alpha * sum_k A(m,k) B(k,n) )
//
// One LDS-tiled MFMA kernel serves every dense product on the path -- Linear forward
// (A=[M,K] k-contiguous, B=W[N,K] k-contiguous), dgrad (B=W read with the reduction
// index on the slow dimension), wgrad (both operands read with the reduction index on
// the slow dimension) and the small batched products -- because both operands are
// addressed through (row stride, k stride) pairs and are transposed, when needed, on
// their way into LDS.  Replaces torch.nn.Linear / torch.bmm call sites of
// src/lxrt/modeling.py:345-347,385,429,442,617 and src/module/gcn.py:28.
//
// Tile: 64x64 per 256-thread workgroup (4 waves as 2x2, 32x32 per wave = 2x2 MFMA
// 16x16 tiles).  The skinny shapes of this workload (M = B*20 / B*36) give 120-900
// workgroups per launch at B = 32, enough to cover 256 CUs, which a 128^2/256^2 tile
// would not.  bf16: v_mfma_f32_16x16x32_bf16, BK = 64; f32: v_mfma_f32_16x16x4_f32
// (exact fp32 FMA chain), BK = 32.  LDS rows are k-contiguous and padded by 16 bytes.
#include "common.h"
#include "xggm.h"

namespace {

typedef __attribute__((ext_vector_type(8))) short short8_t;
typedef __attribute__((ext_vector_type(4))) float float4_t;

struct GemmArgs {
    const void* A;
    const void* B;
    void* C;
    int M, N, K;
    int64_t a_rs, a_ks, b_ns, b_ks, ldc;
    int64_t a_bs, b_bs, c_bs;
    const float* bias;     // [N] or null
    const void* residual;  // T, layout of C (same ldc / batch stride), or null
    void* preact;          // T, layout of C, or null: alpha*acc + bias before the activation
    const void* aux;       // T, layout of C: pre-activation u for act == GELU_GRAD
    float* colsum;         // or null: [batch][ceil(M/32)][N] partial column sums of the stored result, one row per block of 32
                           // output rows, every element written once by exactly one workgroup (no atomics: see xggm.h)
    float* sqsum;          // or null: [ceil(M/64)][ceil(N/64)] sums of squares of the STORED fp32 values per 64 x 64 block
    unsigned char* c8;     // or null: e4m3 copy of the stored result (layout of C), scaled by *c8_qscale
    const float* c8_qscale;
    float* c8_amax;        // or null: raised to max |stored value|
    int amax_slots;        // floats the entry is spread over (xggm_gemm_problem.amax_slots)
    const float* scale_a;  // fp8 operands: reciprocal quantisation scales (device scalars, null = 1)
    const float* scale_b;
    int act;
    int c_f32;         // store C as float regardless of T
    int accumulate;    // C += result
    float alpha;
    int a_mode, b_mode;  // 0 scalar, 1 vector along k, 2 vector along rows
    // tuned path on shapes that are not 8-aligned (A = 2274 answers, 630 edges), see pick_mode:
    int a_tail, b_tail;  // k-contiguous operand with K % 8 != 0: the last chunk's elements >= K are zeroed on the way to LDS
    int a_rows, b_rows;  // rows an operand may be READ at (row-contiguous operands: padded up to a multiple of 8)
    int xcd_swizzle;
    int batch;
    int use_glds;  // k-major operand pairs take the LDS-DMA k-loop (gemm_kloop_glds)
    // block -> tile map of the grouped kernels, constants computed on the host per (problem, tile shape): see tile_from_map
    struct TileMap {
        int gx, gy, gxy;              // tile grid; tiles per batch entry
        int xr, xc, rh, rw;           // XCD rectangles: xr bands of rh tile rows x xc column groups of rw tiles
        int ncg, wb_last;             // non-empty column groups, width of the last one
        unsigned m_rw, m_wbl, m_gxy;  // ceil(2^32 / d) for d = rw, wb_last, gxy (unused where d == 1)
    } tm;
#ifdef XGGM_STAMP
    long long* stamp;  // instrumented build (make stamp): 8 cycle-counter slots per workgroup
    int ablate;        // instrumented build: 1 skips the chunk loop of the epilogue, 2 the whole epilogue
#endif
};

#ifdef XGGM_STAMP
long long* g_stamp = nullptr;
int g_ablate = 0;
#define ABLATE(g, bit) ((g).ablate & (bit))
#ifdef XGGM_KABLATE
#define KABLATE(g, bit) ((g).ablate & (bit))
#else
#define KABLATE(g, bit) false
#endif
#define STAMP(g, slot)                                                                                \
    do {                                                                                              \
        if ((g).stamp && threadIdx.x == 0)                                                            \
            (g).stamp[(int64_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 8 + (slot)] = \
                __builtin_readcyclecounter();                                                         \
    } while (0)
#define SET_STAMP(g) ((g).stamp = g_stamp, (g).ablate = g_ablate)
// slots 5, 6: HW_ID (cu / sh / se fields) and XCC_ID of the workgroup's first wave
#define STAMP_HW(g)                                                                                   \
    do {                                                                                              \
        if ((g).stamp && threadIdx.x == 0) {                                                          \
            unsigned hw, xcc;                                                                         \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));                          \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));                        \
            long long* q = (g).stamp + (int64_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 8; \
            q[5] = hw;                                                                                \
            q[6] = xcc;                                                                               \
        }                                                                                             \
    } while (0)
// fine timeline of the workgroup's first wave (tools/gemm_trace.py): slot -> cycle counter, kept in LDS and copied to
// stamp[XGGM_TRACE_BASE + 128 * workgroup ...] at the end of the tile.  s_memtime returns through lgkmcnt: every TRACE
// point drains the wave's outstanding LDS reads first, so points sit where the wave would have waited anyway.
#define XGGM_TRACE_BASE (8 * 8192)
__shared__ long long xg_trace[128];
#define TRACE(g, slot)                                                                                \
    do {                                                                                              \
        if (((g).ablate & 0x10000) && threadIdx.x == 0 && (slot) < 128) xg_trace[(slot)] = __builtin_readcyclecounter(); \
    } while (0)
#define TRACE_FLUSH(g)                                                                                \
    do {                                                                                              \
        if (((g).ablate & 0x10000) && threadIdx.x < 64) {                                                          \
            long long* q = (g).stamp + XGGM_TRACE_BASE +                                              \
                           (int64_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 128; \
            q[threadIdx.x] = xg_trace[threadIdx.x];                                                   \
            q[threadIdx.x + 64] = xg_trace[threadIdx.x + 64];                                         \
        }                                                                                             \
    } while (0)
#else
#define TRACE(g, slot)
#define TRACE_FLUSH(g)
#define STAMP_HW(g)
#define ABLATE(g, bit) false
#define KABLATE(g, bit) false
#define STAMP(g, slot)
#define SET_STAMP(g)
#endif

template <typename T> struct Tile;
template <> struct Tile<bf16> {
    static constexpr int BK = 64;
    static constexpr int VEC = 8;  // elements per 16-byte chunk
    static constexpr int LDK = BK + 8;
};
template <> struct Tile<float> {
    static constexpr int BK = 32;
    static constexpr int VEC = 4;
    static constexpr int LDK = BK + 4;
};

constexpr int BM = 64, BN = 64, NT = 256;
bool g_force_generic = false;  // test hook: xggm_gemm_set_generic
int g_xcd_swizzle = 1;         // test hook: xggm_gemm_set_tile(variant | 0x100) disables it
int g_glds_stages = 0;        // test hook: xggm_gemm_set_tile(variant | 0x800 / 0x1000) pins 2 / 3 LDS stages
int g_single_grouped = 1;      // single problems with whole k-tiles also take the grouped (LDS-DMA) kernel: -0.05 ms per iteration
                               // (same-box A/B 11.33 / 11.05 / 11.06 vs 11.01 / 11.00 / 11.03); xggm_gemm_set_tile(variant | 0x8000) turns it off
int g_no_8w = 0;               // test hook: xggm_gemm_set_tile(variant | 0x4000): no 8-wave 128 x 128 tile
int g_glds = 1;                // test hook: xggm_gemm_set_tile(variant | 0x400) keeps k-major pairs on the register-staged k-loop
int g_group_tile = 0;          // test hook: 0 heuristic, 1: 64x64, 2: 128x64, 3: 128x128, 4: 128x128 on 8 waves

// stage a [64 rows][BK] operand tile into LDS (k contiguous).  elem(r,k) = base[r*rs + k*ks]
template <typename T>
__device__ __forceinline__ void stage_tile(T* __restrict__ lds, const T* __restrict__ base, int64_t rs, int64_t ks, int r0,
                                           int k0, int R, int K, int mode, int tid) {
    constexpr int BK = Tile<T>::BK, VEC = Tile<T>::VEC, LDK = Tile<T>::LDK;
    typedef typename std::conditional<sizeof(T) == 2, short8_t, float4_t>::type vec_t;
    if (mode == 1) {
        // 16-byte chunks along k
        constexpr int CPR = BK / VEC;  // chunks per row
#pragma unroll
        for (int c = tid; c < 64 * CPR; c += NT) {
            const int row = c / CPR, kc = (c % CPR) * VEC;
            const int gr = r0 + row, gk = k0 + kc;
            vec_t v = {};
            if (gr < R && gk < K) v = *reinterpret_cast<const vec_t*>(base + (int64_t)gr * rs + gk);  // K % VEC == 0
            *reinterpret_cast<vec_t*>(lds + row * LDK + kc) = v;
        }
    } else if (mode == 2) {
        // 16-byte chunks along rows, transposed while written
        constexpr int CPL = 64 / VEC;  // chunks per k-line
#pragma unroll
        for (int c = tid; c < BK * CPL; c += NT) {
            const int kl = c / CPL, rc = (c % CPL) * VEC;
            const int gr = r0 + rc, gk = k0 + kl;
            vec_t v = {};
            if (gr < R && gk < K) v = *reinterpret_cast<const vec_t*>(base + (int64_t)gk * ks + gr);  // R % VEC == 0
            const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
            for (int i = 0; i < VEC; ++i) lds[(rc + i) * LDK + kl] = e[i];
        }
    } else {
        for (int c = tid; c < 64 * BK; c += NT) {
            const int row = c / BK, kk = c % BK;
            const int gr = r0 + row, gk = k0 + kk;
            T v = from_f32<T>(0.0f);
            if (gr < R && gk < K) v = base[(int64_t)gr * rs + (int64_t)gk * ks];
            lds[row * LDK + kk] = v;
        }
    }
}

template <typename T> __device__ __forceinline__ float act_apply(int act, float v, float aux) {
    switch (act) {
        case XGGM_ACT_GELU: return gelu_f(v);
        case XGGM_ACT_SIGMOID: return sigmoid_f(v);
        case XGGM_ACT_TANH: return tanhf(v);
        case XGGM_ACT_GELU_GRAD: return v * gelu_grad_f(aux);
        default: return v;
    }
}

// one 16x16 accumulator tile -> C: lane holds column `col`, rows row0 .. row0+3
template <typename T>
__device__ __forceinline__ float epilogue_tile(const GemmArgs& g, const float4_t& acc, int row0, int col, int bz) {
    if (col >= g.N) return 0.f;
    const int64_t coff = (int64_t)bz * g.c_bs;
    const float bias = g.bias ? g.bias[col] : 0.0f;
    float csum = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = row0 + r;
        if (row >= g.M) continue;
        const int64_t idx = coff + (int64_t)row * g.ldc + col;
        float v = g.alpha * acc[r] + bias;
        if (g.preact) reinterpret_cast<T*>(g.preact)[idx] = from_f32<T>(v);
        float aux = 0.0f;
        if (g.act == XGGM_ACT_GELU_GRAD) aux = to_f32(reinterpret_cast<const T*>(g.aux)[idx]);
        if (g.act != XGGM_ACT_NONE) {
            // forward activations act on the value as stored (bf16-rounded pre-activation)
            if (g.preact) v = round_to<T>(v);
            v = act_apply<T>(g.act, v, aux);
        }
        if (g.residual) v += to_f32(reinterpret_cast<const T*>(g.residual)[idx]);
        if (g.c_f32) {
            float* c = reinterpret_cast<float*>(g.C) + idx;
            *c = g.accumulate ? (*c + v) : v;
        } else {
            T* c = reinterpret_cast<T*>(g.C) + idx;
            *c = from_f32<T>(g.accumulate ? (to_f32(*c) + v) : v);
        }
        csum += v;
    }
    return csum;  // this lane's column over its (valid) rows: folded per 32-row block by the caller
}

template <typename T> __global__ __launch_bounds__(NT) void gemm_kernel(GemmArgs g) {
    constexpr int BK = Tile<T>::BK, LDK = Tile<T>::LDK;
    __shared__ __attribute__((aligned(16))) T As[BM * LDK];
    __shared__ __attribute__((aligned(16))) T Bs[BN * LDK];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = (wid >> 1) * 32, wn = (wid & 1) * 32;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN, bz = blockIdx.z;
    const T* A = reinterpret_cast<const T*>(g.A) + (int64_t)bz * g.a_bs;
    const T* B = reinterpret_cast<const T*>(g.B) + (int64_t)bz * g.b_bs;

    float4_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fq = lane >> 4;
    for (int k0 = 0; k0 < g.K; k0 += BK) {
        stage_tile<T>(As, A, g.a_rs, g.a_ks, m0, k0, g.M, g.K, g.a_mode, tid);
        stage_tile<T>(Bs, B, g.b_ns, g.b_ks, n0, k0, g.N, g.K, g.b_mode, tid);
        __syncthreads();
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int ks = 0; ks < BK; ks += 32) {
                short8_t a[2], b[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    a[i] = *reinterpret_cast<const short8_t*>(As + (wm + i * 16 + fr) * LDK + ks + fq * 8);
                    b[i] = *reinterpret_cast<const short8_t*>(Bs + (wn + i * 16 + fr) * LDK + ks + fq * 8);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, a[i]),
                            __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, b[j]), acc[i][j], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < BK; ks += 4) {
                float a[2], b[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    a[i] = As[(wm + i * 16 + fr) * LDK + ks + fq];
                    b[i] = Bs[(wn + i * 16 + fr) * LDK + ks + fq];
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // epilogue: C/D layout of the 16x16 MFMA: col = lane & 15, row = 4 * (lane >> 4) + reg
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        float cs = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) cs += epilogue_tile<T>(g, acc[i][j], m0 + wm + i * 16 + fq * 4, n0 + wn + j * 16 + fr, bz);
        if (g.colsum) {
            // a wave owns one block of 32 output rows: its column sums are folded over the four row groups (fixed
            // order) and stored by ONE lane per column -- a partial row nobody else writes
            cs += __shfl_xor(cs, 16, 64);
            cs += __shfl_xor(cs, 32, 64);
            const int col = n0 + wn + j * 16 + fr, slot = (m0 + wm) >> 5;
            if (fq == 0 && col < g.N && m0 + wm < g.M)
                g.colsum[((int64_t)bz * ((g.M + 31) >> 5) + slot) * g.N + col] = cs;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Fast bf16 path: BM x BN x 64 tiles, 4 waves (2x2), double-buffered LDS fed by a register
// prefetch of the next k-tile (global loads of tile t+1 are in flight while tile t runs on the
// matrix cores; one barrier per k-tile).  An operand whose reduction index is contiguous in
// memory ("k-major": activations in forward, weights in forward) is stored [row][64] with an XOR
// swizzle of its 16-byte chunks (chunk ^= (row >> 1) & 7: conflict-free ds_read_b128 for the
// 16x16x32 fragment).  An operand whose ROW index is contiguous ("r-major": weights in dgrad, both
// operands in wgrad) is stored as it arrives, [64 k][rows + 16 pad], and transposed by the LDS
// itself on the way out with ds_read_b64_tr_b16 -- no scalar transposing writes.
typedef __attribute__((ext_vector_type(4))) short short4_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;

template <int R, bool KMAJ, int NTH = NT> struct OpLds {
    static constexpr int LD = KMAJ ? 64 : (R + 16);  // elements per LDS row
    static constexpr int ELEMS = KMAJ ? R * 64 : 64 * (R + 16);
    static constexpr int NCH = R * 8 / NTH;  // 16-byte chunks per thread per k-tile (R rows x 8 chunks over NTH threads)
};

// Branch-free tile load through a buffer descriptor: the hardware range check returns zeros for
// the chunks beyond K (their offset is pushed past num_records), rows beyond the edge are clamped
// to the last valid row (they only feed output rows/columns the epilogue never stores).  No select
// on a pointer and no predicated load: hipcc rewrites both into branches with s_waitcnt vmcnt(0)
// behind them, which serialises the k-loop on memory latency.  Nothing depends on a load until its
// registers are copied to LDS D tiles later, so the loads stay in flight behind counted vmcnt(N).
typedef __attribute__((ext_vector_type(4))) unsigned int uint4_t;
constexpr int OOB_OFFSET = 0x7ffffff0;

struct OpSrc {
    __amdgpu_buffer_rsrc_t rsrc;
    int rs, ks;  // element strides of the row / reduction index
};

// `base`, `bytes` must be wave-uniform; readfirstlane makes that provable (a descriptor the
// compiler believes divergent is loaded through a waterfall loop per memory op)
__device__ __forceinline__ OpSrc make_src(const bf16* base, int64_t bytes, int64_t rs, int64_t ks) {
    const uint64_t a = reinterpret_cast<uint64_t>(base);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    const int nb = __builtin_amdgcn_readfirstlane((int)bytes);
    OpSrc s;
    s.rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>((uint64_t)lo | ((uint64_t)hi << 32)), 0, nb, 0x00020000);
    s.rs = __builtin_amdgcn_readfirstlane((int)rs);
    s.ks = __builtin_amdgcn_readfirstlane((int)ks);
    return s;
}

template <int R, bool KMAJ, int ES = 2, int NTH = NT>  // ES: bytes per element (2 = bf16; 1 = fp8, k-major operands only)
__device__ __forceinline__ void fast_load(short8_t (&reg)[R * 8 / NTH], const OpSrc& src, int r0, int k0, int Rtot, int K, int tid) {
    static_assert(ES == 2 || KMAJ, "fp8 operands are k-contiguous");
#pragma unroll
    for (int i = 0; i < R * 8 / NTH; ++i) {
        const int c = tid + NTH * i;
        int off;
        if (KMAJ) {
            const int row = min(r0 + (c >> 3), Rtot - 1), kc = k0 + (c & 7) * (16 / ES);
            off = kc < K ? (row * src.rs + kc) * ES : OOB_OFFSET;
        } else {
            const int kl = k0 + c / (R / 8), rc = min(r0 + (c % (R / 8)) * 8, Rtot - 8);
            off = kl < K ? (kl * src.ks + rc) * 2 : OOB_OFFSET;
        }
        reg[i] = __builtin_bit_cast(short8_t, __builtin_amdgcn_raw_buffer_load_b128(src.rsrc, off, 0, 0));
    }
}

template <int R, bool KMAJ, int NTH = NT>
__device__ __forceinline__ void fast_store(bf16* lds, const short8_t (&reg)[R * 8 / NTH], int tid, int k0, int K, int tail) {
#pragma unroll
    for (int i = 0; i < R * 8 / NTH; ++i) {
        const int c = tid + NTH * i;
        if (KMAJ) {
            const int row = c >> 3, kc = c & 7;
            short8_t v = reg[i];
            if (tail) {  // K % 8 != 0 (K even): zero what the chunk straddling K read beyond it
                const int nv = K - (k0 + kc * 8);  // valid elements of this chunk (>= 8: all, <= 0: loaded as zeros)
                uint4_t d = __builtin_bit_cast(uint4_t, v);
#pragma unroll
                for (int j = 0; j < 4; ++j) d[j] = (2 * j < nv) ? d[j] : 0u;
                v = __builtin_bit_cast(short8_t, d);
            }
            *reinterpret_cast<short8_t*>(lds + row * 64 + ((kc ^ ((row >> 1) & 7)) << 3)) = v;
        } else {
            const int kl = c / (R / 8), rc = (c % (R / 8)) * 8;
            // k-rows 0-3 <-> 4-7 swapped in every odd block of 8 k-rows: see fast_frag
            *reinterpret_cast<short8_t*>(lds + (kl ^ ((kl >> 1) & 4)) * (R + 16) + rc) = reg[i];
        }
    }
}

// fragment of the 16 rows starting at `row0` for k-step `ks` (0 or 32): lane (fr = lane & 15,
// fq = lane >> 4) gets row row0 + fr, k = ks + 8 fq .. + 7
template <int R, bool KMAJ>
__device__ __forceinline__ bf16x8_t fast_frag(const bf16* lds, int row0, int ks, int lane) {
    const int fr = lane & 15, fq = lane >> 4;
    if (KMAJ) {
        const int row = row0 + fr;
        const int chunk = ((ks >> 3) + fq) ^ ((row >> 1) & 7);
        return __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const short8_t*>(lds + row * 64 + (chunk << 3)));
    } else {
        // ds_read_b64_tr_b16: within a 16-lane group, lane 4q+p supplies the address of k-row q,
        // columns 4p..4p+3 of a 4 x 16 block and receives column (lane & 15) of the 4 k-rows
        const int q = fr >> 2, p = fr & 3;
        // Banks of a transposing read: lanes l and l + 16 of a 32-lane half conflict when their 32-byte row pieces
        // share a bank.  k-rows 8 apart do (8 * 160 B and 8 * 288 B are multiples of the 256-byte bank row), and
        // that is exactly what fq = 0 / 1 (2 / 3) read.  k-rows 4 apart sit in the other half of the bank row, so
        // the odd 8-row blocks are stored with their halves swapped and read back swapped: 2-way -> conflict-free
        // (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of the wgrad form 33 % -> 1 %; -0.08 ms per iteration).
        const int sw = (fq & 1) * 4;
        const bf16* a0 = lds + (ks + 8 * fq + sw + q) * (R + 16) + row0 + 4 * p;
        const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) short4_t*)(a0));
        const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) short4_t*)(a0 + (4 - 2 * sw) * (R + 16)));
        short8_t v;
        v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
        v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
        return __builtin_bit_cast(bf16x8_t, v);
    }
}

// e4m3 operands on the block-scaled matrix instruction.  v_mfma_f32_16x16x32_fp8_fp8 runs at the bf16 rate
// (MI355X_MICROARCH.md, Matrix cores); v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 operands does 4 x the K in twice the
// cycles: twice the bf16 rate, and ONE instruction where the plain form needs four.  A lane supplies 32 bytes per operand:
// row (lane & 15), k-bytes [32 q, 32 q + 32) with q = lane >> 4 (tools/micro/mfma_scale_check.hip: exact against the
// host's dot products with this assignment; a reduction only needs A and B to agree on it).  In the k-major LDS image
// (128-byte rows, 16-byte chunks XORed by (row >> 1) & 7) that is chunks 2 q and 2 q + 1 of the row: two ds_read_b128.
// The block scales are E8M0 bytes, 0x7f = 2^0: the per-tensor scales stay fp32 factors of the epilogue.
typedef int int8v_t __attribute__((ext_vector_type(8)));
typedef int int4w_t __attribute__((ext_vector_type(4)));
template <int R> __device__ __forceinline__ int8v_t frag_e4m3(const bf16* lds, int row0, int lane) {
    const int fr = lane & 15, fq = lane >> 4, row = row0 + fr, s = (row >> 1) & 7;
    const int4w_t lo = *reinterpret_cast<const int4w_t*>(lds + row * 64 + (((2 * fq) ^ s) << 3));
    const int4w_t hi = *reinterpret_cast<const int4w_t*>(lds + row * 64 + (((2 * fq + 1) ^ s) << 3));
    return (int8v_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
__device__ __forceinline__ float4_t mfma_e4m3_k128(const int8v_t& first, const int8v_t& second, const float4_t& c) {
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(first, second, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
}

// workgroup barrier that waits for this wave's LDS traffic only.  __syncthreads() also waits
// vmcnt(0), i.e. for the global prefetches that are supposed to stay in flight across it.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// ---- epilogue of the tuned kernels -------------------------------------------------------------
// The accumulators of a half tile (BM/2 rows) are parked in LDS as fp32, then every thread walks
// 8-element row chunks: bias, activation, residual, pre-activation save and the store are done on
// 16/32-byte vectors along the row (full lines instead of 64 scattered 2-byte stores per lane), in a
// rolled loop -- a fully unrolled per-element epilogue with its activation switch made the code
// object ~100 KB and the instruction fetch of that was most of the kernel's fixed cost.
template <int BM, int BN, int TM, int TN, int W = 4>
__device__ __forceinline__ void epilogue_staged(const GemmArgs& g, const float4_t (&acc)[TM][TN], int m0, int n0, int bz,
                                             float* stage, int wm, int wn) {
    constexpr int NT = 64 * W;             // threads of this workgroup (shadows the file-wide 256)
    constexpr int WMH = W / 4;             // waves along M inside one half tile (W / 2 waves along M in all)
    constexpr int LDS_LD = BN + 4;
    constexpr int CPR = BN / 8;           // 8-element chunks per row
    constexpr int HALF = BM / 2;
    constexpr int ITER = HALF * CPR / NT;  // chunks per thread per half tile
    constexpr int RPI = NT / CPR;          // rows between a thread's consecutive chunks
    constexpr int BATCH = ITER < 2 ? ITER : 2;  // chunks whose global loads are issued together
    static_assert(HALF * CPR % NT == 0 && NT % CPR == 0 && ITER % BATCH == 0, "epilogue tiling");
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int64_t coff = (int64_t)bz * g.c_bs;
    const bool vec_ok = (g.ldc % 8 == 0) && (g.N % 8 == 0) && (reinterpret_cast<uintptr_t>(g.C) % 16 == 0) &&
                        (!g.residual || reinterpret_cast<uintptr_t>(g.residual) % 16 == 0) &&
                        (!g.preact || reinterpret_cast<uintptr_t>(g.preact) % 16 == 0) &&
                        (!g.aux || reinterpret_cast<uintptr_t>(g.aux) % 16 == 0) && (coff % 8 == 0);
    bf16* pre = reinterpret_cast<bf16*>(g.preact);
    const bf16* aux = reinterpret_cast<const bf16*>(g.aux);
    const bf16* res = reinterpret_cast<const bf16*>(g.residual);
    // a thread's chunks all sit in one column group (NT is a multiple of CPR): its bias is loaded once
    const int lc = (tid % CPR) * 8, col = n0 + lc, lr0 = tid / CPR;
    const bool col_ok = col < g.N;
    float bias[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bias[e] = (g.bias && col + e < g.N) ? g.bias[col + e] : 0.f;
    // sum of squares of what this thread stores, per 64-row block of the tile (gradient-norm slots, see the end)
    constexpr int RS = BM >= 64 ? BM / 64 : 1, CS = BN >= 64 ? BN / 64 : 1;
    static_assert(BM % 64 == 0 || BM < 64, "norm slots: whole 64-row blocks per tile");
    float sq_acc[RS];
#pragma unroll
    for (int r = 0; r < RS; ++r) sq_acc[r] = 0.f;
    // 64-row block (of the tile) a chunk's row lies in: with HALF = 64 that is the half index; HALF = 96 (BM = 192) cuts
    // the blocks differently
    auto sq_add = [&](int tile_row, float v) {
#pragma unroll
        for (int r = 0; r < RS; ++r)
            if (RS == 1 || (tile_row >> 6) == r) sq_acc[r] += v;
    };
    const Q8 qs(g.c8 ? g.c8_qscale : nullptr);
    const float q8 = qs.q;
    float amax8 = 0.f;
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
        if ((wid >> 1) / WMH == h) {
            const int hr = ((wid >> 1) % WMH) * (HALF / WMH);  // this wave's first row inside the half tile
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    *reinterpret_cast<float4_t*>(stage + (hr + i * 16 + fr) * LDS_LD + wn + j * 16 + fq * 4) = acc[i][j];
        }
        lds_barrier();  // LDS only: global stores of the previous half stay in flight
        if (h == 0) STAMP(g, 7);
        if (ABLATE(g, 1)) {
            if (g.M < 0) reinterpret_cast<float*>(g.C)[tid] = stage[tid];  // never true: keeps the staging alive
        } else if (vec_ok) {
#pragma unroll 1
            for (int it0 = 0; it0 < ITER; it0 += BATCH) {
                // issue every global load of the batch, then consume: one memory round trip per batch
                short8_t rres[BATCH], raux[BATCH], rprev[BATCH];
                float4 p0[BATCH], p1[BATCH];
                int64_t idx[BATCH];
                bool ok[BATCH];
#pragma unroll
                for (int b = 0; b < BATCH; ++b) {
                    const int row = m0 + h * HALF + lr0 + (it0 + b) * RPI;
                    ok[b] = col_ok && row < g.M;
                    idx[b] = coff + (int64_t)row * g.ldc + col;
                    if (ok[b]) {
                        if (res) rres[b] = *reinterpret_cast<const short8_t*>(res + idx[b]);
                        if (g.act == XGGM_ACT_GELU_GRAD) raux[b] = *reinterpret_cast<const short8_t*>(aux + idx[b]);
                        if (g.accumulate) {
                            if (g.c_f32) {
                                const float* c = reinterpret_cast<const float*>(g.C) + idx[b];
                                p0[b] = *reinterpret_cast<const float4*>(c);
                                p1[b] = *reinterpret_cast<const float4*>(c + 4);
                            } else {
                                rprev[b] = *reinterpret_cast<const short8_t*>(reinterpret_cast<const bf16*>(g.C) + idx[b]);
                            }
                        }
                    }
                }
#pragma unroll
                for (int b = 0; b < BATCH; ++b) {
                    if (!ok[b]) continue;
                    float* sp = stage + (lr0 + (it0 + b) * RPI) * LDS_LD + lc;
                    float v[8];
                    {
                        const float4 a = *reinterpret_cast<const float4*>(sp);
                        const float4 c4 = *reinterpret_cast<const float4*>(sp + 4);
                        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = c4.x; v[5] = c4.y; v[6] = c4.z; v[7] = c4.w;
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = g.alpha * v[e] + bias[e];
                    if (pre) {
                        short8_t pv;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const bf16 t = __float2bfloat16(v[e]);
                            pv[e] = __builtin_bit_cast(short, t);
                            v[e] = __bfloat162float(t);  // the activation sees the value as stored
                        }
                        *reinterpret_cast<short8_t*>(pre + idx[b]) = pv;
                    }
                    if (g.act == XGGM_ACT_GELU) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = gelu_f(v[e]);
                    } else if (g.act == XGGM_ACT_GELU_GRAD) {
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            v[e] *= gelu_grad_f(__bfloat162float(__builtin_bit_cast(bf16, (short)raux[b][e])));
                    } else if (g.act != XGGM_ACT_NONE) {
#pragma unroll 1
                        for (int e = 0; e < 8; ++e) v[e] = act_apply<bf16>(g.act, v[e], 0.f);
                    }
                    if (res) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += __bfloat162float(__builtin_bit_cast(bf16, (short)rres[b][e]));
                    }
                    if (g.colsum) {  // park the final values for the column pass below
                        *reinterpret_cast<float4*>(sp) = make_float4(v[0], v[1], v[2], v[3]);
                        *reinterpret_cast<float4*>(sp + 4) = make_float4(v[4], v[5], v[6], v[7]);
                    }
                    if (g.c8) {
                        // the operand of the next fp8 product leaves its producer as e4m3 (v_cvt_pk_fp8_f32 saturates
                        // to +-448 and rounds to nearest even), no quantisation pass
#pragma unroll
                        for (int e = 0; e < 8; ++e) amax8 = fmaxf(amax8, fabsf(v[e]));
                        *reinterpret_cast<int2*>(g.c8 + idx[b]) =
                            make_int2(pack4_e4m3(v[0], v[1], v[2], v[3], q8), pack4_e4m3(v[4], v[5], v[6], v[7], q8));
                    }
                    if (g.c_f32) {
                        float* c = reinterpret_cast<float*>(g.C) + idx[b];
                        float4 o0 = make_float4(v[0], v[1], v[2], v[3]), o1 = make_float4(v[4], v[5], v[6], v[7]);
                        if (g.accumulate) {
                            o0.x += p0[b].x; o0.y += p0[b].y; o0.z += p0[b].z; o0.w += p0[b].w;
                            o1.x += p1[b].x; o1.y += p1[b].y; o1.z += p1[b].z; o1.w += p1[b].w;
                        }
                        if (g.sqsum)
                            sq_add(h * HALF + lr0 + (it0 + b) * RPI, (o0.x * o0.x + o0.y * o0.y) + (o0.z * o0.z + o0.w * o0.w) +
                                                                         (o1.x * o1.x + o1.y * o1.y) + (o1.z * o1.z + o1.w * o1.w));
                        // fp32 outputs are weight gradients (860 MB per pass, next read by the norm / update
                        // passes from HBM anyway): non-temporal, so they do not evict activations and weights
                        // (same-box A/B against plain stores, tools/ab.sh: 12.26 vs 12.39 ms per iteration)
                        typedef float __attribute__((ext_vector_type(4))) f4;
                        __builtin_nontemporal_store((f4){o0.x, o0.y, o0.z, o0.w}, reinterpret_cast<f4*>(c));
                        __builtin_nontemporal_store((f4){o1.x, o1.y, o1.z, o1.w}, reinterpret_cast<f4*>(c + 4));
                    } else {
                        if (g.accumulate) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] += __bfloat162float(__builtin_bit_cast(bf16, (short)rprev[b][e]));
                        }
                        short8_t ov;
#pragma unroll
                        for (int e = 0; e < 8; ++e) ov[e] = __builtin_bit_cast(short, __float2bfloat16(v[e]));
                        *reinterpret_cast<short8_t*>(reinterpret_cast<bf16*>(g.C) + idx[b]) = ov;
                    }
                }
            }
        } else {
            // unaligned output (odd N / ldc): element-wise
#pragma unroll 1
            for (int it = 0; it < ITER; ++it) {
                const int lr = lr0 + it * RPI, row = m0 + h * HALF + lr;
                if (row >= g.M || !col_ok) continue;
                const int nval = min(8, g.N - col);
                const int64_t idx = coff + (int64_t)row * g.ldc + col;
#pragma unroll 1
                for (int e = 0; e < nval; ++e) {
                    float x = g.alpha * stage[lr * LDS_LD + lc + e] + (g.bias ? g.bias[col + e] : 0.f);
                    if (pre) {
                        pre[idx + e] = __float2bfloat16(x);
                        x = round_to<bf16>(x);
                    }
                    if (g.act != XGGM_ACT_NONE)
                        x = act_apply<bf16>(g.act, x, g.act == XGGM_ACT_GELU_GRAD ? __bfloat162float(aux[idx + e]) : 0.f);
                    if (res) x += __bfloat162float(res[idx + e]);
                    if (g.colsum) stage[lr * LDS_LD + lc + e] = x;
                    if (g.c_f32) {
                        float* c = reinterpret_cast<float*>(g.C) + idx + e;
                        const float o = g.accumulate ? (*c + x) : x;
                        *c = o;
                        if (g.sqsum) sq_add(h * HALF + lr, o * o);
                    } else {
                        bf16* c = reinterpret_cast<bf16*>(g.C) + idx + e;
                        *c = __float2bfloat16(g.accumulate ? (__bfloat162float(*c) + x) : x);
                    }
                }
            }
        }
        lds_barrier();  // every chunk of the staged half has been read (and parked back for colsum)
        if (h == 0) STAMP(g, 3);
        if (g.colsum) {
            // bias gradient: column sums of this half tile's stored values per block of 32 output rows, each written
            // by one thread in a fixed order into a partial row of its own (summed over the row blocks by
            // xggm_partial_reduce_batch): no floating-point atomics, the same bits whatever the scheduling
            constexpr int SUB = HALF / 32;
            static_assert(HALF % 32 == 0 && NT % BN == 0, "column sums: 32-row blocks, whole column sets per pass");
            const int cl = tid % BN;
            for (int sub = tid / BN; sub < SUB; sub += NT / BN) {
                const int r0 = m0 + h * HALF + sub * 32;
                if (n0 + cl < g.N && r0 < g.M) {
                    const int rows = min(32, g.M - r0);
                    float sacc = 0.f;
                    for (int r = 0; r < rows; ++r) sacc += stage[(sub * 32 + r) * LDS_LD + cl];
                    g.colsum[((int64_t)bz * ((g.M + 31) >> 5) + (r0 >> 5)) * g.N + n0 + cl] = sacc;
                }
            }
            lds_barrier();
        }
    }
    if (g.c8 && g.c8_amax) {
        // non-negative floats order like their bit patterns: one integer atomic per workgroup.  The loop above ended
        // with a barrier: the staging LDS is free.
        const float wv = wave_max(amax8);
        if (lane == 0) stage[wid] = wv;
        lds_barrier();
        if (tid == 0) {
            float bm = stage[0];
#pragma unroll
            for (int w = 1; w < W; ++w) bm = fmaxf(bm, stage[w]);
            if (bm > qs.thr) amax_record(g.c8_amax, g.amax_slots, (int)(blockIdx.x + blockIdx.y * gridDim.x), bm);
        }
        lds_barrier();
    }
    if (g.sqsum && g.c_f32) {
        // one slot per 64 x 64 block of the output, written by exactly one workgroup, summed in a fixed order: the
        // clip norm of the weight gradients without another pass over them (lxrt/optimization.py clip_grad_norm_).
        // The loop above ended with a barrier: the staging LDS is free.
        const int cs = lc / 64;  // a thread's chunks all sit in one 64-column block
#pragma unroll
        for (int r = 0; r < RS; ++r)
#pragma unroll
            for (int c = 0; c < CS; ++c) {
                const float v = wave_sum(cs == c ? sq_acc[r] : 0.f);
                if (lane == 0) stage[(r * CS + c) * W + wid] = v;
            }
        lds_barrier();
        if (tid < RS * CS) {
            const int r = tid / CS, c = tid % CS;
            const int row = m0 + r * 64, cl = n0 + c * 64;
            const int64_t sc = (g.N + 63) / 64, sr = (g.M + 63) / 64;
            if (row < g.M && cl < g.N) {
                float t = 0.f;
#pragma unroll
                for (int w = 0; w < W; w += 4)  // fixed order
                    t += (stage[tid * W + w] + stage[tid * W + w + 1]) + (stage[tid * W + w + 2] + stage[tid * W + w + 3]);
                g.sqsum[(int64_t)bz * sr * sc + (int64_t)(row / 64) * sc + cl / 64] = t;
            }
        }
    }
}

// ---- register epilogue ----------------------------------------------------------------------------
// The same arithmetic straight out of the accumulators: with the MFMA operands swapped a lane already holds four
// CONSECUTIVE columns of one row per 16 x 16 block (8 bytes of bf16, 16 bytes of fp32), so bias, activation and the
// stores need no trip through LDS.  In-kernel stamps (tools/gemm_phase_report.py, tools/gemm_epi_ablate.py) put the
// staged epilogue of a 128 x 64 tile at 6.1 k cycles, a quarter of the tile's life, and showed that removing its
// global stores changes almost nothing: every SIMD holds two or three waves that reach the epilogue together, so
// the epilogue costs what its INSTRUCTIONS cost to issue.  Hence two lean straight-line kinds, each a few dozen
// instructions per 16 x 16 block, for what the step launches most -- and the staged, feature-complete walk for the rest:
//   kind 0: C = alpha * acc + bias, bf16 or fp32 (weight gradients: non-temporal, optional norm slots)
//   kind 1: pre = bf16(alpha * acc + bias), C = gelu(pre), optional e4m3 copy          (the FFN's first product)
template <int BM, int BN, int W> __device__ __forceinline__ int epilogue_kind(const GemmArgs& g, int bz) {
    const int64_t coff = (int64_t)bz * g.c_bs;
    // the register epilogue adds a wave's squares into ONE norm slot: its tile must lie inside one 64 x 64 block
    if (g.sqsum && !(BM / (W / 2) <= 64 && BN / 2 <= 64)) return 2;
    auto al = [](const void* q, uintptr_t a) { return reinterpret_cast<uintptr_t>(q) % a == 0; };
    if (g.N % 4 || g.ldc % 4 || coff % 4 || g.colsum || g.residual || g.accumulate || !al(g.C, g.c_f32 ? 16 : 8) ||
        (int64_t)g.M * g.ldc >= (1ll << 31) || (g.sqsum && !g.c_f32))
        return 2;
    if (g.act == XGGM_ACT_NONE && !g.preact && !g.c8) return 0;
    if (g.act == XGGM_ACT_GELU && g.preact && !g.c_f32 && !g.sqsum && al(g.preact, 8) && al(g.c8, 4)) return 1;
    return 2;
}

template <int BM, int BN, int TM, int TN, int W, int KIND>
__device__ __forceinline__ void epilogue_direct(const GemmArgs& g, const float4_t (&acc)[TM][TN], int m0, int n0, int bz,
                                                float* stage, int wm, int wn) {
    typedef short short4v __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int64_t coff = (int64_t)bz * g.c_bs;
    float4_t bias[TN];
    const int col0 = n0 + wn + fq * 4, row0 = m0 + wm + fr;
    STAMP(g, 7);
#pragma unroll
    for (int j = 0; j < TN; ++j)
        bias[j] = (g.bias && col0 + j * 16 < g.N) ? *reinterpret_cast<const float4_t*>(g.bias + col0 + j * 16)
                                                  : (float4_t){0.f, 0.f, 0.f, 0.f};
    const Q8 qs(KIND == 1 && g.c8 ? g.c8_qscale : nullptr);
    const float q8 = qs.q, alpha = g.alpha;
    float amax8 = 0.f, sq = 0.f;
#ifdef XGGM_STAMP
    if (g.ablate & 0x10000) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // trace: when the bias has arrived
#endif
    TRACE(g, 1);
    const int off0 = row0 * g.ldc + col0;  // 32-bit offsets from the (batch) base: epilogue_kind checked the range
    float* cf = reinterpret_cast<float*>(g.C) + coff;
    bf16* cb = reinterpret_cast<bf16*>(g.C) + coff;
    bf16* pre = reinterpret_cast<bf16*>(g.preact) + coff;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        if (i == 1) STAMP(g, 3);
        TRACE(g, 2 + i);
        if (row0 + i * 16 >= g.M) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            if (col0 + j * 16 >= g.N) continue;
            const int off = off0 + i * 16 * g.ldc + j * 16;
            float4_t v = alpha * acc[i][j] + bias[j];
            if (KIND == 1) {
                short4v pv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bf16 t = __float2bfloat16(v[e]);
                    pv[e] = __builtin_bit_cast(short, t);
                    v[e] = gelu_f(__bfloat162float(t));  // the activation sees the value as stored
                }
                *reinterpret_cast<short4v*>(pre + off) = pv;
                if (g.c8) {
                    amax8 = fmaxf(fmaxf(amax8, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
                    *reinterpret_cast<int*>(g.c8 + coff + off) = pack4_e4m3(v[0], v[1], v[2], v[3], q8);
                }
            }
            if (KIND == 0 && g.c_f32) {
                sq += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
                __builtin_nontemporal_store(v, reinterpret_cast<float4_t*>(cf + off));
            } else {
                short4v ov;
#pragma unroll
                for (int e = 0; e < 4; ++e) ov[e] = __builtin_bit_cast(short, __float2bfloat16(v[e]));
                if (!ABLATE(g, 4) || ov[0] == 12345) *reinterpret_cast<short4v*>(cb + off) = ov;
            }
        }
    }
    // (the k-loop ended with a barrier: LDS is free)
    if (KIND == 1 && g.c8 && g.c8_amax) {
        const float wv = wave_max(amax8);
        if (lane == 0) stage[wid] = wv;
        lds_barrier();
        if (tid == 0) {
            float bm = stage[0];
#pragma unroll
            for (int w = 1; w < W; ++w) bm = fmaxf(bm, stage[w]);
            if (bm > qs.thr) amax_record(g.c8_amax, g.amax_slots, (int)(blockIdx.x + blockIdx.y * gridDim.x), bm);
        }
    }
    if (KIND == 0 && g.sqsum) {
        // norm slots (see epilogue_staged): a wave's tile lies inside ONE 64 x 64 block of the output; the waves of
        // a block are added in a fixed order
        constexpr int RS = BM >= 64 ? BM / 64 : 1, CS = BN >= 64 ? BN / 64 : 1;
        const float ws = wave_sum(sq);
        if (lane == 0) stage[wid] = ws;
        lds_barrier();
        if (tid < RS * CS) {
            const int r = tid / CS, c = tid % CS;
            const int row = m0 + r * 64, cl = n0 + c * 64;
            if (row < g.M && cl < g.N) {
                float t = 0.f;
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    // (epilogue_kind sends tiles whose waves straddle 64 x 64 blocks to the staged walk)
                    const int wr = (w >> 1) * (BM / (W / 2)) / 64, wc = (w & 1) * (BN / 2) / 64;
                    if (wr == r && wc == c) t += stage[w];
                }
                const int64_t sc = (g.N + 63) / 64, sr = (g.M + 63) / 64;
                g.sqsum[(int64_t)bz * sr * sc + (int64_t)(row / 64) * sc + cl / 64] = t;
            }
        }
    }
}

// F8: both operands are OCP e4m3 bytes, k-contiguous.  A k-tile is then 128 elements -- the same 128 bytes per
// row, so loads, LDS image, swizzle and fragment reads are the bf16 ones byte for byte; a lane's 16-byte fragment
// feeds two v_mfma_f32_16x16x32_fp8_fp8 (8 bytes each).  Which 8 k-indices a lane supplies does not matter to a
// reduction as long as A and B agree, and both go through the same mapping.
// W = waves per workgroup: 4 (2 x 2) or 8 (4 along M x 2 along N; two waves per SIMD inside ONE workgroup: one wave's
// LDS phase runs under the other's MFMAs even when a CU holds a single tile, and a 128 x 128 tile moves a third
// fewer LDS bytes per flop than two 128 x 64 tiles -- the k-loop is bound by the LDS write path)
template <int BM, int BN, bool AK, bool BKM, int D, bool F8 = false, int W = 4>
__device__ __forceinline__ void gemm_kloop(const GemmArgs& g, int tile_m, int tile_n, int bz, bf16* fsm,
                                           float4_t (&acc)[BM / (8 * W)][BN / 32]) {
    static_assert(!F8 || (AK && BKM), "fp8 operands are k-contiguous");
    constexpr int NT = 64 * W;  // threads of this workgroup (shadows the file-wide 256)
    constexpr int ES = F8 ? 1 : 2, KT = 128 / ES;  // bytes per element, elements per k-tile
    // D = prefetch depth: D k-tiles of both operands are in flight in registers while one tile
    // is consumed from LDS.  These GEMMs are skinny (one k-chain per CU), so the k-loop would
    // otherwise run at one L2/HBM round trip per iteration.
    typedef OpLds<BM, AK, NT> LA;
    typedef OpLds<BN, BKM, NT> LB;
    constexpr int WM = W / 2;                        // waves along M (2 along N)
    constexpr int TM = BM / (16 * WM), TN = BN / 32;  // 16x16 tiles per wave (wave tile = BM/WM x BN/2)
    constexpr int STAGE = LA::ELEMS + LB::ELEMS;  // LDS buffer s: A at s*STAGE, B at s*STAGE + LA::ELEMS
    constexpr int UNR = (D % 2 == 0) ? D : 2 * D;  // unroll so that stage (t % D) and buffer (t & 1) are static

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = (wid >> 1) * (BM / WM), wn = (wid & 1) * (BN / 2);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    STAMP(g, 0);
    STAMP_HW(g);
    const bf16* A = reinterpret_cast<const bf16*>(reinterpret_cast<const char*>(g.A) + (int64_t)bz * g.a_bs * ES);
    const bf16* B = reinterpret_cast<const bf16*>(reinterpret_cast<const char*>(g.B) + (int64_t)bz * g.b_bs * ES);
    // bytes from the (batch) base to the end of the operand: last row / k-row start + its valid length
    const OpSrc sa = make_src(A, AK ? ((int64_t)(g.M - 1) * g.a_rs + (g.K + 7) / 8 * 8) * ES : ((int64_t)(g.K - 1) * g.a_ks + g.a_rows) * 2,
                              g.a_rs, g.a_ks);
    const OpSrc sb = make_src(B, BKM ? ((int64_t)(g.N - 1) * g.b_ns + (g.K + 7) / 8 * 8) * ES : ((int64_t)(g.K - 1) * g.b_ks + g.b_rows) * 2,
                              g.b_ns, g.b_ks);

#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

    short8_t ra[D][LA::NCH], rb[D][LB::NCH];
    // the k-loop runs a multiple of UNR tiles; tiles past K load zeros (see fast_load)
    const int nk = ((g.K + KT - 1) / KT + UNR - 1) / UNR * UNR;
#pragma unroll
    for (int s = 0; s < D; ++s) {
        fast_load<BM, AK, ES, NT>(ra[s], sa, m0, s * KT, g.a_rows, g.K, tid);
        fast_load<BN, BKM, ES, NT>(rb[s], sb, n0, s * KT, g.b_rows, g.K, tid);
    }
    fast_store<BM, AK, NT>(fsm, ra[0], tid, 0, g.K, F8 ? 0 : g.a_tail);
    fast_store<BN, BKM, NT>(fsm + LA::ELEMS, rb[0], tid, 0, g.K, F8 ? 0 : g.b_tail);
    lds_barrier();
    STAMP(g, 1);
    for (int t0 = 0; t0 < nk; t0 += UNR) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int t = t0 + u;
            const bf16* Ac = fsm + (u & 1) * STAGE;
            const bf16* Bc = Ac + LA::ELEMS;
            bf16* An = fsm + ((u + 1) & 1) * STAGE;
            // stage (t % D) was copied to LDS one step ago: refill it with tile t + D
            if (!KABLATE(g, 8)) {
                fast_load<BM, AK, ES, NT>(ra[u % D], sa, m0, (t + D) * KT, g.a_rows, g.K, tid);
                fast_load<BN, BKM, ES, NT>(rb[u % D], sb, n0, (t + D) * KT, g.b_rows, g.K, tid);
            }
            // (W == 8, measured and not kept: queueing the iteration's LDS writes ahead of its MFMAs -- slower, the
            // wave stalls on the prefetch's vmcnt in front of the matrix work, 14.97 -> 17.2 k cycles per 128 x 128 x 768
            // tile; running the two wave groups half an iteration apart with a barrier per half -- slower still,
            // 22.2 k: the iteration is a LATENCY chain (barrier, fragment reads, dependent MFMAs, writes), not an LDS
            // throughput limit, and a second barrier lengthens it.)
            if constexpr (F8) {  // one 128-deep step per k-tile on the block-scaled instruction
                int8v_t a8[TM], b8[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) a8[i] = frag_e4m3<BM>(Ac, wm + i * 16, lane);
#pragma unroll
                for (int j = 0; j < TN; ++j) b8[j] = frag_e4m3<BN>(Bc, wn + j * 16, lane);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = mfma_e4m3_k128(b8[j], a8[i], acc[i][j]);
            } else
#pragma unroll
            for (int ks = 0; ks < 64; ks += 32) {
                bf16x8_t a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = fast_frag<BM, AK>(KABLATE(g, 64) ? fsm : Ac, wm + (KABLATE(g, 64) ? 0 : i * 16), ks, lane);
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = fast_frag<BN, BKM>(KABLATE(g, 64) ? fsm : Bc, wn + (KABLATE(g, 64) ? 0 : j * 16), ks, lane);
                if (KABLATE(g, 32)) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) acc[i][0][0] += __builtin_bit_cast(float4_t, a[i])[0];
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[0][j][1] += __builtin_bit_cast(float4_t, b[j])[0];
                    continue;
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        // operands swapped: the 16x16 block comes out transposed, i.e. lane (fr, fq) holds
                        // row fr, columns 4 fq .. 4 fq + 3 -- four values that are contiguous in C
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
            }
            if (!KABLATE(g, 16)) {
                fast_store<BM, AK, NT>(An, ra[(u + 1) % D], tid, (t + 1) * KT, g.K, F8 ? 0 : g.a_tail);
                fast_store<BN, BKM, NT>(An + LA::ELEMS, rb[(u + 1) % D], tid, (t + 1) * KT, g.K, F8 ? 0 : g.b_tail);
            }
            if (!KABLATE(g, 128)) lds_barrier();
        }
    }
    // the k-loop ended with a barrier: LDS is free for the staged epilogue
    STAMP(g, 2);
}

// ---- k-loop for two k-major operands fed by LDS-DMA ------------------------------------------------------------
// tools/gemm_kloop_ablate.py (k-loop with parts removed): of the register-staged iteration above, the LDS STORES are the
// most expensive part -- a third of the k-loop's cycles (ds_write_b128 moves its 16 bytes per lane over the VGPR -> LDS
// path at ~79 B/clk per CU, shared by all waves) -- ahead of the fragment reads, the MFMAs and the global loads.
// `buffer_load_dwordx4 ... lds` writes LDS without passing through registers at all: no ds_write, no prefetch ring in
// VGPRs, no vmcnt-ordered store sequence in front of the barrier.  One wave-instruction fills 1 KB of consecutive LDS
// (64 lanes x 16 bytes) = 8 rows of the [row][64] image, so the XOR swizzle of the image moves to the SOURCE side: the
// lane that owns slot s of row r fetches k-chunk s ^ ((r >> 1) & 7).  NS LDS stages: the loads of tile t + NS - 1 are
// issued right after the barrier that frees the stage tile t - 1 was read from, and are waited for (counted vmcnt, then
// the barrier that makes every wave's part visible) NS - 1 iterations later.  Requires K % k-tile == 0 (no partial
// chunks to zero on the way); the forward products of the step all qualify.
// An r-major operand (row index contiguous: weights in dgrad, both operands in wgrad) cannot keep the padded
// [64 k][R + 16] image of the register path -- a wave-instruction's 1 KB spans several k-rows and would land in the
// padding.  Its LDS-DMA image is [64 k][R] with the 16-byte chunks of k-row k XORed by glds_rx(k): the eight k-rows a
// 32-lane group of ds_read_b64_tr_b16 touches (k = 8 fq + q, fq in {0, 1}, q < 4; +4 for the second read) then sit
// in eight different 32-byte bank groups (R = 128: 256-byte rows; R = 64: two k-rows per bank row, the parity of k
// picks the half), i.e. conflict-free without padding.
template <int R> __device__ __forceinline__ int glds_rx(int k) {
    // R = 192 (384-byte k-rows, 24 chunks): the XOR has to stay below 8 so that a chunk stays inside its aligned group
    // of eight; consecutive k-rows start 128 bytes apart modulo the 256-byte bank row, so k and k + 2 collide and the
    // swizzle separates them by bits 1 and 3 of k as in the 64-row case, plus bit 0 for the half-row pairs
    if (R % 128 != 0 && R >= 128) return (k & 1) | (((k >> 1) & 1) << 1) | (((k >> 3) & 1) << 2);
    return R >= 128 ? 2 * ((k & 3) | (((k >> 3) & 1) << 2)) : 2 * (((k >> 1) & 1) | (((k >> 3) & 1) << 1));
}

template <int R, bool KMAJ> struct GldsImg { static constexpr int ELEMS = R * 64; };

// byte offsets (inside one k-tile) of the chunks this thread moves: chunk c = tid + NTH * i lands at LDS chunk c
template <int R, bool KMAJ, int ES, int NTH>
__device__ __forceinline__ void glds_offsets(int (&v)[R * 8 / NTH], const OpSrc& src, int r0, int Rtot, int tid) {
#pragma unroll
    for (int i = 0; i < R * 8 / NTH; ++i) {
        const int c = tid + NTH * i;
        if (KMAJ) {
            const int row = c >> 3, kc = (c & 7) ^ ((row >> 1) & 7);
            v[i] = (min(r0 + row, Rtot - 1) * src.rs + kc * (16 / ES)) * ES;
        } else {
            const int kl = c / (R / 8), rc = ((c % (R / 8)) ^ glds_rx<R>(kl)) * 8;
            v[i] = (kl * src.ks + min(r0 + rc, Rtot - 8)) * 2;
        }
    }
}

template <int R, bool KMAJ>
__device__ __forceinline__ bf16x8_t glds_frag(const bf16* lds, int row0, int ks, int lane) {
    if (KMAJ) return fast_frag<R, true>(lds, row0, ks, lane);
    const int fr = lane & 15, fq = lane >> 4, q = fr >> 2, p = fr & 3;
    const int k0 = ks + 8 * fq + q, c0 = (row0 >> 3) + (p >> 1);
    const bf16* a0 = lds + k0 * R + ((c0 ^ glds_rx<R>(k0)) << 3) + (p & 1) * 4;
    const bf16* a1 = lds + (k0 + 4) * R + ((c0 ^ glds_rx<R>(k0 + 4)) << 3) + (p & 1) * 4;
    const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)(a0));
    const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)(a1));
    short8_t v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
    v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return __builtin_bit_cast(bf16x8_t, v);
}

// One LDS-DMA instruction: 64 lanes x 16 bytes from buffer offset voff (per lane) + soff (scalar) to LDS bytes
// [lds, lds + 1024).  Written as asm on purpose: for the builtin, hipcc's wait-count pass orders every LDS read it
// cannot prove disjoint behind the DMA -- it put an s_waitcnt vmcnt(0) between the loads of tile t + 2 and the
// ds_read_b64_tr_b16 of tile t, i.e. a full memory round trip into every iteration of the r-major loops (k-loop
// 53 k -> 86 k cycles on the FFN dgrad).  The waits these loads need are the counted ones in the loop below.
typedef int int4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void glds16(const int4v& rsrc, unsigned lds, int voff, int soff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(lds), "v"(voff), "s"(rsrc), "s"(soff)
                 : "memory");
}
__device__ __forceinline__ int4v raw_rsrc(const bf16* base, int64_t bytes) {
    const uint64_t a = reinterpret_cast<uint64_t>(base);
    int4v r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    r[1] = __builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32)) & 0xffff;
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;
    return r;
}

template <int BM, int BN, bool AK, bool BKM, bool F8, int W, int NS>
__device__ __forceinline__ void gemm_kloop_glds(const GemmArgs& g, int tile_m, int tile_n, int bz, bf16* fsm,
                                                float4_t (&acc)[BM / (8 * W)][BN / 32]) {
    static_assert(!F8 || (AK && BKM), "fp8 operands are k-contiguous");
    constexpr int NT = 64 * W;
    constexpr int ES = F8 ? 1 : 2, KT = 128 / ES;
    constexpr int NCA = BM * 8 / NT, NCB = BN * 8 / NT;  // LDS-DMA instructions per thread and k-tile
    constexpr int WM = W / 2;
    constexpr int TM = BM / (16 * WM), TN = BN / 32;
    constexpr int AEL = BM * 64, STAGE = (BM + BN) * 64;
    constexpr int G = NCA + NCB;

    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wid >> 1) * (BM / WM), wn = (wid & 1) * (BN / 2);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    STAMP(g, 0);
    STAMP_HW(g);
    const bf16* A = reinterpret_cast<const bf16*>(reinterpret_cast<const char*>(g.A) + (int64_t)bz * g.a_bs * ES);
    const bf16* B = reinterpret_cast<const bf16*>(reinterpret_cast<const char*>(g.B) + (int64_t)bz * g.b_bs * ES);
    const int64_t bytes_a = AK ? ((int64_t)(g.M - 1) * g.a_rs + g.K) * ES : ((int64_t)(g.K - 1) * g.a_ks + g.a_rows) * 2;
    const int64_t bytes_b = BKM ? ((int64_t)(g.N - 1) * g.b_ns + g.K) * ES : ((int64_t)(g.K - 1) * g.b_ks + g.b_rows) * 2;
    const OpSrc sa = make_src(A, bytes_a, g.a_rs, g.a_ks);
    const OpSrc sb = make_src(B, bytes_b, g.b_ns, g.b_ks);
    const int4v ra = raw_rsrc(A, bytes_a), rb = raw_rsrc(B, bytes_b);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

    int va[NCA], vb[NCB];
    glds_offsets<BM, AK, ES, NT>(va, sa, m0, g.a_rows, tid);
    glds_offsets<BN, BKM, ES, NT>(vb, sb, n0, g.b_rows, tid);
    // a k-tile further on: k-major operands advance along the row, r-major ones by 64 k-rows
    const int step_a = AK ? KT * ES : 64 * sa.ks * 2, step_b = BKM ? KT * ES : 64 * sb.ks * 2;
    const int nk = g.K / KT;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(
        (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(fsm) + wid * 1024);
    auto issue = [&](int t, int stage) {
        const unsigned dst = lds0 + stage * (STAGE * 2);
        const int oa = __builtin_amdgcn_readfirstlane(t * step_a), ob = __builtin_amdgcn_readfirstlane(t * step_b);
#pragma unroll
        for (int i = 0; i < NCA; ++i) glds16(ra, dst + i * NT * 16, va[i], oa);
#pragma unroll
        for (int i = 0; i < NCB; ++i) glds16(rb, dst + AEL * 2 + i * NT * 16, vb[i], ob);
    };
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < nk) issue(s, s);
    STAMP(g, 1);
    TRACE(g, 15);
    int stage = 0;
    for (int t = 0; t < nk; ++t) {
        // tile t has landed once at most the (NS - 2) younger tiles are still in flight
        TRACE(g, 16 + 4 * t);
        if (NS == 2 || t + NS - 2 >= nk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * G) : "memory");
        TRACE(g, 17 + 4 * t);
        if (!KABLATE(g, 128)) lds_barrier();  // every wave's part of tile t is visible; stage (t - 1) % NS has been read by everyone
        TRACE(g, 18 + 4 * t);
        if (t + NS - 1 < nk && !KABLATE(g, 8)) issue(t + NS - 1, stage == 0 ? NS - 1 : stage - 1);
        TRACE(g, 19 + 4 * t);
        const bf16* Ac = fsm + stage * STAGE;
        const bf16* Bc = Ac + AEL;
        if constexpr (F8) {  // one 128-deep step per k-tile on the block-scaled instruction
            int8v_t a8[TM], b8[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a8[i] = frag_e4m3<BM>(Ac, wm + i * 16, lane);
#pragma unroll
            for (int j = 0; j < TN; ++j) b8[j] = frag_e4m3<BN>(Bc, wn + j * 16, lane);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = mfma_e4m3_k128(b8[j], a8[i], acc[i][j]);
        } else
#pragma unroll
        for (int ks = 0; ks < 64; ks += 32) {
            bf16x8_t a[TM], b[TN];
            if (KABLATE(g, 256)) {  // no fragment reads at all: the matrix pipe on whatever the registers hold
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = __builtin_bit_cast(bf16x8_t, acc[i][0]);
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = __builtin_bit_cast(bf16x8_t, acc[0][j]);
            } else {
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = glds_frag<BM, AK>(KABLATE(g, 64) ? fsm : Ac, KABLATE(g, 64) ? 0 : wm + i * 16, ks, lane);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = glds_frag<BN, BKM>(KABLATE(g, 64) ? fsm : Bc, KABLATE(g, 64) ? 0 : wn + j * 16, ks, lane);
            }
            if (KABLATE(g, 32)) {
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i][0][0] += __builtin_bit_cast(float4_t, a[i])[0];
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[0][j][1] += __builtin_bit_cast(float4_t, b[j])[0];
                continue;
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
        }
        stage = stage + 1 == NS ? 0 : stage + 1;
    }
    TRACE(g, 14);
    lds_barrier();  // every wave is done with the last stage: LDS is free for the epilogue
    STAMP(g, 2);
    TRACE(g, 0);
}

// ---- k-loop with wave roles (round 4) ---------------------------------------------------------------------------
// tools/micro/kloop_pipe.hip, profiles/r04_experiments/: in gemm_kloop_glds every wave does everything in turn -- issue
// the LDS-DMA of a later tile (~26 cycles of the CU's address path per instruction, during which the wave issues nothing
// else), wait, barrier, read ALL fragments of a k-step, then its MFMAs -- and with one wave per SIMD the pieces add:
// 1 300-1 600 cycles per 128 x 128 x 64 k-tile against 512 of matrix work.  Here a 512-thread workgroup splits the work:
//   waves 0-3  compute: 2 x 2 over the tile, TWO fragment sets.  Iteration t = [k-step 0: MFMAs on set 0, the ds_reads of
//              (t, k-step 1) into set 1 between them] -> s_barrier -> [k-step 1: MFMAs on set 1, the reads of
//              (t + 1, k-step 0) into set 0 between them].  No vector-memory instruction, no vmcnt wait, and no
//              fragment read whose latency is exposed: after the barrier the matrix pipe runs on registers.
//   waves 4-7  load: every LDS-DMA instruction of the tile (1 KB pieces, loader l takes pieces l, l + 4, ...) and the
//              counted waits.  At the barrier of iteration t every compute wave holds both k-steps of tile t in
//              registers, so stage t % NS is free: tile t + NS goes there, i.e. NS - 1 whole tiles stay in flight.
// One s_barrier per k-tile joins the roles (the loaders arrive after their counted wait for tile t + 1).  The loaders
// exit after the last barrier; s_barrier counts surviving waves only, so the compute waves' epilogue barriers stand.
// Same k order and the same v_mfma_f32_16x16x32_bf16 chain per accumulator as every other tile: results are bit-identical.
// Measured per k-tile (128 x 128, one workgroup per CU, K = 3072): 1 345 cycles before, 1 108 with the fragment double
// buffer alone, 931 with the roles; the micro-benchmark's ideal (no DMA at all) is 740.
template <int BM, int BN, bool AK, bool BKM, int NS>
__device__ __forceinline__ void gemm_role_load(const GemmArgs& g, int tile_m, int tile_n, int bz, bf16* fsm, int l) {
    constexpr int PA = BM / 8, PB = BN / 8;      // 1 KB pieces of the A and of the B image of one stage
    constexpr int NA = PA / 4, NB = PB / 4;      // pieces per loader wave
    constexpr int P = NA + NB;
    constexpr int STAGE_B = (BM + BN) * 128;     // bytes per stage
    static_assert(PA % 4 == 0 && PB % 4 == 0, "whole pieces per loader");
    static_assert(BM <= 128 && BN <= 128, "at most four groups per k-step");
    const int lane = threadIdx.x & 63;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const bf16* A = reinterpret_cast<const bf16*>(g.A) + (int64_t)bz * g.a_bs;
    const bf16* B = reinterpret_cast<const bf16*>(g.B) + (int64_t)bz * g.b_bs;
    const int64_t bytes_a = AK ? ((int64_t)(g.M - 1) * g.a_rs + g.K) * 2 : ((int64_t)(g.K - 1) * g.a_ks + g.a_rows) * 2;
    const int64_t bytes_b = BKM ? ((int64_t)(g.N - 1) * g.b_ns + g.K) * 2 : ((int64_t)(g.K - 1) * g.b_ks + g.b_rows) * 2;
    const int4v ra = raw_rsrc(A, bytes_a), rb = raw_rsrc(B, bytes_b);
    const int a_rs = __builtin_amdgcn_readfirstlane((int)g.a_rs), a_ks = __builtin_amdgcn_readfirstlane((int)g.a_ks);
    const int b_ns = __builtin_amdgcn_readfirstlane((int)g.b_ns), b_ks = __builtin_amdgcn_readfirstlane((int)g.b_ks);
    // byte offset (inside one k-tile) of the chunk this lane moves in piece p of an operand image: LDS chunk c = 64 p + lane
    int va[NA], vb[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int c = (l + 4 * i) * 64 + lane;
        if (AK) {
            const int row = c >> 3, kc = (c & 7) ^ ((row >> 1) & 7);
            va[i] = (min(m0 + row, g.a_rows - 1) * a_rs + kc * 8) * 2;
        } else {
            const int kl = c / (BM / 8), rc = ((c % (BM / 8)) ^ glds_rx<BM>(kl)) * 8;
            va[i] = (kl * a_ks + min(m0 + rc, g.a_rows - 8)) * 2;
        }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int c = (l + 4 * i) * 64 + lane;
        if (BKM) {
            const int row = c >> 3, kc = (c & 7) ^ ((row >> 1) & 7);
            vb[i] = (min(n0 + row, g.b_rows - 1) * b_ns + kc * 8) * 2;
        } else {
            const int kl = c / (BN / 8), rc = ((c % (BN / 8)) ^ glds_rx<BN>(kl)) * 8;
            vb[i] = (kl * b_ks + min(n0 + rc, g.b_rows - 8)) * 2;
        }
    }
    const int step_a = AK ? 128 : 64 * a_ks * 2, step_b = BKM ? 128 : 64 * b_ks * 2;
    const int nk = g.K / 64;
    const unsigned lds0 =
        __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(fsm) + l * 1024);
    auto issue = [&](int t, int stage) {
        const unsigned dst = lds0 + stage * STAGE_B;
        const int oa = __builtin_amdgcn_readfirstlane(t * step_a), ob = __builtin_amdgcn_readfirstlane(t * step_b);
#pragma unroll
        for (int i = 0; i < NA; ++i) glds16(ra, dst + i * 4096, va[i], oa);
#pragma unroll
        for (int i = 0; i < NB; ++i) glds16(rb, dst + BM * 128 + i * 4096, vb[i], ob);
    };
    // tile 0 alone first: the compute waves start on it while tiles 1 .. NS - 1 are being issued
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int s = 1; s < NS; ++s)
        if (s < nk) issue(s, s);
    int stage = 0;
    for (int t = 0; t < nk; ++t) {
        // tile t + 1 has landed once at most the younger tiles t + 2 .. t + NS - 1 are in flight
        if (t + NS - 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * P) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t + NS < nk) issue(t + NS, stage);
        stage = stage + 1 == NS ? 0 : stage + 1;
    }
}

// The order hipcc is asked to emit for group Q of a k-step (it otherwise sinks every read below the MFMAs, i.e. back to
// "all reads, then all MFMAs" with the read latency exposed): one MFMA, the reads of one fragment, one MFMA, the reads
// of the next fragment, ..., the group's remaining MFMAs.  A k-major fragment is one ds_read_b128, an r-major one two
// ds_read_b64_tr_b16.  Fragment order as in gemm_role_compute: b0 a0 b1 .. b(TN-1) a1 .. a(TM-1).
template <int TM, int TN, bool AK, bool BKM, int Q> __device__ __forceinline__ void role_interleave() {
    constexpr int NR = TM + TN, RPG = (NR + TM - 1) / TM;
    constexpr int r0 = Q * RPG, n = (NR - r0) < RPG ? ((NR - r0) > 0 ? NR - r0 : 0) : RPG;
    static_assert(RPG <= 3, "at most three fragments per group");
#define XGGM_FRAG_INS(r) ((((r) == 0 || ((r) >= 2 && (r) <= TN)) ? BKM : AK) ? 1 : 2)
    if constexpr (n >= 1) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, XGGM_FRAG_INS(r0), 0);
    }
    if constexpr (n >= 2) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, XGGM_FRAG_INS(r0 + 1), 0);
    }
    if constexpr (n >= 3) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, XGGM_FRAG_INS(r0 + 2), 0);
    }
#undef XGGM_FRAG_INS
    if constexpr (TN - n > 0) __builtin_amdgcn_sched_group_barrier(0x008, TN - n, 0);
}

template <int BM, int BN, bool AK, bool BKM, int NS>
__device__ __forceinline__ void gemm_role_compute(const GemmArgs& g, bf16* fsm, float4_t (&acc)[BM / 32][BN / 32]) {
    constexpr int TM = BM / 32, TN = BN / 32;  // 16 x 16 blocks per wave (wave tile BM / 2 x BN / 2)
    constexpr int AEL = BM * 64, STAGE = (BM + BN) * 64;
    constexpr int NR = TM + TN, RPG = (NR + TM - 1) / TM;  // fragments per k-step; fragments requested per group of TN MFMAs
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wid >> 1) * (BM / 2), wn = (wid & 1) * (BN / 2);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};
    bf16x8_t a0[TM], b0[TN], a1[TM], b1[TN];
    // fragment r of a k-step in the order the MFMAs need them (i-major over the blocks): b0 a0 b1 .. b(TN-1) a1 .. a(TM-1)
    auto rd = [&](bf16x8_t (&a)[TM], bf16x8_t (&b)[TN], const bf16* st, int ks, int r) {
        if (r == 0) b[0] = glds_frag<BN, BKM>(st + AEL, wn, ks, lane);
        else if (r == 1) a[0] = glds_frag<BM, AK>(st, wm, ks, lane);
        else if (r <= TN) b[r - 1] = glds_frag<BN, BKM>(st + AEL, wn + (r - 1) * 16, ks, lane);
        else a[r - TN] = glds_frag<BM, AK>(st, wm + (r - TN) * 16, ks, lane);
    };
    const int nk = g.K / 64;
    __builtin_amdgcn_s_barrier();  // tile 0 has landed (the loaders waited for it)
    asm volatile("" ::: "memory");
    STAMP(g, 1);
#pragma unroll
    for (int r = 0; r < NR; ++r) rd(a0, b0, fsm, 0, r);
    int stage = 0;
    for (int t = 0; t < nk; ++t) {
        const bf16* sc = fsm + stage * STAGE;
        stage = stage + 1 == NS ? 0 : stage + 1;
        const bf16* sn = fsm + stage * STAGE;
        // k-step 0 on (a0, b0); the fragments of k-step 1 are requested between the MFMAs
#define XGGM_ROLE_GROUP(q, FA, FB, NA_, NB_, st, ks)                                                                   \
    if constexpr ((q) < TM) {                                                                                         \
        _Pragma("unroll") for (int r = (q) * RPG; r < ((q) + 1) * RPG && r < NR; ++r) rd(NA_, NB_, st, ks, r);         \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                                \
            acc[q][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FB[j], FA[q], acc[q][j], 0, 0, 0);                    \
        role_interleave<TM, TN, AK, BKM, (q)>();                                                                      \
    }
        XGGM_ROLE_GROUP(0, a0, b0, a1, b1, sc, 32)
        XGGM_ROLE_GROUP(1, a0, b0, a1, b1, sc, 32)
        XGGM_ROLE_GROUP(2, a0, b0, a1, b1, sc, 32)
        XGGM_ROLE_GROUP(3, a0, b0, a1, b1, sc, 32)
        // every fragment of tile t is in registers: its stage may be refilled; tile t + 1 is visible after the barrier
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // k-step 1 on (a1, b1); the fragments of (t + 1, k-step 0) under them (after the last tile: stale bytes, unused)
        XGGM_ROLE_GROUP(0, a1, b1, a0, b0, sn, 0)
        XGGM_ROLE_GROUP(1, a1, b1, a0, b0, sn, 0)
        XGGM_ROLE_GROUP(2, a1, b1, a0, b0, sn, 0)
        XGGM_ROLE_GROUP(3, a1, b1, a0, b0, sn, 0)
#undef XGGM_ROLE_GROUP
    }
    lds_barrier();  // the compute waves are done with LDS (the loaders have exited or are about to): free for the epilogue
    STAMP(g, 2);
}

// K a whole number of k-tiles (no partial chunk to zero on the way, no tile to skip), r-major rows readable in
// whole 16-byte chunks
template <bool F8> __host__ __device__ __forceinline__ bool glds_ok(const GemmArgs& g) {
    constexpr int KT = F8 ? 128 : 64;
    return g.use_glds && g.K % KT == 0 && !g.a_tail && !g.b_tail && (g.a_mode == 1 || g.a_rows >= 8) &&
           (g.b_mode == 1 || g.b_rows >= 8);
}

// epilogue of one tile (shared by every operand-layout variant of the k-loop: ONE copy of its code per kernel)
template <int BM, int BN, int W = 4>
__device__ __forceinline__ void gemm_finish(const GemmArgs& g, int tile_m, int tile_n, int bz, bf16* fsm,
                                            const float4_t (&acc)[BM / (8 * W)][BN / 32]) {
    constexpr int TM = BM / (8 * W), TN = BN / 32;
    const int wid = threadIdx.x >> 6;
    const int wm = (wid >> 1) * (BM / (W / 2)), wn = (wid & 1) * (BN / 2);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    if (ABLATE(g, 2)) {
        float keep = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) keep += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (keep == 123.456f) reinterpret_cast<float*>(g.C)[threadIdx.x] = keep;
        return;
    }
    const int kind = epilogue_kind<BM, BN, W>(g, bz);
    if (kind == 0) epilogue_direct<BM, BN, TM, TN, W, 0>(g, acc, m0, n0, bz, reinterpret_cast<float*>(fsm), wm, wn);
    else if (kind == 1) epilogue_direct<BM, BN, TM, TN, W, 1>(g, acc, m0, n0, bz, reinterpret_cast<float*>(fsm), wm, wn);
    else epilogue_staged<BM, BN, TM, TN, W>(g, acc, m0, n0, bz, reinterpret_cast<float*>(fsm), wm, wn);
    STAMP(g, 4);
    TRACE(g, 12);
#ifdef XGGM_STAMP
    if (g.ablate & 0x10000) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // trace: when the tile's stores have been acknowledged
#endif
    TRACE(g, 13);
    TRACE_FLUSH(g);
}

template <int BM, int BN, bool AK, bool BKM, int D>
__device__ __forceinline__ void gemm_tile(const GemmArgs& g, int tile_m, int tile_n, int bz, bf16* fsm) {
    float4_t acc[BM / 32][BN / 32];
    gemm_kloop<BM, BN, AK, BKM, D>(g, tile_m, tile_n, bz, fsm, acc);
    gemm_finish<BM, BN>(g, tile_m, tile_n, bz, fsm, acc);
}


// XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (id % 8), each with its own
// 4 MiB L2.  Give every XCD a CONTIGUOUS run of positions (xcd_remap) and read a position as a tile
// of one rectangle of the grid (tile_of_position), so tiles sharing operand panels run on one L2
// instead of pulling every panel into all eight.  Placement changes speed and traffic only.
__device__ __forceinline__ int xcd_remap(int L, int nb) {
    const int q = nb >> 3, r = nb & 7, xcd = L & 7, idx = L >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// (One level further down -- the CU's own L1 -- was measured and buys nothing: HW_ID stamps (tools/gemm_wg_map.py) show
// that an XCD's q-th and (q + 32)-th workgroups start on the same CU within ~15 cycles of each other, so renumbering
// the blocks puts the co-resident workgroups of a CU on neighbouring tiles of one tile row, asking for the same A
// panel at the same time.  k-loop cycles of the forward pairs did not move (16.80 k -> 16.78 k per 128 x 64 x 768
// tile), and the backward groups lost 15-20 % because a grid-wide renumbering breaks the longest-K-first order.)
// The same idea in two dimensions.  An XCD that owns a run of whole tile rows pulls every B panel into its L2
// (8 copies of B across the chip); an xr x xc arrangement of the XCDs over the tile grid fetches A xc times and
// B xr times instead.  Position p of the XCD-contiguous order (xcd_remap) is read rectangle-major: bands of rh
// tile rows, inside a band column groups of rw tiles, inside a group row-major -- a run of ~n/8 positions is
// (about) one rh x rw rectangle; ragged edges only make the last band / group smaller.  Everything here is
// wave-uniform (scalar unit); band and group are found by comparison (at most 8 each), leaving one division.
__device__ __forceinline__ void tile_of_position(int p, int gx, int gy, int M, int N, int& tile_m, int& tile_n) {
    // minimise xc * |A| + xr * |B| ~ xc * M + xr * N over xr * xc = 8
    int xr = 1;
    long best = 8l * M + N;
#pragma unroll
    for (int r = 2; r <= 8; r *= 2) {
        const long c = (long)(8 / r) * M + (long)r * N;
        if (c < best) { best = c; xr = r; }
    }
    const int xc = 8 / xr;
    const int rh = (gy + xr - 1) / xr, rw = (gx + xc - 1) / xc;  // divisions by 1, 2, 4, 8: shifts
    int band = 0, q = p;
#pragma unroll
    for (int b = 1; b < 8; ++b)
        if (b < xr && p >= b * rh * gx) { band = b; q = p - b * rh * gx; }
    const int hb = min(rh, gy - band * rh);
    int cg = 0, r = q;
#pragma unroll
    for (int c = 1; c < 8; ++c)
        if (c < xc && q >= c * rw * hb) { cg = c; r = q - c * rw * hb; }
    const int wb = min(rw, gx - cg * rw);
    const int dr = r / wb;
    tile_m = band * rh + dr;
    tile_n = cg * rw + (r - dr * wb);
}

// The same map for the grouped kernels with every per-problem constant computed on the HOST (set_tile_map): in-kernel
// cycle stamps (tools/micro/kloop_pipe.hip, profiles/r04_experiments/kloop_role_prologue.txt) put the first LDS-DMA
// instruction of a workgroup 3 000 cycles after kernel entry, 2 300 of them between "kernel arguments visible" and "tile
// known": the rectangle shape, rh, rw and r / wb are four or five integer divisions, each a dependent chain of ~30
// vector and scalar instructions (there is no scalar divide).  With the constants in the arguments the map is a dozen
// scalar compares and ONE multiply-high per quotient (n / d = umulhi(n, ceil(2^32 / d)), exact while n * d < 2^32).
__device__ __forceinline__ int div_magic(int n, int d, unsigned m) { return d == 1 ? n : (int)__umulhi((unsigned)n, m); }

__device__ __forceinline__ void tile_from_map(const GemmArgs& g, int local, int nblk, int& tile_m, int& tile_n, int& bz) {
    const GemmArgs::TileMap& tm = g.tm;
    const int L = g.xcd_swizzle ? xcd_remap(local, nblk) : local;
    bz = g.batch == 1 ? 0 : div_magic(L, tm.gxy, tm.m_gxy);
    const int p = L - bz * tm.gxy;
    if (!g.xcd_swizzle) {  // test hook (xggm_gemm_set_tile(| 0x100)): row-major tiles
        tile_m = p / tm.gx;
        tile_n = p - tile_m * tm.gx;
        return;
    }
    const int band_sz = tm.rh * tm.gx;
    int band = 0;
#pragma unroll
    for (int b = 1; b < 8; ++b)
        if (b < tm.xr && p >= b * band_sz) band = b;
    const int q = p - band * band_sz;
    const int hb = min(tm.rh, tm.gy - band * tm.rh);
    const int grp = tm.rw * hb;
    int cg = 0;
#pragma unroll
    for (int c = 1; c < 8; ++c)
        if (c < tm.xc && q >= c * grp) cg = c;
    const int r = q - cg * grp;
    const bool last = cg == tm.ncg - 1;
    const int wb = last ? tm.wb_last : tm.rw;
    const int dr = div_magic(r, wb, last ? tm.m_wbl : tm.m_rw);
    tile_m = band * tm.rh + dr;
    tile_n = cg * tm.rw + (r - dr * wb);
}

inline unsigned magic_of(int d) { return d <= 1 ? 0u : (unsigned)(((1ull << 32) + (unsigned)d - 1) / (unsigned)d); }
inline void set_tile_map(GemmArgs& g, int bm, int bn) {
    GemmArgs::TileMap& tm = g.tm;
    tm.gx = ceil_div(g.N, bn);
    tm.gy = ceil_div(g.M, bm);
    tm.gxy = tm.gx * tm.gy;
    // minimise xc * |A| + xr * |B| ~ xc * M + xr * N over xr * xc = 8
    int xr = 1;
    long best = 8l * g.M + g.N;
    for (int r = 2; r <= 8; r *= 2) {
        const long c = (long)(8 / r) * g.M + (long)r * g.N;
        if (c < best) { best = c; xr = r; }
    }
    tm.xr = xr;
    tm.xc = 8 / xr;
    tm.rh = ceil_div(tm.gy, tm.xr);
    tm.rw = ceil_div(tm.gx, tm.xc);
    tm.ncg = ceil_div(tm.gx, tm.rw);
    tm.wb_last = tm.gx - (tm.ncg - 1) * tm.rw;
    tm.m_rw = magic_of(tm.rw);
    tm.m_wbl = magic_of(tm.wb_last);
    tm.m_gxy = magic_of(tm.gxy);
}

// waves per SIMD the register allocation must leave room for: a second (third, fourth) resident
// workgroup runs its k-loop under this one's prologue and epilogue
template <int BM, int BN> constexpr int min_waves() { return BM * BN >= 128 * 128 ? 2 : 3; }

template <int BM, int BN, bool AK, bool BKM, int D>
__global__ __launch_bounds__(NT, (min_waves<BM, BN>())) void gemm_fast_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) bf16 fsm[];
    int tile_m = blockIdx.y, tile_n = blockIdx.x;
    if (g.xcd_swizzle)
        tile_of_position(xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y), gridDim.x, gridDim.y, g.M,
                         g.N, tile_m, tile_n);
    gemm_tile<BM, BN, AK, BKM, D>(g, tile_m, tile_n, blockIdx.z, fsm);
}

// ---- fp8 forward GEMM (BASELINE config C5: e4m3 operands for the QKV / FFN products, everything else bf16).
// C = epilogue(sa * sb * sum_k A8(m,k) B8(n,k)): A8, B8 OCP e4m3 bytes, k-contiguous, quantised with per-tensor scales
// whose reciprocals sa, sb are device scalars; accumulation in fp32 on the matrix cores, epilogue and output as
// in the bf16 kernel.  Same tile pipeline with half the operand bytes per k (see gemm_kloop).
template <int BM, int BN, int D>
__global__ __launch_bounds__(NT, (min_waves<BM, BN>())) void gemm_fp8_kernel(GemmArgs g, const float* sa, const float* sb) {
    extern __shared__ __attribute__((aligned(16))) bf16 fsm[];
    const float s = (sa ? *sa : 1.f) * (sb ? *sb : 1.f);
    int tile_m = blockIdx.y, tile_n = blockIdx.x;
    if (g.xcd_swizzle)
        tile_of_position(xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y), gridDim.x, gridDim.y, g.M,
                         g.N, tile_m, tile_n);
    float4_t acc[BM / 32][BN / 32];
    gemm_kloop<BM, BN, true, true, D, true>(g, tile_m, tile_n, 0, fsm, acc);
    g.alpha *= s;
    gemm_finish<BM, BN>(g, tile_m, tile_n, 0, fsm, acc);
}

template <int BM, int BN, int D> int launch_fp8_tile(const GemmArgs& g, const float* sa, const float* sb, hipStream_t stream) {
    constexpr size_t lds = 2 * sizeof(bf16) * (OpLds<BM, true>::ELEMS + OpLds<BN, true>::ELEMS);
    static bool attr_set = false;
    if (lds > 48 * 1024 && !attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_fp8_kernel<BM, BN, D>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_fp8_kernel<BM, BN, D>), dim3(ceil_div(g.N, BN), ceil_div(g.M, BM), 1), dim3(NT), lds, stream, g,
                       sa, sb);
    return xggm_check_launch("xggm_gemm_fp8e4m3");
}

// ---- grouped launch: up to MAX_GROUP independent GEMMs (forward of both modalities, dgrad + wgrad of one
// layer, ...) share ONE grid, so skinny problems that cannot fill 256 CUs alone fill them together
// and the per-launch latency is paid once.  Every workgroup looks up its problem from the tile
// prefix sums and runs the layout-specialised tile routine.
// (6 since round 4: launch cost does not depend on the size of the argument block, tools/micro/launch_gap.hip, and the
// cross-attention backward's fifth problem and the graph blocks' six read-out problems were launches of their own)
constexpr int MAX_GROUP = 6;
struct GroupArgs {
    GemmArgs p[MAX_GROUP];
    int tile_start[MAX_GROUP + 1];
    int nprob;
    int stages;  // LDS stages of the LDS-DMA k-loop (2 or 3), chosen per launch: see launch_grouped_tile
};

template <int BM, int BN, int W = 4>
__global__ __launch_bounds__(64 * W, (W == 8 ? 2 : min_waves<BM, BN>())) void gemm_grouped_kernel(GroupArgs ga) {
    extern __shared__ __attribute__((aligned(16))) bf16 fsm[];
    // Which problem, and which tile of it.  Inside each problem the tiles get the XCD-aware order:
    // blocks are dealt to XCDs by id % 8, so the problem's blocks of one residue class take one
    // contiguous run of positions = one rectangle of its tile grid (tile_of_position), while every
    // XCD still receives 1/8 of EVERY problem (a remap over the whole grid would hand whole
    // problems, with different k-loop lengths, to different XCDs).  L2-miss reads per launch
    // (rocprofv3 FETCH_SIZE): natural order 76.6 MB, row-major runs 90.7 MB, rectangles 64.9 MB
    // for the 128x64 kernel (43.8 / 45.5 / 33.3 MB for 64x64); step time equal within noise.
    const int b = blockIdx.x;
    int i = 0, t0 = 0, t1 = ga.tile_start[1];
#pragma unroll
    for (int k = 1; k < MAX_GROUP; ++k)
        if (k < ga.nprob && b >= ga.tile_start[k]) { i = k; t0 = ga.tile_start[k]; t1 = ga.tile_start[k + 1]; }
    const GemmArgs g = ga.p[i];
    int tile_m, tile_n, bz;
    tile_from_map(g, b - t0, t1 - t0, tile_m, tile_n, bz);
    constexpr int DK = (BM * BN <= 64 * 64) ? 4 : 2;  // prefetch depth when an operand is k-major
    float4_t acc[BM / (8 * W)][BN / 32];
    if (glds_ok<false>(g)) {
#define XGGM_GLDS_FORMS(NS)                                                                                          \
    if (g.a_mode == 1) {                                                                                             \
        if (g.b_mode == 1) gemm_kloop_glds<BM, BN, true, true, false, W, NS>(g, tile_m, tile_n, bz, fsm, acc);       \
        else gemm_kloop_glds<BM, BN, true, false, false, W, NS>(g, tile_m, tile_n, bz, fsm, acc);                    \
    } else {                                                                                                         \
        if (g.b_mode == 1) gemm_kloop_glds<BM, BN, false, true, false, W, NS>(g, tile_m, tile_n, bz, fsm, acc);      \
        else gemm_kloop_glds<BM, BN, false, false, false, W, NS>(g, tile_m, tile_n, bz, fsm, acc);                   \
    }
        if (ga.stages == 2) { XGGM_GLDS_FORMS(2) }
        else if (ga.stages == 3) { XGGM_GLDS_FORMS(3) }
        else { XGGM_GLDS_FORMS(4) }
#undef XGGM_GLDS_FORMS
    } else if (g.a_mode == 1) {
        if (g.b_mode == 1) gemm_kloop<BM, BN, true, true, DK, false, W>(g, tile_m, tile_n, bz, fsm, acc);
        else gemm_kloop<BM, BN, true, false, DK, false, W>(g, tile_m, tile_n, bz, fsm, acc);
    } else {
        if (g.b_mode == 1) gemm_kloop<BM, BN, false, true, DK, false, W>(g, tile_m, tile_n, bz, fsm, acc);
        else gemm_kloop<BM, BN, false, false, 2, false, W>(g, tile_m, tile_n, bz, fsm, acc);
    }
    gemm_finish<BM, BN, W>(g, tile_m, tile_n, bz, fsm, acc);
}

template <int BM, int BN, int W = 4> int launch_grouped_tile(GroupArgs& ga, hipStream_t stream) {
    int total = 0;
    for (int i = 0; i < ga.nprob; ++i) {
        ga.tile_start[i] = total;
        set_tile_map(ga.p[i], BM, BN);
        total += ga.p[i].tm.gxy * ga.p[i].batch;
    }
    ga.tile_start[ga.nprob] = total;
    // Stages of the LDS-DMA k-loop (NS - 1 k-tiles in flight per workgroup).  On L2-hot operands (a micro-benchmark
    // that relaunches the same problem) two stages win wherever a third costs a resident workgroup (72 KB per 128 x 64
    // workgroup = two per CU; FFN forward pair, 672 tiles: 18.8 us with two stages, 21.4 with three).  Inside the
    // training step the weights arrive from HBM, every k-step of every tile waits for lines nobody has touched yet,
    // and depth beats occupancy: rocprofv3 averages over the step, registers | 2 | 3 | 4 stages --
    // 128 x 64: 36.5 | 37.2 | 35.2 | 46.6 us (four stages = one workgroup per CU), 64 x 64: 15.3 | 17.9 | 14.6 | 15.4,
    // 128 x 128 on 8 waves: 18.1 | 20.6 | 17.4 | 17.5.  Three everywhere except the 4-wave 128 x 128 tile (3 x 64 KB
    // would leave one workgroup per CU where two fit).
    // four-wave tiles whose three stages would leave ONE workgroup per CU take two (128 x 128: 2 x 64 KB; 192 x 128: 2 x 80 KB)
    ga.stages = g_glds_stages ? g_glds_stages : (W == 4 && 2 * 3 * sizeof(bf16) * (BM + BN) * 64 > 160 * 1024) ? 2 : 3;
    while (ga.stages > 2 && ga.stages * sizeof(bf16) * (BM + BN) * 64 > 160 * 1024) --ga.stages;
    // LDS of the largest operand images this group actually uses (r-major images carry padding)
    size_t lds = 0;
    for (int i = 0; i < ga.nprob; ++i) {
        const size_t a = ga.p[i].a_mode == 1 ? OpLds<BM, true>::ELEMS : OpLds<BM, false>::ELEMS;
        const size_t b = ga.p[i].b_mode == 1 ? OpLds<BN, true>::ELEMS : OpLds<BN, false>::ELEMS;
        lds = std::max(lds, glds_ok<false>(ga.p[i]) ? ga.stages * sizeof(bf16) * (BM + BN) * 64 : 2 * sizeof(bf16) * (a + b));
    }
    constexpr size_t lds_max = std::max<size_t>(2 * sizeof(bf16) * (OpLds<BM, false>::ELEMS + OpLds<BN, false>::ELEMS),
                                                std::min<size_t>(160 * 1024, 4 * sizeof(bf16) * (BM + BN) * 64));
    static bool attr_set = false;
    if (lds_max > 48 * 1024 && !attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_grouped_kernel<BM, BN, W>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_grouped_kernel<BM, BN, W>), dim3(total), dim3(64 * W), lds, stream, ga);
    return xggm_check_launch("xggm_gemm_grouped");
}

// The grouped launch on the role k-loop: 512 threads, waves 4-7 load (gemm_role_load), waves 0-3 compute and run the
// epilogue of the four-wave tiles.  Every problem of the group must pass glds_ok (whole k-tiles).
constexpr int ROLE_NS = 3;
template <int BM, int BN> constexpr int role_min_waves() { return BM * BN >= 128 * 128 ? 2 : 4; }  // workgroups per CU x 2

template <int BM, int BN>
__global__ __launch_bounds__(512, (role_min_waves<BM, BN>())) void gemm_grouped_role_kernel(GroupArgs ga) {
    extern __shared__ __attribute__((aligned(16))) bf16 fsm[];
    const int b = blockIdx.x;
    int i = 0, t0 = 0, t1 = ga.tile_start[1];
#pragma unroll
    for (int k = 1; k < MAX_GROUP; ++k)
        if (k < ga.nprob && b >= ga.tile_start[k]) { i = k; t0 = ga.tile_start[k]; t1 = ga.tile_start[k + 1]; }
    const GemmArgs g = ga.p[i];
    int tile_m, tile_n, bz;
    tile_from_map(g, b - t0, t1 - t0, tile_m, tile_n, bz);
    STAMP(g, 0);
    STAMP_HW(g);
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (wid >= 4) {
        if (g.a_mode == 1) {
            if (g.b_mode == 1) gemm_role_load<BM, BN, true, true, ROLE_NS>(g, tile_m, tile_n, bz, fsm, wid - 4);
            else gemm_role_load<BM, BN, true, false, ROLE_NS>(g, tile_m, tile_n, bz, fsm, wid - 4);
        } else {
            if (g.b_mode == 1) gemm_role_load<BM, BN, false, true, ROLE_NS>(g, tile_m, tile_n, bz, fsm, wid - 4);
            else gemm_role_load<BM, BN, false, false, ROLE_NS>(g, tile_m, tile_n, bz, fsm, wid - 4);
        }
        return;
    }
    float4_t acc[BM / 32][BN / 32];
    if (g.a_mode == 1) {
        if (g.b_mode == 1) gemm_role_compute<BM, BN, true, true, ROLE_NS>(g, fsm, acc);
        else gemm_role_compute<BM, BN, true, false, ROLE_NS>(g, fsm, acc);
    } else {
        if (g.b_mode == 1) gemm_role_compute<BM, BN, false, true, ROLE_NS>(g, fsm, acc);
        else gemm_role_compute<BM, BN, false, false, ROLE_NS>(g, fsm, acc);
    }
    gemm_finish<BM, BN, 4>(g, tile_m, tile_n, bz, fsm, acc);
}

template <int BM, int BN> int launch_grouped_role(GroupArgs& ga, hipStream_t stream) {
    int total = 0;
    for (int i = 0; i < ga.nprob; ++i) {
        ga.tile_start[i] = total;
        set_tile_map(ga.p[i], BM, BN);
        total += ga.p[i].tm.gxy * ga.p[i].batch;
    }
    ga.tile_start[ga.nprob] = total;
    ga.stages = ROLE_NS;
    // the stages, or the staged epilogue's (BM / 2) x (BN + 4) floats of the same memory
    constexpr size_t lds = std::max(ROLE_NS * sizeof(bf16) * (BM + BN) * 64, sizeof(float) * (BM / 2) * (BN + 4));
    static bool attr_set = false;
    if (lds > 48 * 1024 && !attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_grouped_role_kernel<BM, BN>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_grouped_role_kernel<BM, BN>), dim3(total), dim3(512), lds, stream, ga);
    return xggm_check_launch("xggm_gemm_grouped(role)");
}

// The grouped launch with e4m3 operands (forward products of both modality streams of one layer): every problem
// is k-major on both sides, so there is ONE k-loop instantiation; the per-problem dequantisation factor
// scale_a * scale_b is folded into alpha before the (shared) epilogue.
template <int BM, int BN, int W = 4>
__global__ __launch_bounds__(64 * W, (W == 8 ? 2 : min_waves<BM, BN>())) void gemm_grouped_fp8_kernel(GroupArgs ga) {
    extern __shared__ __attribute__((aligned(16))) bf16 fsm[];
    const int b = blockIdx.x;
    int i = 0, t0 = 0, t1 = ga.tile_start[1];
#pragma unroll
    for (int k = 1; k < MAX_GROUP; ++k)
        if (k < ga.nprob && b >= ga.tile_start[k]) { i = k; t0 = ga.tile_start[k]; t1 = ga.tile_start[k + 1]; }
    GemmArgs g = ga.p[i];
    int tile_m, tile_n, bz;
    tile_from_map(g, b - t0, t1 - t0, tile_m, tile_n, bz);
    constexpr int DK = (BM * BN <= 64 * 64) ? 4 : 2;
    float4_t acc[BM / (8 * W)][BN / 32];
    // the dequantisation factor is fetched before the k-loop, not in front of the epilogue
    const float deq = (g.scale_a ? *g.scale_a : 1.f) * (g.scale_b ? *g.scale_b : 1.f);
    if (glds_ok<true>(g)) {  // K a whole number of 128-element k-tiles: the LDS-DMA k-loop (same bytes per k-tile as bf16)
        if (ga.stages == 3) gemm_kloop_glds<BM, BN, true, true, true, W, 3>(g, tile_m, tile_n, bz, fsm, acc);
        else gemm_kloop_glds<BM, BN, true, true, true, W, 2>(g, tile_m, tile_n, bz, fsm, acc);
    } else {
        gemm_kloop<BM, BN, true, true, DK, true, W>(g, tile_m, tile_n, bz, fsm, acc);
    }
    g.alpha *= deq;
    gemm_finish<BM, BN, W>(g, tile_m, tile_n, bz, fsm, acc);
}

template <int BM, int BN, int W = 4> int launch_grouped_fp8_tile(GroupArgs& ga, hipStream_t stream) {
    int total = 0;
    for (int i = 0; i < ga.nprob; ++i) {
        ga.tile_start[i] = total;
        set_tile_map(ga.p[i], BM, BN);
        total += ga.p[i].tm.gxy * ga.p[i].batch;
    }
    ga.tile_start[ga.nprob] = total;
    // k-major images of 128 bytes per row (128 e4m3 values), double buffered; the staged epilogue needs
    // (BM / 2) x (BN + 4) floats of the same memory
    ga.stages = g_glds_stages ? std::min(g_glds_stages, 3) : (BM * BN == 128 * 128 && W == 4) ? 2 : 3;
    bool glds = true;
    for (int i = 0; i < ga.nprob; ++i) glds = glds && glds_ok<true>(ga.p[i]);
    const size_t lds = std::max((glds ? ga.stages : 2) * sizeof(bf16) * (OpLds<BM, true>::ELEMS + OpLds<BN, true>::ELEMS),
                                sizeof(float) * (BM / 2) * (BN + 4));
    constexpr size_t lds_max = std::max(3 * sizeof(bf16) * (OpLds<BM, true>::ELEMS + OpLds<BN, true>::ELEMS),
                                        sizeof(float) * (BM / 2) * (BN + 4));
    static bool attr_set = false;
    if (lds_max > 48 * 1024 && !attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_grouped_fp8_kernel<BM, BN, W>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_grouped_fp8_kernel<BM, BN, W>), dim3(total), dim3(64 * W), lds, stream, ga);
    return xggm_check_launch("xggm_gemm_grouped_fp8e4m3");
}

template <int BM, int BN, int D> int launch_fast_tile(const GemmArgs& g, int batch, hipStream_t stream) {
    dim3 grid(ceil_div(g.N, BN), ceil_div(g.M, BM), batch);
    const bool ak = g.a_mode == 1, bk = g.b_mode == 1;
#define XGGM_FAST(AKv, BKv)                                                                                           \
    do {                                                                                                              \
        constexpr size_t lds = 2 * sizeof(bf16) * (OpLds<BM, AKv>::ELEMS + OpLds<BN, BKv>::ELEMS);                   \
        static bool attr_set = false;                                                                                 \
        if (lds > 48 * 1024 && !attr_set) {                                                                           \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_fast_kernel<BM, BN, AKv, BKv, D>),          \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                          \
            attr_set = true;                                                                                          \
        }                                                                                                             \
        hipLaunchKernelGGL((gemm_fast_kernel<BM, BN, AKv, BKv, D>), grid, dim3(NT), lds, stream, g);                  \
    } while (0)
    if (ak && bk) XGGM_FAST(true, true);
    else if (ak && !bk) XGGM_FAST(true, false);
    else if (!ak && bk) XGGM_FAST(false, true);
    else XGGM_FAST(false, false);
#undef XGGM_FAST
    return xggm_check_launch("xggm_gemm(fast)");
}

// tile / depth choice.  g_tile_override (xggm_gemm_set_tile) pins one variant for A/B tests:
// 1 = 64x64 D2, 2 = 64x64 D4, 3 = 128x64 D2, 5 = 128x128 D2, 6-8 = depth 1; 0 = heuristic.
int g_tile_override = 0;
inline int launch_fast(const GemmArgs& g, int batch, hipStream_t stream) {
    auto tiles = [&](int bm, int bn) { return (int64_t)ceil_div(g.M, bm) * ceil_div(g.N, bn) * batch; };
    int v = g_tile_override;
    if (v == 0) {
        // measured on MI355X at the step's shapes (tools/bench_gemm.py): 64x64 tiles win everywhere
        // (these GEMMs need many workgroups more than big tiles); depth 4 helps when an operand is
        // k-major, depth 2 is better for the all-transposed wgrad form
        (void)tiles;
        v = (g.a_mode == 2 && g.b_mode == 2) ? 1 : 2;
    }
    switch (v) {
        case 6: return launch_fast_tile<64, 64, 1>(g, batch, stream);
        case 7: return launch_fast_tile<128, 64, 1>(g, batch, stream);
        case 8: return launch_fast_tile<128, 128, 1>(g, batch, stream);
        case 1: return launch_fast_tile<64, 64, 2>(g, batch, stream);
        case 2: return launch_fast_tile<64, 64, 4>(g, batch, stream);
        case 3: return launch_fast_tile<128, 64, 2>(g, batch, stream);
        default: return launch_fast_tile<128, 128, 2>(g, batch, stream);
    }
}

// How an operand can be loaded by the tuned kernels: 1 = 16-byte chunks along k, 2 = along rows, 0 = neither.
// A dimension that is not a multiple of 8 is accepted when the operand's own leading stride leaves room to
// read the chunk that straddles the edge (caller-padded buffers: the 2274-answer and 630-edge heads):
// k-contiguous with K % 8 != 0 needs rs >= roundup(K, 8) (the excess is zeroed in fast_store: `tail`),
// row-contiguous with R % 8 != 0 needs ks >= roundup(R, 8) (the excess rows only feed outputs that are
// never stored: `rows` = the padded count).
template <typename T> int pick_mode(const void* base, int64_t rs, int64_t ks, int64_t bs, int R, int K, int* tail, int* rows) {
    constexpr int VEC = Tile<T>::VEC;
    *tail = 0;
    *rows = R;
    const bool base_ok = (reinterpret_cast<uintptr_t>(base) % 16 == 0) && (bs % VEC == 0);
    if (ks == 1 && base_ok && rs % VEC == 0) {
        if (K % VEC == 0) return 1;
        if (sizeof(T) == 2 && K % 2 == 0 && rs >= (K + VEC - 1) / VEC * VEC) {
            *tail = 1;
            return 1;
        }
    }
    if (rs == 1 && base_ok && ks % VEC == 0) {
        if (R % VEC == 0) return 2;
        if (sizeof(T) == 2 && ks >= (R + VEC - 1) / VEC * VEC) {
            *rows = (R + VEC - 1) / VEC * VEC;
            return 2;
        }
    }
    return 0;
}

template <typename T> int launch(GemmArgs g, int batch, hipStream_t stream) {
    XGGM_REQUIRE(g.M > 0 && g.N > 0 && g.K > 0 && batch > 0, "xggm_gemm: empty problem M=%d N=%d K=%d batch=%d", g.M, g.N,
                 g.K, batch);
    XGGM_REQUIRE(g.A && g.B && g.C, "xggm_gemm: null operand");
    XGGM_REQUIRE(g.act >= 0 && g.act <= XGGM_ACT_GELU_GRAD, "xggm_gemm: bad activation %d", g.act);
    XGGM_REQUIRE(g.act != XGGM_ACT_GELU_GRAD || g.aux, "xggm_gemm: GELU_GRAD needs aux");
    XGGM_REQUIRE(g.ldc >= g.N, "xggm_gemm: ldc %lld < N %d", (long long)g.ldc, g.N);
    XGGM_REQUIRE(!g.sqsum || g.c_f32, "xggm_gemm: sqsum is defined for fp32 outputs");
    g.a_mode = pick_mode<T>(g.A, g.a_rs, g.a_ks, g.a_bs, g.M, g.K, &g.a_tail, &g.a_rows);
    g.b_mode = pick_mode<T>(g.B, g.b_ns, g.b_ks, g.b_bs, g.N, g.K, &g.b_tail, &g.b_rows);
    dim3 grid(ceil_div(g.N, BN), ceil_div(g.M, BM), batch);
    XGGM_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "xggm_gemm: grid too large");
    if constexpr (sizeof(T) == 2) {
        if (g.a_mode != 0 && g.b_mode != 0 && !g_force_generic) return launch_fast(g, batch, stream);
    }
    XGGM_REQUIRE(!g.sqsum, "xggm_gemm: sqsum needs the tuned bf16 kernels (8-aligned operands); this problem runs on the "
                           "generic kernel");
    // the generic kernel's vector loads assume exact multiples: padded-edge operands load element-wise there
    if (g.a_tail || g.a_rows != g.M) g.a_mode = 0;
    if (g.b_tail || g.b_rows != g.N) g.b_mode = 0;
    hipLaunchKernelGGL(gemm_kernel<T>, grid, dim3(NT), 0, stream, g);
    return xggm_check_launch("xggm_gemm");
}

}  // namespace

#define XGGM_GEMM_IMPL(NAME, T)                                                                                        \
    extern "C" int NAME(const void* A, const void* B, void* C, int M, int N, int K, int64_t a_rs, int64_t a_ks,       \
                        int64_t b_ns, int64_t b_ks, int64_t ldc, int batch, int64_t a_bs, int64_t b_bs, int64_t c_bs, \
                        const float* bias, const void* residual, void* preact, const void* aux, float* colsum,       \
                        int act, int c_f32, int accumulate, float alpha, hipStream_t stream) {                        \
        GemmArgs g;                                                                                                    \
        g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K;                                                          \
        g.a_rs = a_rs; g.a_ks = a_ks; g.b_ns = b_ns; g.b_ks = b_ks; g.ldc = ldc;                                       \
        g.a_bs = a_bs; g.b_bs = b_bs; g.c_bs = c_bs;                                                                   \
        g.bias = bias; g.residual = residual; g.preact = preact; g.aux = aux; g.colsum = colsum; g.sqsum = nullptr;   \
        g.c8 = nullptr; g.c8_qscale = nullptr; g.c8_amax = nullptr; g.scale_a = g.scale_b = nullptr; g.amax_slots = 1;  \
        g.act = act; g.c_f32 = c_f32; g.accumulate = accumulate; g.alpha = alpha;                                     \
        g.a_mode = g.b_mode = 0; g.xcd_swizzle = g_xcd_swizzle; g.batch = batch; g.use_glds = g_glds; SET_STAMP(g); \
        return launch<T>(g, batch, stream);                                                                            \
    }

XGGM_GEMM_IMPL(xggm_gemm_f32, float)
XGGM_GEMM_IMPL(xggm_gemm_bf16, bf16)

extern "C" int xggm_gemm_fp8e4m3(const void* A, const void* B, void* C, int M, int N, int K, int64_t a_rs, int64_t b_ns,
                                 int64_t ldc, const float* scale_a, const float* scale_b, const float* bias,
                                 const void* residual, void* preact, int act, int c_f32, hipStream_t stream) {
    XGGM_REQUIRE(M > 0 && N > 0 && K > 0, "xggm_gemm_fp8e4m3: empty problem M=%d N=%d K=%d", M, N, K);
    XGGM_REQUIRE(A && B && C, "xggm_gemm_fp8e4m3: null operand");
    XGGM_REQUIRE(K % 16 == 0 && a_rs % 16 == 0 && b_ns % 16 == 0 && a_rs >= K && b_ns >= K,
                 "xggm_gemm_fp8e4m3: K and the row strides must be multiples of 16 (K=%d, strides %lld, %lld)", K,
                 (long long)a_rs, (long long)b_ns);
    XGGM_REQUIRE(reinterpret_cast<uintptr_t>(A) % 16 == 0 && reinterpret_cast<uintptr_t>(B) % 16 == 0,
                 "xggm_gemm_fp8e4m3: operands must be 16-byte aligned");
    XGGM_REQUIRE(act >= 0 && act < XGGM_ACT_GELU_GRAD, "xggm_gemm_fp8e4m3: bad activation %d", act);
    XGGM_REQUIRE(ldc >= N, "xggm_gemm_fp8e4m3: ldc %lld < N %d", (long long)ldc, N);
    XGGM_REQUIRE((int64_t)(M - 1) * a_rs + K < OOB_OFFSET && (int64_t)(N - 1) * b_ns + K < OOB_OFFSET,
                 "xggm_gemm_fp8e4m3: operand larger than 2 GiB");
    GemmArgs g;
    g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K;
    g.a_rs = a_rs; g.a_ks = 1; g.b_ns = b_ns; g.b_ks = 1; g.ldc = ldc;
    g.a_bs = g.b_bs = g.c_bs = 0;
    g.bias = bias; g.residual = residual; g.preact = preact; g.aux = nullptr; g.colsum = nullptr; g.sqsum = nullptr;
    g.c8 = nullptr; g.c8_qscale = nullptr; g.c8_amax = nullptr; g.scale_a = g.scale_b = nullptr; g.amax_slots = 1;
    g.act = act; g.c_f32 = c_f32; g.accumulate = 0; g.alpha = 1.0f;
    g.a_mode = g.b_mode = 1; g.a_tail = g.b_tail = 0; g.a_rows = M; g.b_rows = N;
    g.xcd_swizzle = g_xcd_swizzle; g.batch = 1; g.use_glds = g_glds; SET_STAMP(g);
    switch (g_tile_override) {  // same pins as the bf16 kernels (xggm_gemm_set_tile); default 64 x 64, depth 4
        case 3: return launch_fp8_tile<128, 64, 2>(g, scale_a, scale_b, stream);
        case 5: return launch_fp8_tile<128, 128, 2>(g, scale_a, scale_b, stream);
        case 1: return launch_fp8_tile<64, 64, 2>(g, scale_a, scale_b, stream);
        default: return launch_fp8_tile<64, 64, 4>(g, scale_a, scale_b, stream);
    }
}

// test/diagnostic hook: route bf16 GEMMs through the generic kernel (1) or the tuned one (0)
extern "C" int xggm_gemm_set_generic(int on) {
    g_force_generic = on != 0;
    return XGGM_OK;
}

extern "C" int xggm_gemm_set_tile(int variant) {
    g_tile_override = variant & 0xff;
    g_xcd_swizzle = (variant & 0x100) ? 0 : 1;
    g_glds = (variant & 0x400) ? 0 : 1;
    g_no_8w = (variant & 0x4000) ? 1 : 0;
    g_single_grouped = (variant & 0x8000) ? 0 : 1;
    g_glds_stages = (variant & 0x800) ? 2 : (variant & 0x1000) ? 3 : (variant & 0x2000) ? 4 : 0;
    return XGGM_OK;
}

// ---- grouped entry point -------------------------------------------------------------------------
namespace {
GemmArgs from_problem(const xggm_gemm_problem& p) {
    GemmArgs g;
    g.A = p.A; g.B = p.B; g.C = p.C; g.M = p.M; g.N = p.N; g.K = p.K;
    g.a_rs = p.a_rs; g.a_ks = p.a_ks; g.b_ns = p.b_ns; g.b_ks = p.b_ks; g.ldc = p.ldc;
    g.a_bs = p.a_bs; g.b_bs = p.b_bs; g.c_bs = p.c_bs;
    g.bias = p.bias; g.residual = p.residual; g.preact = p.preact; g.aux = p.aux; g.colsum = p.colsum; g.sqsum = p.sqsum;
    g.c8 = reinterpret_cast<unsigned char*>(p.c8); g.c8_qscale = p.c8_qscale; g.c8_amax = p.c8_amax;
    g.amax_slots = p.amax_slots > 1 ? p.amax_slots : 1;
    g.scale_a = p.scale_a; g.scale_b = p.scale_b;
    g.act = p.act; g.c_f32 = p.c_f32; g.accumulate = p.accumulate; g.alpha = p.alpha;
    g.a_mode = g.b_mode = 0; g.xcd_swizzle = g_xcd_swizzle; g.batch = p.batch; g.use_glds = g_glds; SET_STAMP(g);
    return g;
}

// tile choice from a launch-time model fitted to tools/gemm_ktime.py and tools/bench_gemm.py on
// MI355X (microseconds): T = max(a + b r + s W, f + k nk_max), with r = tiles per CU,
// W = k-tiles per CU and the second term the longest single tile (one CU's critical
// path: long-K problems need small tiles).  Big tiles run the k-loop ~1.3x more efficiently
// per flop, small ones have the shorter critical path and fill the CUs of a small group.
// `kt`: reduction elements per k-tile (64 bf16, 128 e4m3: the same bytes).
int pick_group_tile(const GroupArgs& ga, int n, int kt) {
    struct TileModel { int bm, bn; float a, b, s, f, k; };
    static const TileModel models[3] = {{64, 64, 3.3f, 1.2f, 0.19f, 5.0f, 0.28f},
                                        {128, 64, 4.2f, 2.0f, 0.37f, 5.6f, 0.56f},
                                        {128, 128, 2.3f, 6.1f, 0.71f, 8.4f, 0.75f}};
    static TileModel tuned[3];
    static bool tuned_init = false;
    if (!tuned_init) {  // A/B hook: XGGM_TILE_MODEL="a,b,s,f,k;a,b,s,f,k;a,b,s,f,k" overrides the fitted constants
        for (int c = 0; c < 3; ++c) tuned[c] = models[c];
        if (const char* e = getenv("XGGM_TILE_MODEL")) {
            float q[15];
            if (sscanf(e, "%f,%f,%f,%f,%f;%f,%f,%f,%f,%f;%f,%f,%f,%f,%f", q, q + 1, q + 2, q + 3, q + 4, q + 5, q + 6, q + 7,
                       q + 8, q + 9, q + 10, q + 11, q + 12, q + 13, q + 14) == 15)
                for (int c = 0; c < 3; ++c) {
                    tuned[c].a = q[5 * c]; tuned[c].b = q[5 * c + 1]; tuned[c].s = q[5 * c + 2];
                    tuned[c].f = q[5 * c + 3]; tuned[c].k = q[5 * c + 4];
                }
        }
        tuned_init = true;
    }
    int v = g_group_tile;
    if (v == 0) {
        float best = 0.f;
        for (int c = 0; c < 3; ++c) {
            const TileModel& tm = tuned[c];
            double tiles = 0, work = 0;
            int nkmax = 0;
            for (int i = 0; i < n; ++i) {
                const double t = (double)ceil_div(ga.p[i].M, tm.bm) * ceil_div(ga.p[i].N, tm.bn) * ga.p[i].batch;
                const int nk = ceil_div(ga.p[i].K, kt);
                tiles += t;
                work += t * nk;
                nkmax = std::max(nkmax, nk);
            }
            const float r = std::max(1.0, tiles / 256.0);
            const float est = std::max(tm.a + tm.b * r + tm.s * (float)(work / 256.0), tm.f + tm.k * nkmax);
            if (v == 0 || est < best) {
                best = est;
                v = c + 1;
            }
        }
    }
    return v;
}

template <typename T> int grouped(const xggm_gemm_problem* probs, int n, hipStream_t stream) {
    XGGM_REQUIRE(probs && n > 0, "xggm_gemm_grouped: no problems");
    bool fast = sizeof(T) == 2 && !g_force_generic && n <= MAX_GROUP;
    GroupArgs ga;
    ga.nprob = n;
    for (int i = 0; i < n && fast; ++i) {
        GemmArgs g = from_problem(probs[i]);
        if (!(g.M > 0 && g.N > 0 && g.K > 0 && g.batch > 0 && g.A && g.B && g.C)) fast = false;
        g.a_mode = pick_mode<T>(g.A, g.a_rs, g.a_ks, g.a_bs, g.M, g.K, &g.a_tail, &g.a_rows);
        g.b_mode = pick_mode<T>(g.B, g.b_ns, g.b_ks, g.b_bs, g.N, g.K, &g.b_tail, &g.b_rows);
        if (g.a_mode == 0 || g.b_mode == 0) fast = false;
        ga.p[i] = g;
    }
    // longest k-loops first: their tiles start early and the short ones fill the tail
    for (int i = 1; i < n && fast; ++i)
        for (int j = i; j > 0 && ga.p[j].K > ga.p[j - 1].K; --j) std::swap(ga.p[j], ga.p[j - 1]);
    if (!fast || (n == 1 && !(g_single_grouped && glds_ok<false>(ga.p[0])))) {  // odd shapes, fp32 mode or a single problem: one launch each
        for (int i = 0; i < n; ++i)
            if (int e = launch<T>(from_problem(probs[i]), probs[i].batch, stream)) return e;
        return XGGM_OK;
    }
    int v = pick_group_tile(ga, n, 64);
    if (g_group_tile == 0 && sizeof(T) == 2) {
        // 128 x 128 on eight waves (two per SIMD inside one workgroup, a third fewer LDS bytes per flop than two
        // 128 x 64 tiles): measured faster exactly where one round of such tiles covers the chip and the k-loop is
        // long enough to matter -- the fused QKV forward pair, 17.6 -> 15.5 us; slower where the tiles need a second
        // round (FFN forward, 336 tiles) or leave most CUs idle (tools/gemm_phase_report.py)
        int64_t t128 = 0;
        bool kmaj = true;
        int nkmin = 1 << 30;
        for (int i = 0; i < n; ++i) {
            t128 += (int64_t)ceil_div(ga.p[i].M, 128) * ceil_div(ga.p[i].N, 128) * ga.p[i].batch;
            kmaj = kmaj && ga.p[i].a_mode == 1 && ga.p[i].b_mode == 1;
            nkmin = std::min(nkmin, ceil_div(ga.p[i].K, 64));
        }
        if (kmaj && t128 > 200 && t128 <= 256 && nkmin >= 12 && !g_no_8w) v = 4;
        // From 1.75 128 x 128 tiles per CU: the four-wave 128 x 128 tile (two workgroups per CU, 64 x 64 per wave = half the
        // LDS fragment bytes per MFMA of the other tiles).  The cost model above was fitted at 32 samples, where only the
        // FFN2 backward group is this large (and the in-step tuner pinned it to this tile); at 64 samples and at the
        // reference's batches of 92 / 96 the tuner found the same for every heavy launch -- FFN1 forward pair 58.9 -> 48.8 us,
        // FFN2 backward group 95.6 -> 77.8, QKV forward pair 38.4 -> 35.0 (tools/tune_gemm.py --batch 64 / 92, --order gqa
        // --batch 96) -- and no launch of that size where another tile won.  Below, the tile loses: 372 tiles (FFN1 backward
        // group at 32 samples, 48-deep k-loops on few tiles) 39.3 us against 30.3.
        else if (t128 >= 448) v = 3;
    }
    // (measured and not kept, round 3: 128 x 256 and 256 x 128 on eight waves -- slower on every launch of the step;
    // 192 x 128 on four waves of 96 x 64, the vendor library's tile for the FFN1 shape -- 2 us faster on the FFN1
    // forward pair in isolation, never the fastest inside the step, 91 spilled registers.  The epilogues above stay
    // general in the tile shape: whole 64-row blocks, 32-row column-sum blocks.)
#ifdef XGGM_BIG_TILES  // experiment builds only (make alt ALT=-DXGGM_BIG_TILES): 128 x 256 / 256 x 128 on eight waves
    if (v == 5) return launch_grouped_tile<128, 256, 8>(ga, stream);
    if (v == 6) return launch_grouped_tile<256, 128, 8>(ga, stream);
#endif
    if (v >= 7 && v <= 9) {  // role k-loop: whole k-tiles in every problem, otherwise the four-wave tile of the same shape
        bool ok = true;
        for (int i = 0; i < n; ++i) ok = ok && glds_ok<false>(ga.p[i]);
        if (ok)
            return v == 7 ? launch_grouped_role<128, 128>(ga, stream)
                          : v == 8 ? launch_grouped_role<128, 64>(ga, stream) : launch_grouped_role<64, 64>(ga, stream);
        v = v == 7 ? 3 : v == 8 ? 2 : 1;
    }
    if (v == 4) return launch_grouped_tile<128, 128, 8>(ga, stream);
    if (v == 3) return launch_grouped_tile<128, 128>(ga, stream);
    if (v == 2) return launch_grouped_tile<128, 64>(ga, stream);
    return launch_grouped_tile<64, 64>(ga, stream);
}
}  // namespace

extern "C" int xggm_gemm_grouped_fp8e4m3(const xggm_gemm_problem* probs, int n, hipStream_t stream) {
    XGGM_REQUIRE(probs && n > 0 && n <= MAX_GROUP, "xggm_gemm_grouped_fp8e4m3: 1..%d problems per launch (n = %d)", MAX_GROUP, n);
    GroupArgs ga;
    ga.nprob = n;
    for (int i = 0; i < n; ++i) {
        GemmArgs g = from_problem(probs[i]);
        XGGM_REQUIRE(g.M > 0 && g.N > 0 && g.K > 0 && g.batch > 0 && g.A && g.B && g.C,
                     "xggm_gemm_grouped_fp8e4m3: empty problem or null operand (%d)", i);
        XGGM_REQUIRE(g.a_ks == 1 && g.b_ks == 1, "xggm_gemm_grouped_fp8e4m3: e4m3 operands are k-contiguous");
        XGGM_REQUIRE(g.K % 16 == 0 && g.a_rs % 16 == 0 && g.b_ns % 16 == 0 && g.a_rs >= g.K && g.b_ns >= g.K &&
                         g.a_bs % 16 == 0 && g.b_bs % 16 == 0,
                     "xggm_gemm_grouped_fp8e4m3: K, row and batch strides must be multiples of 16 (K = %d)", g.K);
        XGGM_REQUIRE(reinterpret_cast<uintptr_t>(g.A) % 16 == 0 && reinterpret_cast<uintptr_t>(g.B) % 16 == 0,
                     "xggm_gemm_grouped_fp8e4m3: operands must be 16-byte aligned");
        XGGM_REQUIRE(g.act >= 0 && g.act < XGGM_ACT_GELU_GRAD && !g.aux && !g.colsum && !g.sqsum && !g.accumulate,
                     "xggm_gemm_grouped_fp8e4m3: forward epilogues only");
        XGGM_REQUIRE(g.ldc >= g.N, "xggm_gemm_grouped_fp8e4m3: ldc %lld < N %d", (long long)g.ldc, g.N);
        XGGM_REQUIRE(!g.c8 || (g.ldc % 8 == 0 && g.N % 8 == 0 && reinterpret_cast<uintptr_t>(g.c8) % 8 == 0 && !g.c_f32),
                     "xggm_gemm_grouped_fp8e4m3: the e4m3 copy needs an 8-aligned bf16 output");
        XGGM_REQUIRE((int64_t)(g.M - 1) * g.a_rs + (int64_t)(g.batch - 1) * g.a_bs + g.K < OOB_OFFSET &&
                         (int64_t)(g.N - 1) * g.b_ns + (int64_t)(g.batch - 1) * g.b_bs + g.K < OOB_OFFSET,
                     "xggm_gemm_grouped_fp8e4m3: operand larger than 2 GiB");
        g.a_mode = g.b_mode = 1;
        g.a_tail = g.b_tail = 0;
        g.a_rows = g.M;
        g.b_rows = g.N;
        ga.p[i] = g;
    }
    for (int i = 1; i < n; ++i)  // longest k-loops first
        for (int j = i; j > 0 && ga.p[j].K > ga.p[j - 1].K; --j) std::swap(ga.p[j], ga.p[j - 1]);
    int v = pick_group_tile(ga, n, 128);
    if (g_group_tile == 0) {
        // 128 x 128 on eight waves where ONE round of such tiles covers the chip (the fused QKV pair: 252 tiles), as
        // for the bf16 products -- but unlike there it buys nothing (k-loops of 6 e4m3 k-tiles): XGGM_FP8_8W=1 turns it on
        static const bool use8 = getenv("XGGM_FP8_8W") && atoi(getenv("XGGM_FP8_8W")) == 1;  // measured: 11.18 vs 11.16 ms per iteration without it -- off
        int64_t t128 = 0;
        for (int i = 0; i < n; ++i) t128 += (int64_t)ceil_div(ga.p[i].M, 128) * ceil_div(ga.p[i].N, 128) * ga.p[i].batch;
        if (use8 && t128 > 200 && t128 <= 256 && !g_no_8w) v = 4;
    }
    if (v == 4) return launch_grouped_fp8_tile<128, 128, 8>(ga, stream);
    if (v == 3) return launch_grouped_fp8_tile<128, 128>(ga, stream);
    if (v == 2) return launch_grouped_fp8_tile<128, 64>(ga, stream);
    return launch_grouped_fp8_tile<64, 64>(ga, stream);
}
extern "C" int xggm_gemm_grouped_bf16(const xggm_gemm_problem* probs, int n, hipStream_t stream) {
    return grouped<bf16>(probs, n, stream);
}
extern "C" int xggm_gemm_grouped_f32(const xggm_gemm_problem* probs, int n, hipStream_t stream) {
    return grouped<float>(probs, n, stream);
}
#ifdef XGGM_STAMP
// what the runtime thinks: resident workgroups per CU of the grouped kernels at a given dynamic LDS size
extern "C" int xggm_gemm_occupancy(int tile, int lds_bytes, int* out) {
    int n = -1;
    hipError_t e;
    if (tile == 3) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gemm_grouped_kernel<128, 128>, NT, lds_bytes);
    else if (tile == 2) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gemm_grouped_kernel<128, 64>, NT, lds_bytes);
    else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gemm_grouped_kernel<64, 64>, NT, lds_bytes);
    hipDeviceProp_t pr;
    (void)hipGetDeviceProperties(&pr, 0);
    out[0] = n;
    out[1] = (int)pr.maxSharedMemoryPerMultiProcessor;
    out[2] = (int)pr.sharedMemPerBlock;
    out[3] = pr.regsPerMultiprocessor;
    out[4] = pr.regsPerBlock;
    out[5] = (int)e;
    return XGGM_OK;
}
extern "C" int xggm_gemm_set_ablate(int bits) {
    g_ablate = bits;
    return XGGM_OK;
}
extern "C" int xggm_gemm_set_stamp(long long* buf) {
    g_stamp = buf;
    return XGGM_OK;
}
#endif
extern "C" int xggm_gemm_set_group_tile(int v) {
    g_group_tile = v;
    return XGGM_OK;
}
