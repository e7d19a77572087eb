// bf16 attention core on the matrix cores (the fp32 path keeps the scalar kernels of attention.hip).
//
// One 512-thread workgroup per (sample, head).  Q, K, V (and dO) tiles are staged in LDS as bf16 with
// an 80-element row stride, a layout that serves BOTH operand shapes of the 16x16x32 MFMA: rows read
// with ds_read_b128 when the reduction index is the tile's column (S = Q K^T, dP = dO V^T) and
// columns read with the transposing ds_read_b64_tr_b16 when the reduction index is the tile's row
// (O = P V, dQ = dS K, dK = dS^T Q, dV = P^T dO).  Scores / probabilities pass through LDS once as
// fp32 for the wave-per-row softmax (shuffle reductions) and return as bf16 operands.  All five
// products of the backward are 108 MFMAs per (b, h) at S = 36 instead of ~420 K scalar FMAs.
// Sequences are padded to the MFMA granularity with zero rows, which contribute nothing.
#include "common.h"
#include "xggm.h"

namespace {

constexpr int NT = 512;  // 8 waves (measured: 256 threads 1.97 ms, 512 1.46 ms, 1024 1.77 ms for the backward launches of three passes): every phase (tile products, row softmax, gradient tiles) is spread over twice the waves
constexpr int TILE_IT = NT >= 512 ? 1 : 512 / NT;  // 16-byte chunks per thread of a 64 x 64 bf16 tile
constexpr int D = 64;
constexpr int LDT = D + 16;  // tile row stride in elements (160 B): b128 rows 16-B aligned; tr-reads of k-rows 8 apart share banks (2-way; the GEMM swaps half-blocks against this, see gemm.hip fast_frag -- here the phases are barrier-latency-bound)

typedef __attribute__((ext_vector_type(8))) short short8_t;
typedef __attribute__((ext_vector_type(4))) short short4_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float float4_t;

struct MArgs {
    const bf16 *q, *k, *v;
    const float* mask;
    int64_t q_rs, k_rs, v_rs, o_rs;
    int B, heads, Sq, Sk;
    float scale, p;
    const uint64_t* rng;
    uint32_t sid;
};

// one or two independent problems per launch: workgroups [0, start1) run s[0], the rest s[1]
struct MSeg {
    MArgs a;
    bf16* out;
    unsigned char* out8;  // or null: e4m3 copy of the output (operand of the fp8 output projection)
    const float* qscale;
    float* amax;
    int amax_slots;
    const bf16* d_out;
    bf16 *dq, *dk, *dv;
    int64_t dq_rs, dk_rs, dv_rs;
    float *dbq, *dbk, *dbv;
    int64_t db_bs;
};
struct MGroup {
    MSeg s[2];
    int start1;
    int total;        // workgroups of the two segments; the ones behind them read the queued weight ranges (common.h)
    PrefetchArgs pf;
#ifdef XGGM_STAMP
    long long* stamp;
#endif
};

__device__ __forceinline__ int rup(int x, int m) { return (x + m - 1) / m * m; }

// rows (<= 64) x 64 bf16 tile: two 16-byte chunks per thread.  tile_fetch only ISSUES the loads
// (addresses clamped into the tile, no branches) so the loads of all tiles of a workgroup are in
// flight together; tile_commit zeroes the padding rows and writes LDS (row stride LDT).
struct TileRegs { short8_t v[TILE_IT]; };
__device__ __forceinline__ TileRegs tile_fetch(const bf16* base, int64_t rs, int valid, int tid) {
    TileRegs t;
#pragma unroll
    for (int it = 0; it < TILE_IT; ++it) {
        const int c = (tid + it * NT) & 511, r = min(c >> 3, valid - 1), cc = (c & 7) * 8;
        t.v[it] = *reinterpret_cast<const short8_t*>(base + (int64_t)r * rs + cc);
    }
    return t;
}
__device__ __forceinline__ void tile_commit(bf16* lds, const TileRegs& t, int valid, int rows, int tid) {
#pragma unroll
    for (int it = 0; it < TILE_IT; ++it) {
        const int c = tid + it * NT, r = c >> 3, cc = (c & 7) * 8;
        const short8_t zero = {};
        if (c < 512 && r < rows) *reinterpret_cast<short8_t*>(lds + r * LDT + cc) = r < valid ? t.v[it] : zero;
    }
}

// operand whose reduction index k is contiguous: lane (fr, fq) gets row row0+fr, k = k0+8fq .. +7
__device__ __forceinline__ bf16x8_t frag_rows(const bf16* lds, int ld, int row0, int k0, int lane) {
    const int fr = lane & 15, fq = lane >> 4;
    return __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const short8_t*>(lds + (row0 + fr) * ld + k0 + fq * 8));
}
// operand stored [k][n] (n contiguous): lane gets n = n0+fr, k = k0+8fq .. +7 through two transposing reads
__device__ __forceinline__ bf16x8_t frag_tr(const bf16* lds, int ld, int k0, int n0, int lane) {
    const int fr = lane & 15, fq = lane >> 4, q = fr >> 2, p = fr & 3;
    const bf16* a0 = lds + (k0 + 8 * fq + q) * ld + n0 + 4 * p;
    const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)(a0));
    const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)(a0 + 4 * ld));
    short8_t v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
    v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return __builtin_bit_cast(bf16x8_t, v);
}

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)

#ifdef XGGM_STAMP
// instrumented build (make stamp, tools/attn_stamps.py): 16 cycle-counter slots per workgroup
long long* g_attn_stamp = nullptr;
#define ASTAMP(p, slot)                                                                                      \
    do {                                                                                                     \
        if ((p) && threadIdx.x == 0) (p)[(int64_t)blockIdx.x * 16 + (slot)] = __builtin_readcyclecounter();  \
    } while (0)
#else
#define ASTAMP(p, slot)
#endif

// S = scale * Q K^T + mask  ->  Sf (fp32, row stride lds_s), tiles shared round-robin by the waves
// (staging the sample's mask row in LDS with the Q / K / V tiles instead of reading it inside the tile loop was
// measured: 7.68 -> 7.77 us forward, 13.0 -> 13.3 us backward -- the other waves cover that round trip already)
__device__ __forceinline__ void scores(const MArgs& a, const bf16* Qs, const bf16* Ks, float* Sf, int lds_s, int b, int tid) {
    const int lane = tid & 63, wid = tid >> 6, fr = lane & 15, fq = lane >> 4;
    const int tq = rup(a.Sq, 16) / 16, tk = rup(a.Sk, 16) / 16;
    for (int t = wid; t < tq * tk; t += NT / 64) {
        const int ti = t / tk, tj = t % tk;
        float4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < D; ks += 32)
            acc = MFMA(frag_rows(Qs, LDT, ti * 16, ks, lane), frag_rows(Ks, LDT, tj * 16, ks, lane), acc);
        const int j = tj * 16 + fr;
        const float mk = (a.mask && j < a.Sk) ? a.mask[(int64_t)b * a.Sk + j] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = ti * 16 + fq * 4 + r;
            if (i < a.Sq && j < a.Sk) Sf[i * lds_s + j] = acc[r] * a.scale + mk;
        }
    }
}

// dropout keep-scale of probability (i, j) of this (sample, head): a pure function of the Philox
// state, recomputed wherever it is needed instead of being parked in LDS
struct Drop {
    float p, ik;
    uint64_t seed, off, base;
    uint32_t sid;
    __device__ __forceinline__ Drop(const MArgs& a, int b, int h) : p(a.p), ik(a.p > 0.f ? 1.f / (1.f - a.p) : 1.f), seed(0), off(0),
                                                                  base(((uint64_t)b * a.heads + h) * a.Sq * a.Sk), sid(a.sid) {
        if (a.p > 0.f) {
            seed = a.rng[0];
            off = a.rng[1];
        }
    }
    __device__ __forceinline__ float scale(int i, int j, int Sk) const {
        return p > 0.f ? dropout_scale(p, ik, seed, off, sid, base + (uint64_t)i * Sk + j) : 1.f;
    }
    // keep-scales of (i, j0 .. j0 + 3): one Philox call when the four element indices share it
    __device__ __forceinline__ void scale4(int i, int j0, int Sk, float (&s)[4]) const {
        if (p <= 0.f) {
            s[0] = s[1] = s[2] = s[3] = 1.f;
        } else if ((Sk & 3) == 0) {
            dropout_scale4(p, ik, seed, off, sid, base + (uint64_t)i * Sk + j0, s);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) s[e] = dropout_scale(p, ik, seed, off, sid, base + (uint64_t)i * Sk + j0 + e);
        }
    }
};

// row softmax of Sf in place (fp32).  wave w owns rows w, w+4, ...
__device__ __forceinline__ void softmax_rows(const MArgs& a, float* Sf, int lds_s, int tid) {
    const int lane = tid & 63, wid = tid >> 6;
    for (int i = wid; i < a.Sq; i += NT / 64) {
        const float v = lane < a.Sk ? Sf[i * lds_s + lane] : -INFINITY;
        const float m = wave_max(v);
        const float e = lane < a.Sk ? __expf(v - m) : 0.f;
        const float sum = wave_sum(e);
        if (lane < a.Sk) Sf[i * lds_s + lane] = e / sum;
    }
}

// ---- row phase: EIGHT lanes per score row, no wave-wide shuffles -------------------------------------------------
// The first version gave every row to a whole wave (wave_max / wave_sum = 12 ds_bpermute round trips per row, then a
// second pass over all elements for dropout + bf16): in-kernel stamps put 66 % of the forward and 58 % of the
// backward there.  Here lane 8 r + part owns the column groups part and part + 8 (four consecutive columns each, 8
// values, kept in registers) of row r: maximum and sum are finished with three DPP moves (quad permutes, half-row mirror), one
// Philox call serves the four dropout decisions of a group (their element indices share idx >> 2 when Sk % 4 == 0),
// and the probabilities leave as 8-byte bf16 stores.  Rows / columns outside the problem are written as zeros.
__device__ __forceinline__ float quad_xor1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float quad_xor2(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
}
// lanes i <-> 7 - i of every group of eight (DPP row_half_mirror): after the two quad steps each quad holds its own
// result, so the mirrored lane supplies the other quad's
__device__ __forceinline__ float half_mirror(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
}
__device__ __forceinline__ float quad_max(float v) {  // over the ROW_L = 8 lanes of a row
    v = fmaxf(v, quad_xor1(v));
    v = fmaxf(v, quad_xor2(v));
    return fmaxf(v, half_mirror(v));
}
__device__ __forceinline__ float quad_sum(float v) {
    v += quad_xor1(v);
    v += quad_xor2(v);
    return v + half_mirror(v);
}
constexpr int ROW_L = 8;  // lanes per score row
constexpr int ROW_G = 2;  // column groups per lane: 8 lanes x 2 groups x 4 columns = 64 keys

__global__ __launch_bounds__(NT) void attn_fwd_mfma_kernel(MGroup G) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    if ((int)blockIdx.x >= G.total) {
        prefetch_role(G.pf, (int)blockIdx.x - G.total);
        return;
    }
    const int si = (int)blockIdx.x >= G.start1 ? 1 : 0;
    const MSeg sg = G.s[si];
    const MArgs& a = sg.a;
    bf16* out = sg.out;
    const int blk = blockIdx.x - (si ? G.start1 : 0);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, fr = lane & 15, fq = lane >> 4;
    const int b = blk / a.heads, h = blk % a.heads;
    const int Sq = a.Sq, Sk = a.Sk;
    const int RQ = rup(Sq, 16), RK = rup(Sk, 32);  // Q rows (M of every product), K/V rows (k of P V)
    const int LDP = RK + 8, lds_s = rup(Sk, 16) + 1;
    bf16* Qs = reinterpret_cast<bf16*>(smem_raw);
    bf16* Ks = Qs + RQ * LDT;
    bf16* Vs = Ks + RK * LDT;
    bf16* Pb = Vs + RK * LDT;                                    // [RQ][LDP] bf16 probabilities (dropout folded in)
    float* Sf = reinterpret_cast<float*>(Pb + RQ * LDP);         // [RQ][lds_s]
#ifdef XGGM_STAMP
    long long* stp = G.stamp;
#endif
    ASTAMP(stp, 0);
    {
        const TileRegs tq_ = tile_fetch(a.q + (int64_t)b * Sq * a.q_rs + h * D, a.q_rs, Sq, tid);
        const TileRegs tk_ = tile_fetch(a.k + (int64_t)b * Sk * a.k_rs + h * D, a.k_rs, Sk, tid);
        const TileRegs tv_ = tile_fetch(a.v + (int64_t)b * Sk * a.v_rs + h * D, a.v_rs, Sk, tid);
        tile_commit(Qs, tq_, Sq, RQ, tid);
        tile_commit(Ks, tk_, Sk, RK, tid);
        tile_commit(Vs, tv_, Sk, RK, tid);
    }
    __syncthreads();
    ASTAMP(stp, 1);
    scores(a, Qs, Ks, Sf, lds_s, b, tid);
    __syncthreads();
    ASTAMP(stp, 2);
    ASTAMP(stp, 3);
    {
        // softmax + dropout + bf16 operand copy of the probabilities, eight lanes per row (see above)
        const Drop dr(a, b, h);
        const int row = tid / ROW_L, part = tid % ROW_L, NG = RK >> 2;
        if (row < RQ) {
            float v[ROW_G][4];
            float m = -INFINITY;
#pragma unroll
            for (int gi = 0; gi < ROW_G; ++gi) {
                const int j0 = 4 * (part + ROW_L * gi);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool in = row < Sq && j0 + e < Sk;
                    v[gi][e] = in ? Sf[row * lds_s + j0 + e] : -INFINITY;
                    m = fmaxf(m, v[gi][e]);
                }
            }
            m = quad_max(m);
            float sum = 0.f;
#pragma unroll
            for (int gi = 0; gi < ROW_G; ++gi)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[gi][e] = v[gi][e] > -INFINITY ? __expf(v[gi][e] - m) : 0.f;
                    sum += v[gi][e];
                }
            sum = quad_sum(sum);
            const float inv = sum > 0.f ? 1.f / sum : 0.f;
#pragma unroll
            for (int gi = 0; gi < ROW_G; ++gi) {
                const int g = part + ROW_L * gi, j0 = 4 * g;
                if (g < NG) {
                    float ds[4] = {1.f, 1.f, 1.f, 1.f};
                    if (row < Sq && j0 < Sk) dr.scale4(row, j0, Sk, ds);
                    short4_t pk;
#pragma unroll
                    for (int e = 0; e < 4; ++e) pk[e] = __builtin_bit_cast(short, __float2bfloat16(v[gi][e] * inv * ds[e]));
                    *reinterpret_cast<short4_t*>(Pb + row * LDP + j0) = pk;
                }
            }
        }
    }
    __syncthreads();
    ASTAMP(stp, 4);
    // O = P V : tiles (ti, tc), k = j over RK.  The tiles go back through LDS (the Q image is dead since the scores)
    // so that the context leaves as whole 16-byte row pieces -- 128 contiguous bytes per (row, head) -- instead of
    // 2-byte stores at a row stride, and, for the fp8 forward, as e4m3 next to it in the same pass.
    const int tq = RQ / 16;
    bf16* Os = Qs;
    for (int t = wid; t < tq * 4; t += NT / 64) {
        const int ti = t >> 2, tc = t & 3;
        float4_t acc = {0.f, 0.f, 0.f, 0.f};
        for (int ks = 0; ks < RK; ks += 32)
            acc = MFMA(frag_rows(Pb, LDP, ti * 16, ks, lane), frag_tr(Vs, LDT, ks, tc * 16, lane), acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) Os[(ti * 16 + fq * 4 + r) * LDT + tc * 16 + fr] = __float2bfloat16(acc[r]);
    }
    __syncthreads();
    ASTAMP(stp, 5);
    const Q8 qs(sg.out8 ? sg.qscale : nullptr);
    const float q8 = qs.q;
    float amax8 = 0.f;
    for (int c = tid; c < Sq * 8; c += NT) {
        const int i = c >> 3, cc = (c & 7) * 8;
        const short8_t v = *reinterpret_cast<const short8_t*>(Os + i * LDT + cc);
        const int64_t o = ((int64_t)b * Sq + i) * a.o_rs + h * D + cc;
        *reinterpret_cast<short8_t*>(out + o) = v;
        if (sg.out8) {
            float f[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                f[e] = __bfloat162float(__builtin_bit_cast(bf16, (short)v[e]));
                amax8 = fmaxf(amax8, fabsf(f[e]));
            }
            *reinterpret_cast<int2*>(sg.out8 + o) = make_int2(pack4_e4m3(f[0], f[1], f[2], f[3], q8), pack4_e4m3(f[4], f[5], f[6], f[7], q8));
        }
    }
    ASTAMP(stp, 6);
    if (sg.out8 && sg.amax) {
        __shared__ float red8[NT / 64];
        const float wv = wave_max(amax8);
        if (lane == 0) red8[wid] = wv;
        __syncthreads();
        if (tid == 0) {
            float bm = red8[0];
#pragma unroll
            for (int w = 1; w < NT / 64; ++w) bm = fmaxf(bm, red8[w]);
            if (bm > qs.thr) amax_record(sg.amax, sg.amax_slots, (int)blockIdx.x, bm);
        }
    }
}

// store a 16x16 gradient tile (rows row0.., 16 columns at col0) and fold its column sums into csum.  The tile comes out
// of an MFMA with SWAPPED operands: lane (fr, fq) holds row fr, columns 4 fq .. 4 fq + 3 -- one 8-byte store per lane
// (was four 2-byte ones) and column sums by DPP moves inside the 16-lane rows (was two ds_bpermute round trips).
__device__ __forceinline__ void store_grad_tile(const float4_t& acc, bf16* dst, int64_t rs, int row_base, int rows_valid,
                                                int row0, int col0, float* csum, int lane) {
    const int fr = lane & 15, fq = lane >> 4;
    const int i = row0 + fr;
    const bool ok = i < rows_valid;
    if (ok) {
        short4_t pk;
#pragma unroll
        for (int r = 0; r < 4; ++r) pk[r] = __builtin_bit_cast(short, __float2bfloat16(acc[r]));
        *reinterpret_cast<short4_t*>(dst + (int64_t)(row_base + i) * rs + col0 + 4 * fq) = pk;
    }
    if (csum) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float s = ok ? acc[r] : 0.f;
            s += dpp_move<0xB1>(s);
            s += dpp_move<0x4E>(s);
            s += dpp_move<0x141>(s);
            s += dpp_move<0x140>(s);  // every lane of the 16-lane row holds the column's sum over the tile's rows
            if (fr == 0) csum[col0 + 4 * fq + r] = s;  // this row tile's own slot: no atomics, see the kernel's tail
        }
    }
}

__global__ __launch_bounds__(NT) void attn_bwd_mfma_kernel(MGroup G) {
    // bias gradients: column sums of dQ / dK / dV, one slot per 16-row tile (<= 4: S <= 64), written by the wave that
    // owns the tile and added in tile order at the end -- the same bits whatever the scheduling
    __shared__ float csum[3][4][D];
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    if ((int)blockIdx.x >= G.total) {
        prefetch_role(G.pf, (int)blockIdx.x - G.total);
        return;
    }
    const int si = (int)blockIdx.x >= G.start1 ? 1 : 0;
    const MSeg sg = G.s[si];
    const MArgs& a = sg.a;
    const bf16* d_out = sg.d_out;
    bf16 *dq = sg.dq, *dk = sg.dk, *dv = sg.dv;
    const int64_t dq_rs = sg.dq_rs, dk_rs = sg.dk_rs, dv_rs = sg.dv_rs;
    float *dbq = sg.dbq, *dbk = sg.dbk, *dbv = sg.dbv;
    const int blk = blockIdx.x - (si ? G.start1 : 0);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, fr = lane & 15, fq = lane >> 4;
    const int b = blk / a.heads, h = blk % a.heads;
    const int Sq = a.Sq, Sk = a.Sk;
    const int RQ = rup(Sq, 32), RK = rup(Sk, 32);  // both row counts also serve as reduction lengths here
    const int LDP = RK + 8, R16 = rup(Sq, 16), lds_s = rup(Sk, 16) + 1;
    bf16* Qs = reinterpret_cast<bf16*>(smem_raw);
    bf16* Ks = Qs + RQ * LDT;
    bf16* Vs = Ks + RK * LDT;
    bf16* dOs = Vs + RK * LDT;
    bf16* dSb = dOs + RQ * LDT;   // [RQ][LDP] bf16: dS (scaled)
    bf16* Pdb = dSb + RQ * LDP;   // [RQ][LDP] bf16: P with dropout folded in
    float* Sf = reinterpret_cast<float*>(Pdb + RQ * LDP);  // [R16][lds_s]: S then P
    float* dPf = Sf + R16 * lds_s;
#ifdef XGGM_STAMP
    long long* stp = G.stamp;
#endif
    ASTAMP(stp, 0);
    {
        const TileRegs tq_ = tile_fetch(a.q + (int64_t)b * Sq * a.q_rs + h * D, a.q_rs, Sq, tid);
        const TileRegs tk_ = tile_fetch(a.k + (int64_t)b * Sk * a.k_rs + h * D, a.k_rs, Sk, tid);
        const TileRegs tv_ = tile_fetch(a.v + (int64_t)b * Sk * a.v_rs + h * D, a.v_rs, Sk, tid);
        const TileRegs to_ = tile_fetch(d_out + (int64_t)b * Sq * a.o_rs + h * D, a.o_rs, Sq, tid);
        tile_commit(Qs, tq_, Sq, RQ, tid);
        tile_commit(Ks, tk_, Sk, RK, tid);
        tile_commit(Vs, tv_, Sk, RK, tid);
        tile_commit(dOs, to_, Sq, RQ, tid);
    }
    __syncthreads();
    ASTAMP(stp, 1);
    scores(a, Qs, Ks, Sf, lds_s, b, tid);
    // dP = dO V^T (both operands read by rows: the reduction index is the feature)
    {
        const int tq = rup(Sq, 16) / 16, tk = rup(Sk, 16) / 16;
        for (int t = wid; t < tq * tk; t += NT / 64) {
            const int ti = t / tk, tj = t % tk;
            float4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < D; ks += 32)
                acc = MFMA(frag_rows(dOs, LDT, ti * 16, ks, lane), frag_rows(Vs, LDT, tj * 16, ks, lane), acc);
            const int j = tj * 16 + fr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = ti * 16 + fq * 4 + r;
                if (i < Sq && j < Sk) dPf[i * lds_s + j] = acc[r];
            }
        }
    }
    __syncthreads();
    ASTAMP(stp, 2);
    ASTAMP(stp, 3);
    // softmax recomputed, then dS = P (dP D - rowsum(dP D P)) scale and Pd = P D, eight lanes per row (see the forward
    // kernel).  Rows / columns outside the problem are zero.
    {
        const Drop dr(a, b, h);
        const int row = tid / ROW_L, part = tid % ROW_L, NG = RK >> 2;
        if (row < RQ) {
            float v[ROW_G][4], dp[ROW_G][4], dm[ROW_G][4];
            float m = -INFINITY;
#pragma unroll
            for (int gi = 0; gi < ROW_G; ++gi) {
                const int j0 = 4 * (part + ROW_L * gi);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool in = row < Sq && j0 + e < Sk;
                    v[gi][e] = in ? Sf[row * lds_s + j0 + e] : -INFINITY;
                    dp[gi][e] = in ? dPf[row * lds_s + j0 + e] : 0.f;
                    m = fmaxf(m, v[gi][e]);
                }
            }
            m = quad_max(m);
            float sum = 0.f;
#pragma unroll
            for (int gi = 0; gi < ROW_G; ++gi)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[gi][e] = v[gi][e] > -INFINITY ? __expf(v[gi][e] - m) : 0.f;
                    sum += v[gi][e];
                }
            sum = quad_sum(sum);
            const float inv = sum > 0.f ? 1.f / sum : 0.f;
            float rs = 0.f;
#pragma unroll
            for (int gi = 0; gi < ROW_G; ++gi) {
                const int j0 = 4 * (part + ROW_L * gi);
                dm[gi][0] = dm[gi][1] = dm[gi][2] = dm[gi][3] = 1.f;
                if (row < Sq && j0 < Sk) dr.scale4(row, j0, Sk, dm[gi]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[gi][e] *= inv;          // P
                    dp[gi][e] *= dm[gi][e];   // dP D
                    rs += v[gi][e] * dp[gi][e];
                }
            }
            rs = quad_sum(rs);
#pragma unroll
            for (int gi = 0; gi < ROW_G; ++gi) {
                const int g = part + ROW_L * gi, j0 = 4 * g;
                if (g < NG) {
                    short4_t ks, kp;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        ks[e] = __builtin_bit_cast(short, __float2bfloat16(v[gi][e] * (dp[gi][e] - rs) * a.scale));
                        kp[e] = __builtin_bit_cast(short, __float2bfloat16(v[gi][e] * dm[gi][e]));
                    }
                    *reinterpret_cast<short4_t*>(dSb + row * LDP + j0) = ks;
                    *reinterpret_cast<short4_t*>(Pdb + row * LDP + j0) = kp;
                }
            }
        }
    }
    __syncthreads();
    ASTAMP(stp, 4);
    const int tq = rup(Sq, 16) / 16, tk = rup(Sk, 16) / 16;
    // 4 column tiles each for dQ (tq row tiles), dK and dV (tk row tiles): one list shared by the waves
    const int n_dq = tq * 4, n_dk = tk * 4;
    for (int t = wid; t < n_dq + 2 * n_dk; t += NT / 64) {
        float4_t acc = {0.f, 0.f, 0.f, 0.f};
        if (t < n_dq) {
            const int ti = t >> 2, tc = t & 3;  // dQ = dS K: k = j
            for (int ks = 0; ks < RK; ks += 32)
                acc = MFMA(frag_tr(Ks, LDT, ks, tc * 16, lane), frag_rows(dSb, LDP, ti * 16, ks, lane), acc);
            store_grad_tile(acc, dq + h * D, dq_rs, b * Sq, Sq, ti * 16, tc * 16, dbq ? csum[0][ti] : nullptr, lane);
        } else if (t < n_dq + n_dk) {
            const int u = t - n_dq, tj = u >> 2, tc = u & 3;  // dK = dS^T Q: k = i
            for (int ks = 0; ks < RQ; ks += 32)
                acc = MFMA(frag_tr(Qs, LDT, ks, tc * 16, lane), frag_tr(dSb, LDP, ks, tj * 16, lane), acc);
            store_grad_tile(acc, dk + h * D, dk_rs, b * Sk, Sk, tj * 16, tc * 16, dbk ? csum[1][tj] : nullptr, lane);
        } else {
            const int u = t - n_dq - n_dk, tj = u >> 2, tc = u & 3;  // dV = (P D)^T dO: k = i
            for (int ks = 0; ks < RQ; ks += 32)
                acc = MFMA(frag_tr(dOs, LDT, ks, tc * 16, lane), frag_tr(Pdb, LDP, ks, tj * 16, lane), acc);
            store_grad_tile(acc, dv + h * D, dv_rs, b * Sk, Sk, tj * 16, tc * 16, dbv ? csum[2][tj] : nullptr, lane);
        }
    }
    ASTAMP(stp, 5);
    if (dbq || dbk) {
        __syncthreads();
        const int t = threadIdx.x, kind = t / D, c = t - kind * D;
        float* dst = kind == 0 ? dbq : kind == 1 ? dbk : kind == 2 ? dbv : nullptr;
        if (dst) {
            const int nt = kind == 0 ? tq : tk;
            float s = csum[kind][0][c];
            for (int u = 1; u < nt; ++u) s += csum[kind][u][c];
            dst[(int64_t)b * sg.db_bs + h * D + c] = s;  // partial row of this sample: summed over the batch by the reduce launch
        }
    }
}

template <typename K> void allow_big_lds(K kernel, size_t lds) {
    if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

}  // namespace

// called by attention.hip for bf16 storage (arguments already validated there): n = 1 or 2 problems
namespace {
MSeg make_seg(const xggm_attn_problem& q, const uint64_t* rng) {
    MSeg g;
    g.a = MArgs{(const bf16*)q.q, (const bf16*)q.k, (const bf16*)q.v, q.mask, q.q_rs, q.k_rs, q.v_rs, q.o_rs,
                q.B, q.heads, q.Sq, q.Sk, q.scale, q.p, rng, q.sid};
    g.out = (bf16*)q.out;
    g.out8 = (unsigned char*)q.out8;
    g.qscale = q.qscale;
    g.amax = q.amax;
    g.amax_slots = q.amax_slots > 1 ? q.amax_slots : 1;
    g.d_out = (const bf16*)q.d_out;
    g.dq = (bf16*)q.dq; g.dk = (bf16*)q.dk; g.dv = (bf16*)q.dv;
    g.dq_rs = q.dq_rs; g.dk_rs = q.dk_rs; g.dv_rs = q.dv_rs;
    g.dbq = q.dbq; g.dbk = q.dbk; g.dbv = q.dbv; g.db_bs = q.db_bs;
    return g;
}
}  // namespace

int xggm_attn_fwd_mfma_group(const xggm_attn_problem* probs, int n, const uint64_t* rng, hipStream_t st) {
    MGroup G;
    size_t lds = 0;
    int total = 0;
    for (int i = 0; i < n; ++i) {
        const xggm_attn_problem& q = probs[i];
        G.s[i] = make_seg(q, rng);
        const int RQ = (q.Sq + 15) / 16 * 16, RK = (q.Sk + 31) / 32 * 32, lds_s = (q.Sk + 15) / 16 * 16 + 1;
        lds = std::max(lds, sizeof(bf16) * ((size_t)(RQ + 2 * RK) * LDT + (size_t)RQ * (RK + 8)) + sizeof(float) * RQ * lds_s);
        if (i == 1) G.start1 = total;
        total += q.B * q.heads;
    }
    if (n == 1) {
        G.s[1] = G.s[0];
        G.start1 = total;
    }
#ifdef XGGM_STAMP
    G.stamp = g_attn_stamp;
#endif
    allow_big_lds(attn_fwd_mfma_kernel, lds);
    G.total = total;
    G.pf = xggm_take_prefetch();
    hipLaunchKernelGGL(attn_fwd_mfma_kernel, dim3(total + G.pf.blocks), dim3(NT), lds, st, G);
    return xggm_check_launch("xggm_attn_fwd(mfma)");
}

int xggm_attn_bwd_mfma_group(const xggm_attn_problem* probs, int n, const uint64_t* rng, hipStream_t st) {
    MGroup G;
    size_t lds = 0;
    int total = 0;
    for (int i = 0; i < n; ++i) {
        const xggm_attn_problem& q = probs[i];
        G.s[i] = make_seg(q, rng);
        const int RQ = (q.Sq + 31) / 32 * 32, RK = (q.Sk + 31) / 32 * 32, lds_s = (q.Sk + 15) / 16 * 16 + 1;
        const int R16 = (q.Sq + 15) / 16 * 16;
        lds = std::max(lds, sizeof(bf16) * ((size_t)(2 * RQ + 2 * RK) * LDT + 2 * (size_t)RQ * (RK + 8)) +
                                sizeof(float) * 2 * R16 * lds_s);
        if (i == 1) G.start1 = total;
        total += q.B * q.heads;
    }
    if (n == 1) {
        G.s[1] = G.s[0];
        G.start1 = total;
    }
#ifdef XGGM_STAMP
    G.stamp = g_attn_stamp;
#endif
    allow_big_lds(attn_bwd_mfma_kernel, lds);
    G.total = total;
    G.pf = xggm_take_prefetch();
    hipLaunchKernelGGL(attn_bwd_mfma_kernel, dim3(total + G.pf.blocks), dim3(NT), lds, st, G);
    return xggm_check_launch("xggm_attn_bwd(mfma)");
}

#ifdef XGGM_STAMP
extern "C" int xggm_attn_set_stamp(long long* buf) {
    g_attn_stamp = buf;
    return XGGM_OK;
}
#endif
