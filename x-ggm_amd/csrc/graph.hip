// Graph-generative kernels over the dense N x N adjacency of the object regions (N = 36,
// 64 in the stress configuration).  Adjacency-shaped tensors are always fp32 (5-16 KB per
// sample: they live in LDS inside the kernels and their values feed divisions, sigmoids and
// an arg-max whose index must be exact); node features are T.
//
//   xggm_aggregate        out = [out +] self*x + scale * M' @ x     M' = M | M^T | M + M^T
//                         GCNConv aggregate (src/module/gcn.py:28), GIN aggregate
//                         (src/module/gin.py:32) and every backward-through-x of them and of
//                         S = x x^T.  The adjacency tile is staged (transposed) in LDS, each
//                         thread owns one feature column and keeps all N outputs in registers:
//                         x is read from HBM once, out written once.
//   xggm_adj_regen_fwd/bwd  column-max normalisation + sigmoid + zero diagonal of S = x x^T
//                         (src/module/graph_generative_modeling.py:225-228); per-column max and
//                         first-index arg-max by wave64 shuffle reduction.
//   xggm_adj_init_fwd/bwd  encoder_adj scatter into the strict upper triangle, symmetrise, add
//                         symmetric Gaussian noise, emit grad_log_noise
//                         (src/vqa/vqacpv2.py:195-202 + src/module/graph_utils.py:162-168).
//   xggm_feature_noise    src/module/graph_utils.py:144-149
//   xggm_pool_concat_fwd/bwd  [x, tanh(mean_n nodes)]  (src/vqa/vqacpv2.py:216-218)
#include "common.h"
#include "xggm.h"

namespace {

constexpr int NT = 256;

// ------------------------------------------------------------------------------- aggregate
// out[b] = scale * M'[b] x[b] (+ self_w x[b]) (+ out[b]) on the matrix cores, fp32 in and fp32 accumulate
// (v_mfma_f32_16x16x4_f32: exact in fp32 storage mode as well).  One workgroup = one sample x 64 feature
// columns: the adjacency M' (plain / transposed / symmetrised, <= 64 x 64) and the x slab [N, 64] sit in LDS as
// fp32, wave w owns columns 16 w .. 16 w + 15 and all N / 16 row tiles.  The MFMA operands are swapped
// (D^T = x^T M'^T) so a lane ends up with FOUR CONSECUTIVE columns of one output row: 8-byte (bf16) / 16-byte
// (fp32) stores.  LDS strides: NP + 2 for M' and 80 for x make both fragment reads bank-conflict-free.
// HBM traffic is one read of x and M, one write of out; B * H / 64 workgroups (768 at the 64-object stress
// configuration, 384 in training).
typedef __attribute__((ext_vector_type(4))) float float4_t;
constexpr int AGG_COLS = 64, AGG_LDX = 80;
template <typename T, int NP>
__global__ __launch_bounds__(NT) void aggregate_kernel(const float* __restrict__ Mx, const T* __restrict__ x, T* out, int N,
                                                       int H, int mode, float scale, const float* scale_ptr, float self_w,
                                                       int accumulate) {
    constexpr int LDM = NP + 2;
    __shared__ float Mp[NP * LDM];       // Mp[i][j] = M'[i][j]
    __shared__ float xs[NP * AGG_LDX];   // xs[j][c]
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int cb = blockIdx.x * AGG_COLS;
    const float* Mb = Mx + (int64_t)b * N * N;
    for (int e = tid; e < NP * NP; e += NT) {
        const int i = e / NP, j = e % NP;
        float v = 0.f;
        if (i < N && j < N) {
            if (mode == XGGM_AGG_PLAIN) v = Mb[i * N + j];
            else if (mode == XGGM_AGG_TRANSPOSE) v = Mb[j * N + i];
            else v = Mb[i * N + j] + Mb[j * N + i];
        }
        Mp[i * LDM + j] = v;
    }
    const T* xb = x + (int64_t)b * N * H + cb;
    for (int e = tid; e < NP * (AGG_COLS / 4); e += NT) {  // 4 columns per thread
        const int j = e / (AGG_COLS / 4), c = (e % (AGG_COLS / 4)) * 4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (j < N && cb + c < H) load4(xb + (int64_t)j * H + c, v);  // H % 4 == 0
        *reinterpret_cast<float4*>(xs + j * AGG_LDX + c) = make_float4(v[0], v[1], v[2], v[3]);
    }
    __syncthreads();
    if (scale_ptr) scale *= (1.0f + *scale_ptr);  // GIN: (1 + eps)
    const int fr = lane & 15, fq = lane >> 4, c0 = wid * 16;
    float4_t acc[NP / 16];
#pragma unroll
    for (int t = 0; t < NP / 16; ++t) acc[t] = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int j0 = 0; j0 < NP; j0 += 4) {
        const float xa = xs[(j0 + fq) * AGG_LDX + c0 + fr];  // A operand: x^T, row = column c0 + fr, k = j0 + fq
#pragma unroll
        for (int t = 0; t < NP / 16; ++t)                     // B operand: M'^T, k = j0 + fq, column = row t*16 + fr
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa, Mp[(t * 16 + fr) * LDM + j0 + fq], acc[t], 0, 0, 0);
    }
    // lane holds out[i = t*16 + fr][c = c0 + 4 fq .. + 3]
    const int c = cb + c0 + 4 * fq;
    if (c >= H) return;
    T* ob = out + (int64_t)b * N * H + c;
#pragma unroll
    for (int t = 0; t < NP / 16; ++t) {
        const int i = t * 16 + fr;
        if (i < N) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = scale * acc[t][r];
            if (self_w != 0.f) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += self_w * xs[i * AGG_LDX + c0 + 4 * fq + r];
            }
            if (accumulate) {
                float o[4];
                load4(ob + (int64_t)i * H, o);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += o[r];
            }
            store4(ob + (int64_t)i * H, v);
        }
    }
}

// ---- bf16 storage: the same product on the bf16 matrix cores -------------------------------------------------
// With bf16 node features the fp32 MFMA above spends 16x the matrix-core time the data deserve (v_mfma_f32_16x16x4
// runs at 1/16 of the bf16 rate; at N = 64 it weighs as much as the HBM traffic).  Here M' is split into two bf16
// terms, M' = hi + lo with hi = bf16(M'), lo = bf16(M' - hi): out = hi x + lo x accumulated in fp32 carries M' to 16
// mantissa bits (relative 2^-17), far below the bf16 rounding of x and of the result, on v_mfma_f32_16x16x32_bf16.
//   x slab [N, 64] bf16 by 16-byte loads -> LDS [k = j][n = c] (row stride 72), read transposed (ds_read_b64_tr_b16)
//   as the A operand x^T; M' staged raw as fp32 (coalesced), then hi / lo [i][k = j] (row stride 72), read by rows
//   (ds_read_b128) as the B operand M'^T.  D^T = x^T M'^T: a lane holds four consecutive columns of one output row.
// One workgroup = one sample x 64 columns; the column blocks of a sample run on ONE XCD (blocks are dealt round-robin
// over the 8 XCDs, so sample b gets the ids with id % 8 == b % 8): its adjacency is fetched into one L2, not eight.
typedef __attribute__((ext_vector_type(8))) short agg_short8;
typedef __attribute__((ext_vector_type(4))) short agg_short4;
typedef __attribute__((ext_vector_type(8))) __bf16 agg_bf16x8;
template <int NI>
__global__ __launch_bounds__(NT) void aggregate_bf16_kernel(const float* __restrict__ Mx, const bf16* __restrict__ x, bf16* out,
                                                            int B, int N, int H, int mode, float scale, const float* scale_ptr,
                                                            float self_w, int accumulate) {
    constexpr int NK = 64, LDX = 72, LDM = 72, LDR = 65;
    __shared__ float Mr[64 * LDR];
    __shared__ __attribute__((aligned(16))) bf16 xs[NK * LDX];
    __shared__ __attribute__((aligned(16))) bf16 Mh[NI * 16 * LDM];
    __shared__ __attribute__((aligned(16))) bf16 Ml[NI * 16 * LDM];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int ncb = (H + AGG_COLS - 1) / AGG_COLS;
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int b = (q / ncb) * 8 + xcd, cb = (q % ncb) * AGG_COLS;
    if (b >= B) return;  // (whole workgroup: the padding blocks of a batch that is not a multiple of 8)
    const float* Mb = Mx + (int64_t)b * N * N;
    if ((N & 3) == 0 && (reinterpret_cast<uintptr_t>(Mb) & 15) == 0) {  // 16-byte loads (N = 36, 64)
        for (int e = tid; e < N * N / 4; e += NT) {
            const float4 v = *reinterpret_cast<const float4*>(Mb + 4 * e);
            float* d = Mr + ((4 * e) / N) * LDR + (4 * e) % N;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
    } else {
        for (int e = tid; e < N * N; e += NT) Mr[(e / N) * LDR + e % N] = Mb[e];
    }
    const bf16* xb = x + (int64_t)b * N * H + cb;
    for (int e = tid; e < NK * 8; e += NT) {  // 8 chunks of 8 columns per k-row
        const int j = e >> 3, c = (e & 7) * 8;
        agg_short8 v = {};
        if (j < N && cb + c + 8 <= H) v = *reinterpret_cast<const agg_short8*>(xb + (int64_t)j * H + c);  // H % 8 == 0 here
        *reinterpret_cast<agg_short8*>(xs + j * LDX + c) = v;
    }
    __syncthreads();
    for (int e = tid; e < NI * 16 * NK / 4; e += NT) {  // four consecutive k per thread: 8-byte LDS stores
        const int i = (4 * e) / NK, j0 = (4 * e) % NK;
        agg_short4 h4, l4;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + u;
            float v = 0.f;
            if (i < N && j < N) {
                if (mode == XGGM_AGG_PLAIN) v = Mr[i * LDR + j];
                else if (mode == XGGM_AGG_TRANSPOSE) v = Mr[j * LDR + i];
                else v = Mr[i * LDR + j] + Mr[j * LDR + i];
            }
            const bf16 hi = __float2bfloat16(v);
            h4[u] = __builtin_bit_cast(short, hi);
            l4[u] = __builtin_bit_cast(short, __float2bfloat16(v - __bfloat162float(hi)));
        }
        *reinterpret_cast<agg_short4*>(Mh + i * LDM + j0) = h4;
        *reinterpret_cast<agg_short4*>(Ml + i * LDM + j0) = l4;
    }
    __syncthreads();
    if (scale_ptr) scale *= (1.0f + *scale_ptr);  // GIN: (1 + eps)
    const int fr = lane & 15, fq = lane >> 4, c0 = wid * 16;
    float4_t acc[NI];
#pragma unroll
    for (int t = 0; t < NI; ++t) acc[t] = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < NK; ks += 32) {
        // A operand x^T: lane (fr, fq) gets column c0 + fr, k = ks + 8 fq .. + 7 through two transposing reads
        const int qq = fr >> 2, pp = fr & 3;
        const bf16* a0 = xs + (ks + 8 * fq + qq) * LDX + c0 + 4 * pp;
        const agg_short4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) agg_short4*)(a0));
        const agg_short4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) agg_short4*)(a0 + 4 * LDX));
        agg_short8 av;
        av[0] = lo4[0]; av[1] = lo4[1]; av[2] = lo4[2]; av[3] = lo4[3];
        av[4] = hi4[0]; av[5] = hi4[1]; av[6] = hi4[2]; av[7] = hi4[3];
        const agg_bf16x8 a = __builtin_bit_cast(agg_bf16x8, av);
#pragma unroll
        for (int t = 0; t < NI; ++t) {
            const agg_bf16x8 bh = __builtin_bit_cast(agg_bf16x8, *reinterpret_cast<const agg_short8*>(Mh + (t * 16 + fr) * LDM + ks + fq * 8));
            const agg_bf16x8 bl = __builtin_bit_cast(agg_bf16x8, *reinterpret_cast<const agg_short8*>(Ml + (t * 16 + fr) * LDM + ks + fq * 8));
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bh, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bl, acc[t], 0, 0, 0);
        }
    }
    // lane holds out[i = t*16 + fr][c = c0 + 4 fq .. + 3]
    const int c = cb + c0 + 4 * fq;
    if (c >= H) return;
    bf16* ob = out + (int64_t)b * N * H + c;
#pragma unroll
    for (int t = 0; t < NI; ++t) {
        const int i = t * 16 + fr;
        if (i < N) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = scale * acc[t][r];
            if (self_w != 0.f) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += self_w * __bfloat162float(xs[i * LDX + c0 + 4 * fq + r]);
            }
            if (accumulate) {
                float o[4];
                load4(ob + (int64_t)i * H, o);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += o[r];
            }
            store4(ob + (int64_t)i * H, v);
        }
    }
}

// ---- GCNConv's tail in one launch: out = LayerNorm(res + M y) ------------------------------------------------------
// GCNConv (src/module/gcn.py:22-29) is LN(x + W (A x)) = LN(x + A (x W^T)): with the product y = x W^T taken FIRST (a
// plain GEMM over all B * N rows), what is left per sample is M y + x and a LayerNorm over whole rows -- one kernel
// instead of aggregate, product epilogue and LayerNorm (three launches, two round trips of a [B, N, H] tensor).
// One workgroup = one sample x 16 output rows x ALL H columns, so the row statistics never leave the workgroup:
//   y slab [N, H] bf16 -> LDS [k = j][c] (row stride H + 8), read transposed as the A operand y^T exactly as in
//   aggregate_bf16_kernel; the 16 adjacency rows split into hi + lo bf16 (16 mantissa bits) as the B operand;
//   wave w owns the column groups 16 (w CG .. w CG + CG - 1), CG = H / 64: a lane ends with row fr, four consecutive
//   columns of each of its CG groups; + res, rounded to the storage type (the saved z, as ln_fwd_kernel does), mean and
//   centred variance by two shuffles across the lanes of a row and a fixed-order sum over the four waves through LDS.
// The row tiles of a sample run on ONE XCD (its y slab is fetched into one L2), as the column blocks of the aggregate.
template <int CG>
__global__ __launch_bounds__(NT) void agg_res_ln_kernel(const float* __restrict__ Mx, const bf16* __restrict__ y,
                                                        const bf16* __restrict__ res, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, bf16* out, bf16* z_out, float* stats,
                                                        int B, int N, float eps) {
    constexpr int H = CG * 64, NK = 64, LDY = H + 8, LDM = 72;
    extern __shared__ __attribute__((aligned(16))) unsigned char agg_lds[];
    bf16* ys = reinterpret_cast<bf16*>(agg_lds);                    // [NK][LDY]
    bf16* Mh = ys + NK * LDY;                                       // [16][LDM]
    bf16* Ml = Mh + 16 * LDM;                                       // [16][LDM]
    float* red = reinterpret_cast<float*>(Ml + 16 * LDM);           // [4][16]
    float* gb = red + 64;                                           // gamma [H], beta [H]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int ntile = (N + 15) >> 4;
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int b = (q / ntile) * 8 + xcd, i0 = (q % ntile) * 16;
    if (b >= B) return;  // whole workgroup (padding of a batch that is not a multiple of 8)
    const int fr = lane & 15, fq = lane >> 4;
    const int row = i0 + fr;
    const bool row_ok = row < N;
    // the residual rows go straight to registers: issued first, used last
    agg_short4 rres[CG];
    const bf16* rb = res + ((int64_t)b * N + (row_ok ? row : 0)) * H + wid * CG * 16 + 4 * fq;
#pragma unroll
    for (int g = 0; g < CG; ++g) rres[g] = *reinterpret_cast<const agg_short4*>(rb + g * 16);
    // gamma / beta through LDS: read here, beside the slab, instead of twelve dependent global loads per lane behind the
    // row statistics (the kernel's tail was those round trips)
    for (int e = tid; e < H / 4; e += NT) {
        *reinterpret_cast<float4*>(gb + 4 * e) = *reinterpret_cast<const float4*>(gamma + 4 * e);
        *reinterpret_cast<float4*>(gb + H + 4 * e) = *reinterpret_cast<const float4*>(beta + 4 * e);
    }
    // the 16 adjacency rows of this tile -> hi / lo bf16, k beyond N zero
    const float* Mb = Mx + (int64_t)b * N * N;
    for (int e = tid; e < 16 * NK / 4; e += NT) {
        const int i = (4 * e) / NK, j0 = (4 * e) % NK;
        agg_short4 h4, l4;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + u;
            const float v = (i0 + i < N && j < N) ? Mb[(int64_t)(i0 + i) * N + j] : 0.f;
            const bf16 hi = __float2bfloat16(v);
            h4[u] = __builtin_bit_cast(short, hi);
            l4[u] = __builtin_bit_cast(short, __float2bfloat16(v - __bfloat162float(hi)));
        }
        *reinterpret_cast<agg_short4*>(Mh + i * LDM + j0) = h4;
        *reinterpret_cast<agg_short4*>(Ml + i * LDM + j0) = l4;
    }
    // y slab of the sample: 16-byte chunks, rows beyond N zero (they meet zero adjacency entries, but not as NaNs).
    // ALL loads of a thread are issued before the first LDS store (rows beyond N read row N - 1 and are zeroed on the
    // way): the first form, a load -> store loop behind a row test, waited for every load on its own -- 24 serial round
    // trips per thread, 17.8 us per launch at B = 64, N = 64 where aggregate + LayerNorm together had taken 17.5.
    const bf16* yb = y + (int64_t)b * N * H;
    constexpr int CH = H / 8, ITER = NK * CH / NT;
    static_assert(NK * CH % NT == 0, "the slab's chunks divide over the workgroup");
    {
        agg_short8 v[ITER];
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int e = tid + it * NT, j = e / CH, c = (e % CH) * 8;
            v[it] = *reinterpret_cast<const agg_short8*>(yb + (int64_t)min(j, N - 1) * H + c);
        }
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int e = tid + it * NT, j = e / CH, c = (e % CH) * 8;
            const agg_short8 zero = {};
            *reinterpret_cast<agg_short8*>(ys + j * LDY + c) = j < N ? v[it] : zero;
        }
    }
    __syncthreads();
    float4_t acc[CG];
#pragma unroll
    for (int g = 0; g < CG; ++g) acc[g] = (float4_t){0.f, 0.f, 0.f, 0.f};
    const int qq = fr >> 2, pp = fr & 3;
#pragma unroll
    for (int ks = 0; ks < NK; ks += 32) {
        const agg_bf16x8 bh = __builtin_bit_cast(agg_bf16x8, *reinterpret_cast<const agg_short8*>(Mh + fr * LDM + ks + fq * 8));
        const agg_bf16x8 bl = __builtin_bit_cast(agg_bf16x8, *reinterpret_cast<const agg_short8*>(Ml + fr * LDM + ks + fq * 8));
#pragma unroll
        for (int g = 0; g < CG; ++g) {
            const bf16* a0 = ys + (ks + 8 * fq + qq) * LDY + (wid * CG + g) * 16 + 4 * pp;
            const agg_short4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) agg_short4*)(a0));
            const agg_short4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) agg_short4*)(a0 + 4 * LDY));
            agg_short8 av;
            av[0] = lo4[0]; av[1] = lo4[1]; av[2] = lo4[2]; av[3] = lo4[3];
            av[4] = hi4[0]; av[5] = hi4[1]; av[6] = hi4[2]; av[7] = hi4[3];
            const agg_bf16x8 a = __builtin_bit_cast(agg_bf16x8, av);
            acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bh, acc[g], 0, 0, 0);
            acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bl, acc[g], 0, 0, 0);
        }
    }
    // z = res + M y, rounded to the storage type; lane: row fr, columns (wid CG + g) 16 + 4 fq + r
    float sum = 0.f;
#pragma unroll
    for (int g = 0; g < CG; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float v = acc[g][r] + __bfloat162float(__builtin_bit_cast(bf16, (short)rres[g][r]));
            acc[g][r] = __bfloat162float(__float2bfloat16(v));
            sum += acc[g][r];
        }
    auto row_total = [&](float v) {  // over the four lanes of the row, then the four waves in index order
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        __syncthreads();  // (the previous round's reads of red are done)
        if (fq == 0) red[wid * 16 + fr] = v;
        __syncthreads();
        return ((red[fr] + red[16 + fr]) + red[32 + fr]) + red[48 + fr];
    };
    const float mean = row_total(sum) / (float)H;
    float var = 0.f;
#pragma unroll
    for (int g = 0; g < CG; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float t = acc[g][r] - mean;
            var += t * t;
        }
    const float rstd = rsqrtf(row_total(var) / (float)H + eps);
    if (!row_ok) return;
    if (stats && wid == 0 && fq == 0) {
        stats[2 * ((int64_t)b * N + row)] = mean;
        stats[2 * ((int64_t)b * N + row) + 1] = rstd;
    }
    const int64_t ob = ((int64_t)b * N + row) * H + wid * CG * 16 + 4 * fq;
#pragma unroll
    for (int g = 0; g < CG; ++g) {
        const int c = wid * CG * 16 + g * 16 + 4 * fq;
        float o[4], zz[4];
        const float4 g4v = *reinterpret_cast<const float4*>(gb + c), be4v = *reinterpret_cast<const float4*>(gb + H + c);
        const float g4[4] = {g4v.x, g4v.y, g4v.z, g4v.w}, be4[4] = {be4v.x, be4v.y, be4v.z, be4v.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            zz[r] = acc[g][r];
            o[r] = (acc[g][r] - mean) * rstd * g4[r] + be4[r];
        }
        if (z_out) store4(z_out + ob + g * 16, zz);
        store4(out + ob + g * 16, o);
    }
}

// d eps of GIN: sum_{b,i,c} dh[b,i,c] * (A @ x)[b,i,c]  -> one scalar
template <typename T, int NP>
__global__ __launch_bounds__(NT) void agg_dot_kernel(const float* __restrict__ Mx, const T* __restrict__ x,
                                                     const T* __restrict__ dh, float* out, int N, int H, float* ws) {
    __shared__ __attribute__((aligned(16))) float Mt[NP * NP];
    __shared__ float red[NT / 64];
    const int b = blockIdx.y, tid = threadIdx.x;
    const float* Mb = Mx + (int64_t)b * N * N;
    for (int e = tid; e < NP * NP; e += NT) {
        const int j = e / NP, i = e % NP;
        Mt[e] = (i < N && j < N) ? Mb[i * N + j] : 0.f;
    }
    __syncthreads();
    const int c = blockIdx.x * NT + tid;
    float total = 0.f;
    if (c < H) {
        const T* xb = x + (int64_t)b * N * H + c;
        const T* db = dh + (int64_t)b * N * H + c;
        float acc[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) acc[i] = 0.f;
        for (int j = 0; j < N; ++j) {
            const float xj = to_f32(xb[(int64_t)j * H]);
#pragma unroll
            for (int i = 0; i < NP; ++i) acc[i] += Mt[j * NP + i] * xj;
        }
#pragma unroll
        for (int i = 0; i < NP; ++i)
            if (i < N) total += acc[i] * to_f32(db[(int64_t)i * H]);
    }
    total = wave_sum(total);
    if ((tid & 63) == 0) red[tid >> 6] = total;
    __syncthreads();
    // the workgroups' sums are added in index order by the one that finishes last (ordered_grid_sum): no float atomics
    float sum;
    if (ordered_grid_sum((red[0] + red[1]) + (red[2] + red[3]), ws, gridDim.x * gridDim.y, blockIdx.y * gridDim.x + blockIdx.x, sum))
        *out += sum;
}

// ------------------------------------------------------------------------------- adjacency regeneration
// one workgroup per sample; wave w owns columns w, w+4, ...
__global__ __launch_bounds__(NT) void adj_regen_fwd_kernel(const float* __restrict__ S, float* adj, float* colmax,
                                                           int32_t* argmax, int N) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ss = smem;       // [N][N+1]
    float* mx = Ss + N * (N + 1);  // [N]
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const float* Sb = S + (int64_t)b * N * N;
    for (int e = tid; e < N * N; e += NT) Ss[(e / N) * (N + 1) + e % N] = Sb[e];
    __syncthreads();
    for (int i = wid; i < N; i += NT / 64) {
        // max over rows r of S[r][i]; torch.max(dim=1) returns the FIRST arg-max on ties
        const float v = lane < N ? Ss[lane * (N + 1) + i] : -INFINITY;
        const float m = wave_max(v);
        const unsigned long long hit = __ballot(lane < N && v == m);
        if (lane == 0) {
            const int first = __ffsll((long long)hit) - 1;
            mx[i] = m;
            colmax[(int64_t)b * N + i] = m;
            argmax[(int64_t)b * N + i] = first;
        }
    }
    __syncthreads();
    for (int e = tid; e < N * N; e += NT) {
        const int i = e / N, j = e % N;
        adj[(int64_t)b * N * N + e] = (i == j) ? 0.f : sigmoid_f(Ss[i * (N + 1) + j] / mx[i]);
    }
}

// dS from d_adj:  R = S[i][j]/m_i, A = sigmoid(R) off-diagonal;  dR = dA * A (1 - A);
// dS[i][j] = dR/m_i;  dm_i = -sum_j dR[i][j] S[i][j] / m_i^2;  dS[argmax_i][i] += dm_i
__global__ __launch_bounds__(NT) void adj_regen_bwd_kernel(const float* __restrict__ d_adj, const float* __restrict__ S,
                                                           const float* __restrict__ adj, const float* __restrict__ colmax,
                                                           const int32_t* __restrict__ argmax, float* dS, int N) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* D = smem;  // [N][N+1]
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int64_t ob = (int64_t)b * N * N;
    float* dm = D + N * (N + 1);
    for (int i = wid; i < N; i += NT / 64) {
        const float m = colmax[(int64_t)b * N + i];
        float part = 0.f;
        if (lane < N) {
            const int e = i * N + lane;
            const float a = adj[ob + e];
            const float dr = (lane == i) ? 0.f : d_adj[ob + e] * a * (1.f - a);
            D[i * (N + 1) + lane] = dr / m;
            part = dr * S[ob + e];
        }
        part = wave_sum(part);
        if (lane == 0) dm[i] = -part / (m * m);
    }
    __syncthreads();
    if (tid < N) {
        const int r = argmax[(int64_t)b * N + tid];
        D[r * (N + 1) + tid] += dm[tid];  // column tid: unique target per thread
    }
    __syncthreads();
    for (int e = tid; e < N * N; e += NT) dS[ob + e] = D[(e / N) * (N + 1) + e % N];
}

// ------------------------------------------------------------------------------- adjacency init + edge noise
// k-th strict-upper-triangle entry in row-major order: (0,1),(0,2)...(0,N-1),(1,2)...
__host__ __device__ inline int triu_k(int i, int j, int N) { return i * N - (i * (i + 1)) / 2 + (j - i - 1); }

__global__ __launch_bounds__(NT) void adj_init_fwd_kernel(const float* __restrict__ e, const float* __restrict__ randn,
                                                          float* adj, float* gradlog, int B, int N, int NE, float sigma,
                                                          const uint64_t* rng, uint32_t sid) {
    const int64_t total = (int64_t)B * N * N;
    uint64_t seed = 0, off = 0;
    if (!randn && rng) {
        seed = rng[0];
        off = rng[1];
    }
    for (int64_t t = (int64_t)blockIdx.x * NT + threadIdx.x; t < total; t += (int64_t)gridDim.x * NT) {
        const int b = (int)(t / (N * N)), r = (int)(t % (N * N)), i = r / N, j = r % N;
        float a = 0.f, n = 0.f;
        if (i != j) {
            const int lo = min(i, j), hi = max(i, j);
            if (e) a = e[(int64_t)b * NE + triu_k(lo, hi, N)];
            const int64_t uidx = ((int64_t)b * N + lo) * N + hi;  // the upper-triangle draw is mirrored
            const float z = randn ? randn[uidx] : (rng ? philox_normal(seed, off, sid, (uint64_t)uidx) : 0.f);
            n = z * sigma;
        }
        adj[t] = a + n;
        if (gradlog) gradlog[t] = -n / (sigma * sigma);
    }
}

__global__ __launch_bounds__(NT) void adj_init_bwd_kernel(const float* __restrict__ d_adj, float* d_e, int B, int N, int NE) {
    const int64_t total = (int64_t)B * NE;
    for (int64_t t = (int64_t)blockIdx.x * NT + threadIdx.x; t < total; t += (int64_t)gridDim.x * NT) {
        const int b = (int)(t / NE), k = (int)(t % NE);
        // invert k -> (i, j): walk rows (N <= 64, trivial)
        int i = 0, rem = k;
        while (rem >= N - 1 - i) {
            rem -= N - 1 - i;
            ++i;
        }
        const int j = i + 1 + rem;
        const float* g = d_adj + (int64_t)b * N * N;
        d_e[t] = g[i * N + j] + g[j * N + i];
    }
}

// ------------------------------------------------------------------------------- feature noise
template <typename T>
__global__ __launch_bounds__(NT) void feature_noise_kernel(const T* __restrict__ x, const float* __restrict__ randn, T* out,
                                                           float* gradlog, int64_t n, float sigma, const uint64_t* rng,
                                                           uint32_t sid) {
    uint64_t seed = 0, off = 0;
    if (!randn && rng) {
        seed = rng[0];
        off = rng[1];
    }
    for (int64_t t = (int64_t)blockIdx.x * NT + threadIdx.x; t < n; t += (int64_t)gridDim.x * NT) {
        const float z = randn ? randn[t] : philox_normal(seed, off, sid, (uint64_t)t);
        const float nz = z * sigma;
        out[t] = from_f32<T>(to_f32(x[t]) + nz);
        gradlog[t] = -nz / (sigma * sigma);
    }
}

// ------------------------------------------------------------------------------- pool + concat
// out[b, 0:H] = x[b,:];  out[b, H:2H] = tanh(mean_i nodes[b,i,:])
template <typename T>
__global__ __launch_bounds__(NT) void pool_concat_fwd_kernel(const T* __restrict__ x, const T* __restrict__ nodes, T* out, int B,
                                                             int N, int H) {
    const int b = blockIdx.y, c = blockIdx.x * NT + threadIdx.x;
    if (c >= H) return;
    float s = 0.f;
    for (int i = 0; i < N; ++i) s += to_f32(nodes[((int64_t)b * N + i) * H + c]);
    out[(int64_t)b * 2 * H + c] = x[(int64_t)b * H + c];
    out[(int64_t)b * 2 * H + H + c] = from_f32<T>(tanhf(s / (float)N));
}
template <typename T>
__global__ __launch_bounds__(NT) void pool_concat_bwd_kernel(const T* __restrict__ d_out, const T* __restrict__ out, T* dx,
                                                             T* dnodes, int B, int N, int H, int accumulate_dx) {
    const int b = blockIdx.y, c = blockIdx.x * NT + threadIdx.x;
    if (c >= H) return;
    const float g0 = to_f32(d_out[(int64_t)b * 2 * H + c]);
    const float t = to_f32(out[(int64_t)b * 2 * H + H + c]);
    const float g1 = to_f32(d_out[(int64_t)b * 2 * H + H + c]) * (1.f - t * t) / (float)N;
    T* px = dx + (int64_t)b * H + c;
    *px = from_f32<T>(accumulate_dx ? to_f32(*px) + g0 : g0);
    const T gv = from_f32<T>(g1);
    for (int i = 0; i < N; ++i) dnodes[((int64_t)b * N + i) * H + c] = gv;
}

// broadcast one row per sample to N node rows (x.unsqueeze(1).repeat(1, N, 1)) and its
// backward (sum over the N rows).  node_fc is applied to the B pooled rows only and its output
// broadcast: exactly the values the reference computes on 36 identical rows (vqacpv2.py:228-229)
template <typename T>
__global__ __launch_bounds__(NT) void bcast_rows_kernel(const T* __restrict__ x, T* out, int B, int N, int H) {
    const int b = blockIdx.y, c = blockIdx.x * NT + threadIdx.x;
    if (c >= H) return;
    const T v = x[(int64_t)b * H + c];
    for (int i = 0; i < N; ++i) out[((int64_t)b * N + i) * H + c] = v;
}
template <typename T>
__global__ __launch_bounds__(NT) void sum_rows_kernel(const T* __restrict__ g, T* out, int B, int N, int H) {
    const int b = blockIdx.y, c = blockIdx.x * NT + threadIdx.x;
    if (c >= H) return;
    float s = 0.f;
    for (int i = 0; i < N; ++i) s += to_f32(g[((int64_t)b * N + i) * H + c]);
    out[(int64_t)b * H + c] = from_f32<T>(s);
}

inline int grid1d(int64_t n) { return (int)std::min<int64_t>(ceil_div64(n, NT), 2048); }

template <typename T>
int aggregate(const float* M, const void* x, void* out, int B, int N, int H, int mode, float scale, const float* scale_ptr,
              float self_w, int accumulate, hipStream_t st) {
    XGGM_REQUIRE(M && x && out && B > 0 && N > 0 && H > 0, "xggm_aggregate: bad arguments");
    XGGM_REQUIRE(N <= 64, "xggm_aggregate: N=%d exceeds the 64x64 LDS adjacency tile", N);
    XGGM_REQUIRE(mode >= 0 && mode <= XGGM_AGG_SYMMETRIZE, "xggm_aggregate: bad mode %d", mode);
    XGGM_REQUIRE(B <= 65535, "xggm_aggregate: batch too large");
    XGGM_REQUIRE(H % 4 == 0, "xggm_aggregate: H=%d must be a multiple of 4", H);
    if constexpr (sizeof(T) == 2) {
        // bf16 storage, 8-aligned rows: bf16 matrix cores with the adjacency split into hi + lo (aggregate_bf16_kernel)
        if (H % 8 == 0 && reinterpret_cast<uintptr_t>(x) % 16 == 0 && reinterpret_cast<uintptr_t>(out) % 8 == 0) {
            const dim3 g8(ceil_div(B, 8) * 8 * ceil_div(H, AGG_COLS));
            if (N <= 48)
                hipLaunchKernelGGL((aggregate_bf16_kernel<3>), g8, dim3(NT), 0, st, M, (const bf16*)x, (bf16*)out, B, N, H, mode,
                                   scale, scale_ptr, self_w, accumulate);
            else
                hipLaunchKernelGGL((aggregate_bf16_kernel<4>), g8, dim3(NT), 0, st, M, (const bf16*)x, (bf16*)out, B, N, H, mode,
                                   scale, scale_ptr, self_w, accumulate);
            return xggm_check_launch("xggm_aggregate");
        }
    }
    dim3 grid(ceil_div(H, AGG_COLS), B);
    if (N <= 48)
        hipLaunchKernelGGL((aggregate_kernel<T, 48>), grid, dim3(NT), 0, st, M, (const T*)x, (T*)out, N, H, mode, scale,
                           scale_ptr, self_w, accumulate);
    else
        hipLaunchKernelGGL((aggregate_kernel<T, 64>), grid, dim3(NT), 0, st, M, (const T*)x, (T*)out, N, H, mode, scale,
                           scale_ptr, self_w, accumulate);
    return xggm_check_launch("xggm_aggregate");
}

template <int CG>
int launch_agg_res_ln(const float* M, const bf16* y, const bf16* res, const float* gamma, const float* beta, bf16* out, bf16* z_out,
                      float* stats, int B, int N, float eps, hipStream_t st) {
    constexpr int H = CG * 64;
    constexpr size_t lds = sizeof(bf16) * (64 * (H + 8) + 2 * 16 * 72) + sizeof(float) * (64 + 2 * H);
    static bool once = [] {  // more than the default 64 KB of dynamic LDS: opt in once per instantiation
        return hipFuncSetAttribute(reinterpret_cast<const void*>(agg_res_ln_kernel<CG>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds) == hipSuccess;
    }();
    XGGM_REQUIRE(once, "xggm_agg_residual_ln_bf16: cannot reserve %zu bytes of LDS", lds);
    const int grid = ceil_div(B, 8) * 8 * ceil_div(N, 16);
    hipLaunchKernelGGL((agg_res_ln_kernel<CG>), dim3(grid), dim3(NT), lds, st, M, y, res, gamma, beta, out, z_out, stats, B, N, eps);
    return xggm_check_launch("xggm_agg_residual_ln_bf16");
}

template <typename T>
int agg_dot(const float* M, const void* x, const void* dh, float* out, int B, int N, int H, float* ws, hipStream_t st) {
    XGGM_REQUIRE(M && x && dh && out && ws && B > 0 && N > 0 && N <= 64 && H > 0 && B <= 65535, "xggm_agg_dot: bad arguments");
    dim3 grid(ceil_div(H, NT), B);
    XGGM_REQUIRE((int64_t)grid.x * grid.y <= XGGM_SUM_WS_FLOATS - 8, "xggm_agg_dot: %lld workgroups exceed the sum workspace",
                 (long long)grid.x * grid.y);
    if (N <= 40)
        hipLaunchKernelGGL((agg_dot_kernel<T, 40>), grid, dim3(NT), 0, st, M, (const T*)x, (const T*)dh, out, N, H, ws);
    else
        hipLaunchKernelGGL((agg_dot_kernel<T, 64>), grid, dim3(NT), 0, st, M, (const T*)x, (const T*)dh, out, N, H, ws);
    return xggm_check_launch("xggm_agg_dot");
}

}  // namespace

extern "C" int xggm_agg_residual_ln_bf16(const float* M, const void* y, const void* res, const float* gamma, const float* beta,
                                        void* out, void* z_out, float* stats, int B, int N, int H, float eps, hipStream_t st) {
    XGGM_REQUIRE(M && y && res && gamma && beta && out && B > 0 && N > 0, "xggm_agg_residual_ln_bf16: bad arguments");
    XGGM_REQUIRE(N <= 64, "xggm_agg_residual_ln_bf16: N=%d exceeds the 64-row adjacency tile", N);
    XGGM_REQUIRE(H == 768 || H == 256 || H == 128 || H == 64, "xggm_agg_residual_ln_bf16: H=%d (built for 64, 128, 256, 768)", H);
    XGGM_REQUIRE((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(res) | reinterpret_cast<uintptr_t>(out) |
                  reinterpret_cast<uintptr_t>(z_out) | reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta)) % 16 == 0,
                 "xggm_agg_residual_ln_bf16: pointers must be 16-byte aligned");
    XGGM_REQUIRE(out != y && out != res && z_out != y && (z_out == nullptr || z_out != out),
                 "xggm_agg_residual_ln_bf16: out / z_out may not alias y (read by every row tile of the sample), each other or res");
    const bf16 *yy = (const bf16*)y, *rr = (const bf16*)res;
    bf16 *oo = (bf16*)out, *zz = (bf16*)z_out;
    if (H == 768) return launch_agg_res_ln<12>(M, yy, rr, gamma, beta, oo, zz, stats, B, N, eps, st);
    if (H == 256) return launch_agg_res_ln<4>(M, yy, rr, gamma, beta, oo, zz, stats, B, N, eps, st);
    if (H == 128) return launch_agg_res_ln<2>(M, yy, rr, gamma, beta, oo, zz, stats, B, N, eps, st);
    return launch_agg_res_ln<1>(M, yy, rr, gamma, beta, oo, zz, stats, B, N, eps, st);
}

extern "C" int xggm_adj_regen_fwd(const float* S, float* adj, float* colmax, int32_t* argmax, int B, int N, hipStream_t st) {
    XGGM_REQUIRE(S && adj && colmax && argmax && B > 0 && N > 0, "xggm_adj_regen_fwd: bad arguments");
    XGGM_REQUIRE(N <= 64, "xggm_adj_regen_fwd: N=%d > 64", N);
    const size_t lds = sizeof(float) * (N * (N + 1) + N);
    hipLaunchKernelGGL(adj_regen_fwd_kernel, dim3(B), dim3(NT), lds, st, S, adj, colmax, argmax, N);
    return xggm_check_launch("xggm_adj_regen_fwd");
}

extern "C" int xggm_adj_regen_bwd(const float* d_adj, const float* S, const float* adj, const float* colmax,
                                  const int32_t* argmax, float* dS, int B, int N, hipStream_t st) {
    XGGM_REQUIRE(d_adj && S && adj && colmax && argmax && dS && B > 0 && N > 0, "xggm_adj_regen_bwd: bad arguments");
    XGGM_REQUIRE(N <= 64, "xggm_adj_regen_bwd: N=%d > 64", N);
    const size_t lds = sizeof(float) * (N * (N + 1) + N);
    hipLaunchKernelGGL(adj_regen_bwd_kernel, dim3(B), dim3(NT), lds, st, d_adj, S, adj, colmax, argmax, dS, N);
    return xggm_check_launch("xggm_adj_regen_bwd");
}

extern "C" int xggm_adj_init_fwd(const float* e, const float* randn, float* adj, float* gradlog, int B, int N, float sigma,
                                 const uint64_t* rng, uint32_t sid, hipStream_t st) {
    XGGM_REQUIRE(adj && B > 0 && N > 1 && N <= 64, "xggm_adj_init_fwd: bad arguments B=%d N=%d", B, N);
    XGGM_REQUIRE(sigma > 0.f, "xggm_adj_init_fwd: sigma must be > 0");
    hipLaunchKernelGGL(adj_init_fwd_kernel, dim3(grid1d((int64_t)B * N * N)), dim3(NT), 0, st, e, randn, adj, gradlog, B, N,
                       N * (N - 1) / 2, sigma, rng, sid);
    return xggm_check_launch("xggm_adj_init_fwd");
}

extern "C" int xggm_adj_init_bwd(const float* d_adj, float* d_e, int B, int N, hipStream_t st) {
    XGGM_REQUIRE(d_adj && d_e && B > 0 && N > 1 && N <= 64, "xggm_adj_init_bwd: bad arguments");
    const int NE = N * (N - 1) / 2;
    hipLaunchKernelGGL(adj_init_bwd_kernel, dim3(grid1d((int64_t)B * NE)), dim3(NT), 0, st, d_adj, d_e, B, N, NE);
    return xggm_check_launch("xggm_adj_init_bwd");
}

extern "C" int xggm_triu_index(int k, int N, int* i_out, int* j_out) {
    // host helper exposing the index map used by the kernels (bit-exactness tests)
    XGGM_REQUIRE(N > 1 && k >= 0 && k < N * (N - 1) / 2 && i_out && j_out, "xggm_triu_index: bad arguments");
    int i = 0, rem = k;
    while (rem >= N - 1 - i) {
        rem -= N - 1 - i;
        ++i;
    }
    *i_out = i;
    *j_out = i + 1 + rem;
    return (triu_k(*i_out, *j_out, N) == k) ? XGGM_OK : XGGM_ERR_ARG;
}

#define GRAPH_API(SUF, T)                                                                                                  \
    extern "C" int xggm_aggregate_##SUF(const float* M, const void* x, void* out, int B, int N, int H, int mode,          \
                                        float scale, const float* scale_ptr, float self_w, int accumulate,               \
                                        hipStream_t st) {                                                                 \
        return aggregate<T>(M, x, out, B, N, H, mode, scale, scale_ptr, self_w, accumulate, st);                          \
    }                                                                                                                      \
    extern "C" int xggm_agg_dot_##SUF(const float* M, const void* x, const void* dh, float* out, int B, int N, int H,     \
                                      float* ws, hipStream_t st) {                                                        \
        return agg_dot<T>(M, x, dh, out, B, N, H, ws, st);                                                                \
    }                                                                                                                      \
    extern "C" int xggm_feature_noise_##SUF(const void* x, const float* randn, void* out, float* gradlog, int64_t n,      \
                                            float sigma, const uint64_t* rng, uint32_t sid, hipStream_t st) {             \
        XGGM_REQUIRE(x && out && gradlog && n > 0 && sigma > 0.f && (randn || rng), "xggm_feature_noise: bad arguments"); \
        hipLaunchKernelGGL((feature_noise_kernel<T>), dim3(grid1d(n)), dim3(NT), 0, st, (const T*)x, randn, (T*)out,      \
                           gradlog, n, sigma, rng, sid);                                                                  \
        return xggm_check_launch("xggm_feature_noise");                                                                   \
    }                                                                                                                      \
    extern "C" int xggm_pool_concat_fwd_##SUF(const void* x, const void* nodes, void* out, int B, int N, int H,           \
                                              hipStream_t st) {                                                           \
        XGGM_REQUIRE(x && nodes && out && B > 0 && B <= 65535 && N > 0 && H > 0, "xggm_pool_concat_fwd: bad arguments");  \
        hipLaunchKernelGGL((pool_concat_fwd_kernel<T>), dim3(ceil_div(H, NT), B), dim3(NT), 0, st, (const T*)x,           \
                           (const T*)nodes, (T*)out, B, N, H);                                                            \
        return xggm_check_launch("xggm_pool_concat_fwd");                                                                 \
    }                                                                                                                      \
    extern "C" int xggm_pool_concat_bwd_##SUF(const void* d_out, const void* out, void* dx, void* dnodes, int B, int N,   \
                                              int H, int accumulate_dx, hipStream_t st) {                                 \
        XGGM_REQUIRE(d_out && out && dx && dnodes && B > 0 && B <= 65535 && N > 0 && H > 0,                               \
                     "xggm_pool_concat_bwd: bad arguments");                                                              \
        hipLaunchKernelGGL((pool_concat_bwd_kernel<T>), dim3(ceil_div(H, NT), B), dim3(NT), 0, st, (const T*)d_out,       \
                           (const T*)out, (T*)dx, (T*)dnodes, B, N, H, accumulate_dx);                                    \
        return xggm_check_launch("xggm_pool_concat_bwd");                                                                 \
    }                                                                                                                      \
    extern "C" int xggm_bcast_rows_##SUF(const void* x, void* out, int B, int N, int H, hipStream_t st) {                 \
        XGGM_REQUIRE(x && out && B > 0 && B <= 65535 && N > 0 && H > 0, "xggm_bcast_rows: bad arguments");                \
        hipLaunchKernelGGL((bcast_rows_kernel<T>), dim3(ceil_div(H, NT), B), dim3(NT), 0, st, (const T*)x, (T*)out, B, N, \
                           H);                                                                                            \
        return xggm_check_launch("xggm_bcast_rows");                                                                      \
    }                                                                                                                      \
    extern "C" int xggm_sum_rows_##SUF(const void* g, void* out, int B, int N, int H, hipStream_t st) {                   \
        XGGM_REQUIRE(g && out && B > 0 && B <= 65535 && N > 0 && H > 0, "xggm_sum_rows: bad arguments");                   \
        hipLaunchKernelGGL((sum_rows_kernel<T>), dim3(ceil_div(H, NT), B), dim3(NT), 0, st, (const T*)g, (T*)out, B, N,   \
                           H);                                                                                            \
        return xggm_check_launch("xggm_sum_rows");                                                                        \
    }

GRAPH_API(f32, float)
GRAPH_API(bf16, bf16)
