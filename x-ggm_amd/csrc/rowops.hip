// Row kernels: everything on the path that normalises a row of H features.
//   xggm_ln_fwd/bwd        out = [out +] s * drop_post( LN( drop_pre(in + bias) + residual ) )
//                          = BertAttOutput / BertOutput (src/lxrt/modeling.py:384-388, 441-445),
//                            GCNConv's LN (src/module/gcn.py:29), the "GeLU -> LN" tails of the heads and
//                            GNN read-outs incl. their dropout(.5) and jump-knowledge sum (gcn.py:70-77)
//   xggm_embed_fwd/bwd     BertEmbeddings (modeling.py:298-313): 3 gathers + LN + dropout
//   xggm_visn_embed_fwd/bwd VisualFeatEncoder tail (modeling.py:546-556): (LN(u+b) + LN(W_b box + b_b))/2, dropout
// One wave64 per row, 4 rows per 256-thread workgroup, 16-byte accesses, fp32 math, wave
// shuffle reductions; the row stays in registers between the statistics and the output pass,
// so HBM traffic is one read of each input and one write of each output.
#include "common.h"
#include "xggm.h"

namespace {

constexpr int NT = 256;
constexpr int WPB = 4;  // waves (rows in flight) per block

struct DropArgs {
    float p_pre, p_post;
    const uint64_t* rng;
    uint32_t s_pre, s_post;
};

__device__ __forceinline__ void rng_load(const uint64_t* rng, uint64_t& seed, uint64_t& off) {
    seed = rng ? rng[0] : 0;
    off = rng ? rng[1] : 0;
}


// ---- cross-row reductions without contended atomics --------------------------------------------
// Hundreds of waves adding into the SAME H addresses serialise in the memory-side atomic units
// (measured: 80 us for an LN backward that needs 10).  Instead every workgroup reduces its 4
// waves through LDS and stores one partial row per vector into a caller-provided workspace
// ws[block][K][H]; a second tiny kernel sums the blocks and adds into the fp32 gradients
// (one owner per address -> plain read-modify-write, deterministic order).
template <int NV, int W = WPB>
__device__ __forceinline__ void block_store_partial(const float (&p)[NV][4], float* lds, float* dst, int H, int lane,
                                                    int wid) {
    __syncthreads();
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int c = (v * 64 + lane) * 4;
        if (c < H) {
#pragma unroll
            for (int i = 0; i < 4; ++i) lds[wid * H + c + i] = p[v][i];
        }
    }
    __syncthreads();
    // the partial rows are read once, by the reduce launch at the end of the backward pass: keep them out of the caches
    for (int c = threadIdx.x; c < H; c += W * 64) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < W; ++w) t += lds[w * H + c];
        __builtin_nontemporal_store(t, dst + c);
    }
}

// the three partial rows of a LayerNorm backward workgroup (dgamma, dbeta, dbias) in ONE pass through LDS: 16-byte
// LDS writes, one barrier pair instead of three, 16-byte sums and non-temporal stores.  `lds` holds 3 * W * H floats.
// Summation order over the waves as in block_store_partial (bitwise the same partial rows).
template <int NV, int W>
__device__ __forceinline__ void block_store_partial3(const float (&p0)[NV][4], const float (&p1)[NV][4],
                                                     const float (&p2)[NV][4], float* lds, float* dst, int H, int lane,
                                                     int wid) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    __syncthreads();
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int c = (v * 64 + lane) * 4;
        if (c < H) {
            *reinterpret_cast<f4*>(lds + (0 * W + wid) * H + c) = (f4){p0[v][0], p0[v][1], p0[v][2], p0[v][3]};
            *reinterpret_cast<f4*>(lds + (1 * W + wid) * H + c) = (f4){p1[v][0], p1[v][1], p1[v][2], p1[v][3]};
            *reinterpret_cast<f4*>(lds + (2 * W + wid) * H + c) = (f4){p2[v][0], p2[v][1], p2[v][2], p2[v][3]};
        }
    }
    __syncthreads();
    const int h4 = H >> 2;
    for (int q = threadIdx.x; q < 3 * h4; q += W * 64) {
        const int k = q / h4, c = (q - k * h4) * 4;
        f4 t = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < W; ++w) t += *reinterpret_cast<const f4*>(lds + (k * W + w) * H + c);
        __builtin_nontemporal_store(t, reinterpret_cast<f4*>(dst + k * H + c));
    }
}

struct ReduceTargets {
    float* t[10];
    int stride[10];  // element stride of the target (1, or 4 for the columns of box_fc.weight [H,4])
};

__global__ __launch_bounds__(NT) void partial_reduce_kernel(const float* __restrict__ ws, int nblk, int K, int H,
                                                            ReduceTargets tg) {
    // 64 (vector, column) pairs per workgroup, 4 slices of the block range each: the loads of a
    // slice are independent and unrolled, so they pipeline instead of paying one L2 round trip each
    __shared__ float red[4][64];
    const int ci = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + ci;
    float s = 0.f;
    if (idx < K * H) {
        const int k = idx / H, c = idx % H;
        const float* p = ws + (int64_t)k * H + c;
        const int64_t stride = (int64_t)K * H;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int b = sl;
        for (; b + 12 < nblk; b += 16) {
            a0 += p[(int64_t)b * stride];
            a1 += p[(int64_t)(b + 4) * stride];
            a2 += p[(int64_t)(b + 8) * stride];
            a3 += p[(int64_t)(b + 12) * stride];
        }
        for (; b < nblk; b += 4) a0 += p[(int64_t)b * stride];
        s = (a0 + a1) + (a2 + a3);
    }
    red[sl][ci] = s;
    __syncthreads();
    if (sl == 0 && idx < K * H) {
        const int k = idx / H, c = idx % H;
        float* t = tg.t[k];
        if (t) t[(int64_t)c * tg.stride[k]] += (red[0][ci] + red[1][ci]) + (red[2][ci] + red[3][ci]);
    }
}

// the same reduction for a batch of workspaces (deferred second stage of many LN backwards)
// 72 jobs: 3.7 KB of kernel arguments (the limit is 4 KB).  A backward pass of the full model leaves ~90 jobs (one per
// LayerNorm, per attention block, per FFN bias): two launches
constexpr int MAX_JOBS = 72;
struct BatchJobs {
    xggm_reduce_job j[MAX_JOBS];
    int start[MAX_JOBS + 1];  // first workgroup of each job
    int n;
};

// 256 threads = 16 column groups (4 consecutive columns, one 16-byte load per partial row) x 16 row slices: a workgroup
// owns 64 columns of one job, a thread the partial rows sl, sl + 16, ... with EIGHT loads in flight -- the 80 / 144 partial
// rows of a LayerNorm backward are then read in one round trip (two for the longest jobs).  History: 4 bytes per lane,
// 64 columns x 4 slices 35 us per launch of ~29 jobs; 16 bytes per lane, 128 columns x 8 slices with four loads in flight
// 37 us per launch of 72 jobs (41 MB = 1.1 TB/s: five dependent round trips per thread, rocprofv3 round 3).
constexpr int RB_COLS = 64;
constexpr int RB_SL = 16;
constexpr int RB_FLIGHT = 16;  // loads a thread issues before it adds the first one
__global__ __launch_bounds__(NT) void partial_reduce_batch_kernel(BatchJobs bj) {
    __shared__ float4 red[RB_SL][RB_COLS / 4];
    int ji = 0;
    for (int k = 1; k < bj.n; ++k)
        if ((int)blockIdx.x >= bj.start[k]) ji = k;
    const xggm_reduce_job job = bj.j[ji];
    const int ci = threadIdx.x & (RB_COLS / 4 - 1), sl = threadIdx.x / (RB_COLS / 4);
    const int idx = (blockIdx.x - bj.start[ji]) * RB_COLS + ci * 4;
    const int KH = job.K * job.H;  // H % 4 == 0: a column group never straddles two of the K rows
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    if (idx < KH) {
        const float* p = job.ws + idx;
        const f4 zero = {0.f, 0.f, 0.f, 0.f};
        // a thread's rows are added in row order whatever the batching: the result does not depend on scheduling
        // (round 4: SIXTEEN loads in flight -- the 144 partial rows of the vision stream's LayerNorm backward took two
        // dependent round trips with eight)
        for (int b = sl; b < job.nblk; b += RB_FLIGHT * RB_SL) {  // read once: non-temporal
            f4 r[RB_FLIGHT];
#pragma unroll
            for (int q = 0; q < RB_FLIGHT; ++q) {
                const int row = b + q * RB_SL;
                r[q] = row < job.nblk ? __builtin_nontemporal_load(reinterpret_cast<const f4*>(p + (int64_t)row * KH)) : zero;
            }
#pragma unroll
            for (int q = 0; q < RB_FLIGHT; ++q) acc += r[q];
        }
    }
    red[sl][ci] = make_float4(acc[0], acc[1], acc[2], acc[3]);
    __syncthreads();
    if (sl == 0 && idx < KH) {
        const int k = idx / job.H, c = idx % job.H;
        float* t = job.target[k];
        if (t) {
            float4 s = red[0][ci];
#pragma unroll
            for (int q = 1; q < RB_SL; ++q) {  // fixed order
                const float4 r = red[q][ci];
                s.x += r.x; s.y += r.y; s.z += r.z; s.w += r.w;
            }
            t[c] += s.x; t[c + 1] += s.y; t[c + 2] += s.z; t[c + 3] += s.w;
        }
    }
}

// ------------------------------------------------------------------------------- LN fwd
// Up to MAX_SEG independent row sets (the language and the vision stream of one layer) share a
// launch: each is far too small to fill the GPU and the launches are latency-bound.
constexpr int MAX_SEG = 4;
struct LnFwdGroup {
    xggm_ln_fwd_problem s[MAX_SEG];
    int start[MAX_SEG + 1];  // first workgroup of each segment
    int n;
};

// rows in flight per workgroup; same-box A/B per iteration: 4 -> 8 rows -0.05 ms, 8 -> 16 rows +0.07 ms
constexpr int LN_FWD_W = 8;
template <typename T, int NV>
__global__ __launch_bounds__(LN_FWD_W * 64) void ln_fwd_kernel(LnFwdGroup G, int H, float eps, float p_pre, float p_post,
                                                    const uint64_t* rng, int accumulate, float out_scale, PrefetchArgs pf) {
    if ((int)blockIdx.x >= G.start[G.n]) {  // appended workgroups: read the queued weight ranges (common.h)
        prefetch_role(pf, (int)blockIdx.x - G.start[G.n]);
        return;
    }
    int si = 0;
#pragma unroll
    for (int k = 1; k < MAX_SEG; ++k)
        if (k < G.n && (int)blockIdx.x >= G.start[k]) si = k;
    const xggm_ln_fwd_problem sg = G.s[si];
    const T* in = reinterpret_cast<const T*>(sg.in);
    const float* __restrict__ bias = sg.bias;
    const T* __restrict__ residual = reinterpret_cast<const T*>(sg.residual);
    const float* __restrict__ gamma = sg.gamma;
    const float* __restrict__ beta = sg.beta;
    T* out = reinterpret_cast<T*>(sg.out);
    T* z_out = reinterpret_cast<T*>(sg.z_out);
    float* stats = sg.stats;
    const int M = sg.M;
    const DropArgs d{p_pre, p_post, rng, sg.sid_pre, sg.sid_post};
    const int blk = blockIdx.x - G.start[si], nblk = G.start[si + 1] - G.start[si];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint64_t seed, off;
    rng_load(d.rng, seed, off);
    const float ik_pre = d.p_pre > 0.f ? 1.f / (1.f - d.p_pre) : 1.f;
    const float ik_post = d.p_post > 0.f ? 1.f / (1.f - d.p_post) : 1.f;
    unsigned char* out8 = sizeof(T) == 2 ? reinterpret_cast<unsigned char*>(sg.out8) : nullptr;
    const Q8 qs(out8 ? sg.qscale : nullptr);
    const float q8 = qs.q;
    float amax8 = 0.f;
    // Every load of a row is issued before the first one is used: the row's critical path is then ONE memory round
    // trip for the inputs (+ one for the accumulate target) instead of one per vector and operand.  The first version
    // walked the vectors one by one, each behind its own `c < H` branch, and hipcc put an s_waitcnt vmcnt(0) behind
    // every load -- nine to twelve serial round trips per row, which WAS the kernel (9 us per launch for 5 MB).
    // Columns past H are read at a clamped address and masked out of the arithmetic; only the stores are predicated.
    int cc[NV];
    bool ok[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int c = (v * 64 + lane) * 4;
        ok[v] = c < H;
        cc[v] = ok[v] ? c : H - 4;
    }
    float g4[NV][4], be4[NV][4];
    {   // gamma / beta do not depend on the row
        float4 rg[NV], rbt[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            rg[v] = load_raw4<float>(gamma + cc[v]);
            rbt[v] = load_raw4<float>(beta + cc[v]);
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            cvt4(rg[v], g4[v]);
            cvt4(rbt[v], be4[v]);
        }
    }
    float b4[NV][4];
    if (bias) {
        float4 rb4[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) rb4[v] = load_raw4<float>(bias + cc[v]);
#pragma unroll
        for (int v = 0; v < NV; ++v) cvt4(rb4[v], b4[v]);
    } else {
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int i = 0; i < 4; ++i) b4[v][i] = 0.f;
    }
    for (int row = blk * LN_FWD_W + wid; row < M; row += nblk * LN_FWD_W) {
        const int64_t rb = (int64_t)row * H;
        float z[NV][4];
        typename Raw4<T>::type rin[NV], rres[NV], racc[NV];
        if (sg.in_slabs > 0) {  // split-K partial sums (fp32), added in slab order
            const float* part = reinterpret_cast<const float*>(sg.in) + rb;
            const int64_t slab = (int64_t)M * H;
            float4 rs[4][NV];
#pragma unroll
            for (int sl = 0; sl < 4; ++sl)
                if (sl < sg.in_slabs) {  // uniform: the loads of one slab back to back, nothing used in between
#pragma unroll
                    for (int v = 0; v < NV; ++v) rs[sl][v] = load_raw4<float>(part + sl * slab + cc[v]);
                }
#pragma unroll
            for (int v = 0; v < NV; ++v) cvt4(rs[0][v], z[v]);
#pragma unroll
            for (int sl = 1; sl < 4; ++sl)
                if (sl < sg.in_slabs) {
#pragma unroll
                    for (int v = 0; v < NV; ++v) {
                        float t4[4];
                        cvt4(rs[sl][v], t4);
#pragma unroll
                        for (int i = 0; i < 4; ++i) z[v][i] += t4[i];
                    }
                }
            for (int sl = 4; sl < sg.in_slabs; ++sl) {
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    float t4[4];
                    load4(part + sl * slab + cc[v], t4);
#pragma unroll
                    for (int i = 0; i < 4; ++i) z[v][i] += t4[i];
                }
            }
        } else {
#pragma unroll
            for (int v = 0; v < NV; ++v) rin[v] = load_raw4<T>(in + rb + cc[v]);
        }
        if (residual) {
#pragma unroll
            for (int v = 0; v < NV; ++v) rres[v] = load_raw4<T>(residual + rb + cc[v]);
        }
        if (accumulate) {
#pragma unroll
            for (int v = 0; v < NV; ++v) racc[v] = load_raw4<T>(out + rb + cc[v]);
        }
        if (sg.in_slabs <= 0) {
#pragma unroll
            for (int v = 0; v < NV; ++v) cvt4(rin[v], z[v]);
        }
        float sum = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
#pragma unroll
            for (int i = 0; i < 4; ++i) z[v][i] += b4[v][i];
            if (d.p_pre > 0.f) {
                float s4[4];
                dropout_scale4(d.p_pre, ik_pre, seed, off, d.s_pre, (uint64_t)(rb + cc[v]), s4);
#pragma unroll
                for (int i = 0; i < 4; ++i) z[v][i] *= s4[i];
            }
            if (residual) {
                float r4[4];
                cvt4(rres[v], r4);
#pragma unroll
                for (int i = 0; i < 4; ++i) z[v][i] += r4[i];
            }
            if (z_out) {
#pragma unroll
                for (int i = 0; i < 4; ++i) z[v][i] = round_to<T>(z[v][i]);
                if (ok[v]) store4(z_out + rb + cc[v], z[v]);
            }
            if (ok[v]) {
#pragma unroll
                for (int i = 0; i < 4; ++i) sum += z[v][i];
            }
        }
        const float mean = wave_sum(sum) / (float)H;
        float var = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            if (ok[v]) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float t = z[v][i] - mean;
                    var += t * t;
                }
            }
        }
        const float rstd = rsqrtf(wave_sum(var) / (float)H + eps);
        if (stats && lane == 0) {
            stats[2 * row] = mean;
            stats[2 * row + 1] = rstd;
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            float y[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) y[i] = (z[v][i] - mean) * rstd * g4[v][i] + be4[v][i];
            if (d.p_post > 0.f) {
                float s4[4];
                dropout_scale4(d.p_post, ik_post, seed, off, d.s_post, (uint64_t)(rb + cc[v]), s4);
#pragma unroll
                for (int i = 0; i < 4; ++i) y[i] *= s4[i];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) y[i] *= out_scale;
            if (accumulate) {
                float o4[4];
                cvt4(racc[v], o4);
#pragma unroll
                for (int i = 0; i < 4; ++i) y[i] += o4[i];
            }
            if (ok[v]) {
                store4(out + rb + cc[v], y);
                if (out8) {  // e4m3 operand of the next fp8 product: 4 bytes per lane, 256 contiguous bytes per wave
                    amax8 = fmaxf(fmaxf(amax8, fmaxf(fabsf(y[0]), fabsf(y[1]))), fmaxf(fabsf(y[2]), fabsf(y[3])));
                    *reinterpret_cast<int*>(out8 + rb + cc[v]) = pack4_e4m3(y[0], y[1], y[2], y[3], q8);
                }
            }
        }
    }
    if (out8 && sg.amax) {  // one atomic per workgroup
        __shared__ float red8[LN_FWD_W];
        const float wv = wave_max(amax8);
        if (lane == 0) red8[wid] = wv;
        __syncthreads();
        if (threadIdx.x == 0) {
            float bm = red8[0];
#pragma unroll
            for (int w = 1; w < LN_FWD_W; ++w) bm = fmaxf(bm, red8[w]);
            if (bm > qs.thr) amax_record(sg.amax, sg.amax_slots, (int)blockIdx.x, bm);
        }
    }
}

// ------------------------------------------------------------------------------- sum of LayerNorms
// The jump-knowledge read-out of the graph blocks (src/module/gcn.py:70-77, src/module/gin.py:80-87) adds up to four
// dropout(LayerNorm_k(x_k)): launched as a chain of accumulating ln_fwd calls it was k launches that each re-read and
// re-wrote the sum (rounded to the storage type in between); here a wave holds the row's sum in fp32 registers while it
// walks the terms, and the sum is rounded once.  The terms have no bias / residual / input dropout, so the saved
// pre-normalisation row IS the input: only (mean, rstd) are written per term.
struct LnSumArgs {
    const void* in[4];
    const float* gamma[4];
    const float* beta[4];
    float* stats[4];
    uint32_t sid_post[4];
    void* out;
    int n, M;
};
template <typename T, int NV>
__global__ __launch_bounds__(LN_FWD_W * 64) void ln_sum_fwd_kernel(LnSumArgs a, int H, float eps, float p_post, const uint64_t* rng) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint64_t seed, off;
    rng_load(rng, seed, off);
    const float ik_post = p_post > 0.f ? 1.f / (1.f - p_post) : 1.f;
    int cc[NV];
    bool ok[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int c = (v * 64 + lane) * 4;
        ok[v] = c < H;
        cc[v] = ok[v] ? c : H - 4;
    }
    T* out = reinterpret_cast<T*>(a.out);
    for (int row = blockIdx.x * LN_FWD_W + wid; row < a.M; row += gridDim.x * LN_FWD_W) {
        const int64_t rb = (int64_t)row * H;
        typename Raw4<T>::type rin[4][NV];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < a.n) {  // uniform: every term's row is on its way before the first one is used
#pragma unroll
                for (int v = 0; v < NV; ++v) rin[k][v] = load_raw4<T>(reinterpret_cast<const T*>(a.in[k]) + rb + cc[v]);
            }
        float acc[NV][4];
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[v][i] = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < a.n) {
                float z[NV][4];
                float sum = 0.f;
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    cvt4(rin[k][v], z[v]);
                    if (ok[v]) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) sum += z[v][i];
                    }
                }
                const float mean = wave_sum(sum) / (float)H;
                float var = 0.f;
#pragma unroll
                for (int v = 0; v < NV; ++v)
                    if (ok[v]) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float t = z[v][i] - mean;
                            var += t * t;
                        }
                    }
                const float rstd = rsqrtf(wave_sum(var) / (float)H + eps);
                if (a.stats[k] && lane == 0) {
                    a.stats[k][2 * row] = mean;
                    a.stats[k][2 * row + 1] = rstd;
                }
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    float g4[4], be4[4], y[4];
                    load4(a.gamma[k] + cc[v], g4);
                    load4(a.beta[k] + cc[v], be4);
#pragma unroll
                    for (int i = 0; i < 4; ++i) y[i] = (z[v][i] - mean) * rstd * g4[i] + be4[i];
                    if (p_post > 0.f) {
                        float s4[4];
                        dropout_scale4(p_post, ik_post, seed, off, a.sid_post[k], (uint64_t)(rb + cc[v]), s4);
#pragma unroll
                        for (int i = 0; i < 4; ++i) y[i] *= s4[i];
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[v][i] += y[i];
                }
            }
#pragma unroll
        for (int v = 0; v < NV; ++v)
            if (ok[v]) store4(out + rb + cc[v], acc[v]);
    }
}

// ------------------------------------------------------------------------------- LN bwd
// dy: gradient w.r.t. `out`.  Produces d_in (grad of `in`, i.e. through drop_pre), d_res
// (grad of `residual`), and accumulates dgamma / dbeta / dbias (bias of `in`) with one fp32
// atomic per column per wave after an in-register partial sum over the wave's rows.
struct LnBwdGroup {
    xggm_ln_bwd_problem s[MAX_SEG];
    int start[MAX_SEG + 1];
    int n;
};

// LN backward runs LN_BWD_W waves (rows in flight) per workgroup: eight halve the number of partial rows it writes
// and the reduce launch reads (one [3, H] row per workgroup)
constexpr int LN_BWD_W = 8;  // (same-box A/B against 4: 12.32 vs 12.37 ms per iteration)
// the three partial rows go through LDS together when 3 x W x H floats fit beside nothing else (H = 768: 72 KB)
__host__ __device__ inline bool ln_bwd_fused_tail(int H) { return (size_t)3 * LN_BWD_W * H * sizeof(float) <= 144 * 1024; }
template <typename T, int NV>
__global__ __launch_bounds__(LN_BWD_W * 64) void ln_bwd_kernel(LnBwdGroup G, int H, float p_pre, float p_post,
                                                             const uint64_t* rng, float out_scale, PrefetchArgs pf) {
    if ((int)blockIdx.x >= G.start[G.n]) {  // appended workgroups: read the queued weight ranges (common.h)
        prefetch_role(pf, (int)blockIdx.x - G.start[G.n]);
        return;
    }
    int si = 0;
#pragma unroll
    for (int k = 1; k < MAX_SEG; ++k)
        if (k < G.n && (int)blockIdx.x >= G.start[k]) si = k;
    const xggm_ln_bwd_problem sg = G.s[si];
    const T* __restrict__ dy = reinterpret_cast<const T*>(sg.dy);
    const T* __restrict__ z = reinterpret_cast<const T*>(sg.z);
    const float* __restrict__ stats = sg.stats;
    const float* __restrict__ gamma = sg.gamma;
    T* d_in = reinterpret_cast<T*>(sg.d_in);
    T* d_res = reinterpret_cast<T*>(sg.d_res);
    float* ws = sg.ws;
    const int M = sg.M, accumulate_dres = sg.accumulate_dres;
    const T* __restrict__ gelu_aux = reinterpret_cast<const T*>(sg.gelu_aux);
    const DropArgs d{p_pre, p_post, rng, sg.sid_pre, sg.sid_post};
    const int blk = blockIdx.x - G.start[si], nblk = G.start[si + 1] - G.start[si];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint64_t seed, off;
    rng_load(d.rng, seed, off);
    const float ik_pre = d.p_pre > 0.f ? 1.f / (1.f - d.p_pre) : 1.f;
    const float ik_post = d.p_post > 0.f ? 1.f / (1.f - d.p_post) : 1.f;
    // as in the forward: clamped columns, every load of a row issued before the first use (one memory round trip per
    // row instead of one per vector and operand); only stores and the column partials are predicated
    float pg[NV][4], pb[NV][4], pbias[NV][4], g4[NV][4];
    int cc[NV];
    bool ok[NV];
    {
        float4 rg[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int c = (v * 64 + lane) * 4;
            ok[v] = c < H;
            cc[v] = ok[v] ? c : H - 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) pg[v][i] = pb[v][i] = pbias[v][i] = 0.f;
            rg[v] = load_raw4<float>(gamma + cc[v]);
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) cvt4(rg[v], g4[v]);
    }
    for (int row = blk * LN_BWD_W + wid; row < M; row += nblk * LN_BWD_W) {
        const int64_t rb = (int64_t)row * H;
        typename Raw4<T>::type rdy[NV], rz[NV], rprev[NV], raux[NV];
        const float2 st = *reinterpret_cast<const float2*>(stats + 2 * row);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            rdy[v] = load_raw4<T>(dy + rb + cc[v]);
            rz[v] = load_raw4<T>(z + rb + cc[v]);
        }
        if (d_res && accumulate_dres) {
#pragma unroll
            for (int v = 0; v < NV; ++v) rprev[v] = load_raw4<T>(d_res + rb + cc[v]);
        }
        if (gelu_aux) {
#pragma unroll
            for (int v = 0; v < NV; ++v) raux[v] = load_raw4<T>(gelu_aux + rb + cc[v]);
        }
        const float mean = st.x, rstd = st.y;
        float dyn[NV][4], xh[NV][4];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            float zz[4];
            cvt4(rdy[v], dyn[v]);
            cvt4(rz[v], zz);
            if (d.p_post > 0.f) {
                float s4[4];
                dropout_scale4(d.p_post, ik_post, seed, off, d.s_post, (uint64_t)(rb + cc[v]), s4);
#pragma unroll
                for (int i = 0; i < 4; ++i) dyn[v][i] *= s4[i];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                dyn[v][i] *= out_scale;
                xh[v][i] = (zz[i] - mean) * rstd;
            }
            if (ok[v]) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    pg[v][i] += dyn[v][i] * xh[v][i];
                    pb[v][i] += dyn[v][i];
                    const float t = dyn[v][i] * g4[v][i];
                    s1 += t;
                    s2 += t * xh[v][i];
                }
            }
        }
        s1 = wave_sum(s1) / (float)H;
        s2 = wave_sum(s2) / (float)H;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            float dz[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) dz[i] = rstd * (dyn[v][i] * g4[v][i] - s1 - xh[v][i] * s2);
            if (d_res) {
                if (accumulate_dres) {
                    float o4[4], w4[4];
                    cvt4(rprev[v], o4);
#pragma unroll
                    for (int i = 0; i < 4; ++i) w4[i] = o4[i] + dz[i];
                    if (ok[v]) store4(d_res + rb + cc[v], w4);
                } else if (ok[v]) {
                    store4(d_res + rb + cc[v], dz);
                }
            }
            if (d.p_pre > 0.f) {
                float s4[4];
                dropout_scale4(d.p_pre, ik_pre, seed, off, d.s_pre, (uint64_t)(rb + cc[v]), s4);
#pragma unroll
                for (int i = 0; i < 4; ++i) dz[i] *= s4[i];
            }
            if (gelu_aux) {  // `in` was gelu(u): chain through the activation
                float u4[4];
                cvt4(raux[v], u4);
#pragma unroll
                for (int i = 0; i < 4; ++i) dz[i] *= gelu_grad_f(u4[i]);
            }
            if (ok[v]) {
                if (d_in) store4(d_in + rb + cc[v], dz);
#pragma unroll
                for (int i = 0; i < 4; ++i) pbias[v][i] += dz[i];
            }
        }
    }
    extern __shared__ __attribute__((aligned(16))) float red_lds[];
    float* wsb = ws + (int64_t)blk * 3 * H;
    if (ln_bwd_fused_tail(H)) {
        block_store_partial3<NV, LN_BWD_W>(pg, pb, pbias, red_lds, wsb, H, lane, wid);
    } else {
        block_store_partial<NV, LN_BWD_W>(pg, red_lds, wsb, H, lane, wid);
        block_store_partial<NV, LN_BWD_W>(pb, red_lds, wsb + H, H, lane, wid);
        block_store_partial<NV, LN_BWD_W>(pbias, red_lds, wsb + 2 * H, H, lane, wid);
    }
}

// ------------------------------------------------------------------------------- embeddings
// e4m3 copy of a row kernel's output (fp8 forward, BASELINE configs[4]): the two encoder inputs leave their producers
// as e4m3 next to the bf16 form, like every other operand of an fp8 product -- no quantisation launch in the step
struct Emit8 {
    unsigned char* out8;  // or null
    const float* qscale;
    float* amax;
    int slots;
};
template <int W> __device__ __forceinline__ void emit8_record(const Emit8& e8, const Q8& qs, float amax8, int lane, int wid) {
    __shared__ float red8[W];
    const float wv = wave_max(amax8);
    if (lane == 0) red8[wid] = wv;
    __syncthreads();
    if (threadIdx.x == 0) {
        float bm = red8[0];
#pragma unroll
        for (int w = 1; w < W; ++w) bm = fmaxf(bm, red8[w]);
        if (bm > qs.thr) amax_record(e8.amax, e8.slots, (int)blockIdx.x, bm);  // one atomic per workgroup, spread over the slots
    }
}

// Input glue of a pass riding on its first row kernel (xggm_embed_fwd_side_*): the additive attention mask
// (src/lxrt/modeling.py:919-928) and the fp32 -> bf16 casts of the visual inputs were launches of their own in front
// of the embeddings; here workgroups appended to the embedding kernel's grid do them.  Nothing in the embedding kernel
// reads their outputs.
struct SideJobs {
    int n, main_blocks;
    int kind[XGGM_SIDE_MAX];
    const void* src[XGGM_SIDE_MAX];
    void* dst[XGGM_SIDE_MAX];
    long long count[XGGM_SIDE_MAX];
    int blk0[XGGM_SIDE_MAX + 1];  // first appended workgroup of each job
};
__device__ __forceinline__ void side_role(const SideJobs& sj, int b) {
    int k = 0;
#pragma unroll
    for (int j = 1; j < XGGM_SIDE_MAX; ++j)
        if (j < sj.n && b >= sj.blk0[j]) k = j;
    const int nb = sj.blk0[k + 1] - sj.blk0[k];
    const long long n = sj.count[k], stride = (long long)nb * NT;
    long long i = (long long)(b - sj.blk0[k]) * NT + threadIdx.x;
    if (sj.kind[k] == XGGM_SIDE_ADDITIVE_MASK) {
        const int64_t* m = reinterpret_cast<const int64_t*>(sj.src[k]);
        float* o = reinterpret_cast<float*>(sj.dst[k]);
        for (; i < n; i += stride) o[i] = (1.0f - (float)m[i]) * -10000.0f;
    } else {  // XGGM_SIDE_CAST_BF16
        const float* x = reinterpret_cast<const float*>(sj.src[k]);
        bf16* o = reinterpret_cast<bf16*>(sj.dst[k]);
        for (; i < n; i += stride) o[i] = __float2bfloat16(x[i]);
    }
}

template <typename T, int NV>
__global__ __launch_bounds__(NT) void embed_fwd_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ seg,
                                                       const T* __restrict__ word, const T* __restrict__ pos,
                                                       const T* __restrict__ type, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, T* out, T* z_out, float* stats, int M,
                                                       int Tlen, int H, float eps, DropArgs d, Emit8 e8, SideJobs sj) {
    if ((int)blockIdx.x >= sj.main_blocks) {  // appended workgroups (whole workgroups: no barrier below is left waiting)
        side_role(sj, (int)blockIdx.x - sj.main_blocks);
        return;
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint64_t seed, off;
    rng_load(d.rng, seed, off);
    const float ik_post = d.p_post > 0.f ? 1.f / (1.f - d.p_post) : 1.f;
    unsigned char* out8 = sizeof(T) == 2 ? e8.out8 : nullptr;
    const Q8 qs(out8 ? e8.qscale : nullptr);
    float amax8 = 0.f;
    for (int row = blockIdx.x * WPB + wid; row < M; row += sj.main_blocks * WPB) {
        const int64_t rb = (int64_t)row * H;
        const int64_t wi = ids[row], pi = row % Tlen, ti = seg ? seg[row] : 0;
        float z[NV][4];
        float sum = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int c = (v * 64 + lane) * 4;
            if (c < H) {
                float a[4], b[4], e[4];
                load4(word + wi * H + c, a);
                load4(pos + pi * H + c, b);
                load4(type + ti * H + c, e);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    z[v][i] = round_to<T>(a[i] + b[i] + e[i]);
                    sum += z[v][i];
                }
                store4(z_out + rb + c, z[v]);
            }
        }
        const float mean = wave_sum(sum) / (float)H;
        float var = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int c = (v * 64 + lane) * 4;
            if (c < H) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float t = z[v][i] - mean;
                    var += t * t;
                }
            }
        }
        const float rstd = rsqrtf(wave_sum(var) / (float)H + eps);
        if (lane == 0) {
            stats[2 * row] = mean;
            stats[2 * row + 1] = rstd;
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int c = (v * 64 + lane) * 4;
            if (c < H) {
                float g4[4], b4[4], y[4];
                load4(gamma + c, g4);
                load4(beta + c, b4);
#pragma unroll
                for (int i = 0; i < 4; ++i) y[i] = (z[v][i] - mean) * rstd * g4[i] + b4[i];
                if (d.p_post > 0.f) {
                    float s4[4];
                    dropout_scale4(d.p_post, ik_post, seed, off, d.s_post, (uint64_t)(rb + c), s4);
#pragma unroll
                    for (int i = 0; i < 4; ++i) y[i] *= s4[i];
                }
                store4(out + rb + c, y);
                if (out8) {
                    amax8 = fmaxf(fmaxf(amax8, fmaxf(fabsf(y[0]), fabsf(y[1]))), fmaxf(fabsf(y[2]), fabsf(y[3])));
                    *reinterpret_cast<int*>(out8 + rb + c) = pack4_e4m3(y[0], y[1], y[2], y[3], qs.q);
                }
            }
        }
    }
    if (out8 && e8.amax) emit8_record<WPB>(e8, qs, amax8, lane, wid);
}

// The gradients of the three embedding tables: table row k receives the sum of the dz rows that looked it up; row 0 of
// every table is nn.Embedding's padding_idx and never receives gradient (modeling.py:284-290).
// No scatter, no atomics -- an OWNER gathers: one wave per candidate (word: every row of the batch; type: every row,
// when there are segment ids; position: every position).  The wave of row r owns table row ids[r] iff no earlier row
// looked up the same id; it then lists the rows r' >= r with that id (ballot + prefix count into LDS), adds their dz
// rows IN ROW ORDER, eight loads in flight at a time, and adds the sum to the table gradient (one owner per address:
// a plain read-modify-write).  The sum's order is fixed by the data, not by the scheduling: same inputs, same bits.
// ([CLS] / [SEP] occur once per sample: 32 serial atomics per address before, four rounds of eight loads now.)
constexpr int ES_CAP = 512;  // list entries per wave; longer lists are summed in several rounds
// Optional by-products for the caller that keeps the word table's gradient row-sparse between passes (the table is 94 MB
// of which a pass touches <= B * T rows): ids[r] = the table row batch row r looked up, sq[r] = |gradient row|^2 after
// the add where r is the row's owner and 0 elsewhere, *n = M.  All three null: nothing is written.
struct RowList {
    int64_t* ids;
    float* sq;
    int* n;
};
template <typename T, int NV>
__global__ __launch_bounds__(NT) void embed_scatter_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ seg,
                                                           const T* __restrict__ dz, float* dword, float* dpos, float* dtype,
                                                           int M, int Tlen, int H, RowList rl) {
    __shared__ int list[WPB][ES_CAP];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int task = blockIdx.x * WPB + wid;  // wave-uniform
    const int n_word = M, n_type = seg ? M : 0;
    if (task >= n_word + n_type + Tlen) return;
    int cc[NV];
    bool ok[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int c = (v * 64 + lane) * 4;
        ok[v] = c < H;
        cc[v] = ok[v] ? c : H - 4;
    }
    float acc[NV][4];
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[v][i] = 0.f;
    int* mine = list[wid];
    // the list is written and read by the lanes of ONE wave: order its LDS traffic, no workgroup barrier (waves of a
    // workgroup return at different points)
    auto wsync = [] {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    // add the dz rows mine[0 .. cnt) in list order, eight rows' loads issued before the first add
    auto add_listed = [&](int cnt) {
        for (int i0 = 0; i0 < cnt; i0 += 8) {
            typename Raw4<T>::type raw[8][NV];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int row = mine[min(i0 + u, cnt - 1)];
#pragma unroll
                for (int v = 0; v < NV; ++v) raw[u][v] = load_raw4<T>(dz + (int64_t)row * H + cc[v]);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (i0 + u < cnt) {
#pragma unroll
                    for (int v = 0; v < NV; ++v) {
                        float t4[4];
                        cvt4(raw[u][v], t4);
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[v][i] += t4[i];
                    }
                }
        }
    };
    float* target = nullptr;
    bool listed = false;  // wave-uniform: this wave owns a listed row of the word table
    int list_slot = 0;
    if (task < n_word + n_type) {
        const bool is_word = task < n_word;
        const int r = is_word ? task : task - n_word;
        const int64_t* key = is_word ? ids : seg;
        const int64_t k = key[r];
        if (is_word && rl.ids && lane == 0) {  // the rows of the word table this pass leaves non-zero (xggm_embed_bwd_listed_*)
            rl.ids[r] = k;
            rl.sq[r] = 0.f;  // the owner of the row overwrites it below
            if (r == 0) *rl.n = M;
        }
        if (k == 0) return;  // padding_idx
        // owner = first row with this key
        bool seen = false;
        for (int q0 = 0; q0 < r && !seen; q0 += 64) {
            const int q = q0 + lane;
            seen = __ballot(q < r && key[q] == k) != 0ull;
        }
        if (seen) return;
        listed = is_word && rl.ids;
        list_slot = r;
        int cnt = 0;
        for (int q0 = r & ~63; q0 < M; q0 += 64) {
            const int q = q0 + lane;
            const bool hit = q >= r && q < M && key[q] == k;
            const unsigned long long m = __ballot(hit);
            if (hit) mine[cnt + __popcll(m & ((1ull << lane) - 1ull))] = q;
            cnt += __popcll(m);
            if (cnt > ES_CAP - 64) {  // wave-uniform
                wsync();
                add_listed(cnt);
                wsync();
                cnt = 0;
            }
        }
        wsync();
        add_listed(cnt);
        target = (is_word ? dword : dtype) + k * H;
    } else {
        const int t = task - n_word - n_type;
        if (t == 0) return;  // padding_idx of the position table
        const int B = M / Tlen;
        for (int b0 = 0; b0 < B; b0 += ES_CAP) {
            const int n = min(ES_CAP, B - b0);
            for (int i = lane; i < n; i += 64) mine[i] = (b0 + i) * Tlen + t;
            wsync();
            add_listed(n);
            wsync();
        }
        target = dpos + (int64_t)t * H;
    }
    float sq = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v)
        if (ok[v]) {
            float o[4];
            load4(target + cc[v], o);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                o[i] += acc[v][i];
                sq += o[i] * o[i];
            }
            store4(target + cc[v], o);
        }
    if (listed) {  // |row|^2 of the table gradient as it stands now: clip_grad_norm_ adds these instead of reading the table
        sq = wave_sum(sq);
        if (lane == 0) rl.sq[list_slot] = sq;
    }
}

// ------------------------------------------------------------------------------- visual embedding
// u = feat @ W_f^T comes from the GEMM (no bias); this kernel adds b_f, normalises, computes
// the K = 4 box projection in registers, normalises it, averages and applies dropout.
// Saved for backward: z1 = u + b_f (T, over u in place), z2 = box projection (T), stats [M,4].
template <typename T, int NV>
__global__ __launch_bounds__(NT) void visn_embed_fwd_kernel(const T* u, const float* __restrict__ bf,
                                                            const T* __restrict__ boxes, const float* __restrict__ Wb,
                                                            const float* __restrict__ bb, const float* __restrict__ g1,
                                                            const float* __restrict__ b1, const float* __restrict__ g2,
                                                            const float* __restrict__ b2, T* out, T* z1_out, T* z2_out,
                                                            float* stats, int M, int H, float eps, DropArgs d, Emit8 e8) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint64_t seed, off;
    rng_load(d.rng, seed, off);
    const float ik_post = d.p_post > 0.f ? 1.f / (1.f - d.p_post) : 1.f;
    unsigned char* out8 = sizeof(T) == 2 ? e8.out8 : nullptr;
    const Q8 qs(out8 ? e8.qscale : nullptr);
    float amax8 = 0.f;
    for (int row = blockIdx.x * WPB + wid; row < M; row += gridDim.x * WPB) {
        const int64_t rb = (int64_t)row * H;
        float bx[4];
        load4(boxes + (int64_t)row * 4, bx);
        float z1[NV][4], z2[NV][4];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int c = (v * 64 + lane) * 4;
            if (c < H) {
                float f4[4], bb4[4];
                load4(u + rb + c, z1[v]);
                load4(bf + c, f4);
                load4(bb + c, bb4);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float w4[4];
                    load4(Wb + (int64_t)(c + i) * 4, w4);
                    z1[v][i] = round_to<T>(z1[v][i] + f4[i]);
                    z2[v][i] = round_to<T>(bb4[i] + w4[0] * bx[0] + w4[1] * bx[1] + w4[2] * bx[2] + w4[3] * bx[3]);
                    s1 += z1[v][i];
                    s2 += z2[v][i];
                }
                store4(z1_out + rb + c, z1[v]);
                store4(z2_out + rb + c, z2[v]);
            }
        }
        const float m1 = wave_sum(s1) / (float)H, m2 = wave_sum(s2) / (float)H;
        float v1 = 0.f, v2 = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int c = (v * 64 + lane) * 4;
            if (c < H) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float t1 = z1[v][i] - m1, t2 = z2[v][i] - m2;
                    v1 += t1 * t1;
                    v2 += t2 * t2;
                }
            }
        }
        const float r1 = rsqrtf(wave_sum(v1) / (float)H + eps), r2 = rsqrtf(wave_sum(v2) / (float)H + eps);
        if (lane == 0) {
            stats[4 * row] = m1;
            stats[4 * row + 1] = r1;
            stats[4 * row + 2] = m2;
            stats[4 * row + 3] = r2;
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int c = (v * 64 + lane) * 4;
            if (c < H) {
                float ga[4], ba[4], gb[4], bbv[4], y[4];
                load4(g1 + c, ga);
                load4(b1 + c, ba);
                load4(g2 + c, gb);
                load4(b2 + c, bbv);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    y[i] = 0.5f * (((z1[v][i] - m1) * r1 * ga[i] + ba[i]) + ((z2[v][i] - m2) * r2 * gb[i] + bbv[i]));
                if (d.p_post > 0.f) {
                    float s4[4];
                    dropout_scale4(d.p_post, ik_post, seed, off, d.s_post, (uint64_t)(rb + c), s4);
#pragma unroll
                    for (int i = 0; i < 4; ++i) y[i] *= s4[i];
                }
                store4(out + rb + c, y);
                if (out8) {
                    amax8 = fmaxf(fmaxf(amax8, fmaxf(fabsf(y[0]), fabsf(y[1]))), fmaxf(fabsf(y[2]), fabsf(y[3])));
                    *reinterpret_cast<int*>(out8 + rb + c) = pack4_e4m3(y[0], y[1], y[2], y[3], qs.q);
                }
            }
        }
    }
    if (out8 && e8.amax) emit8_record<WPB>(e8, qs, amax8, lane, wid);
}

// backward: du (grad of the GEMM output u, T), and atomically accumulated fp32 grads of
// b_f, g1, b1, W_b [H,4], b_b, g2, b2.
template <typename T, int NV>
__global__ __launch_bounds__(NT) void visn_embed_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ z1,
                                                            const T* __restrict__ z2, const float* __restrict__ stats,
                                                            const T* __restrict__ boxes, const float* __restrict__ g1,
                                                            const float* __restrict__ g2, T* du, float* ws, int M, int H,
                                                            DropArgs d) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint64_t seed, off;
    rng_load(d.rng, seed, off);
    const float ik_post = d.p_post > 0.f ? 1.f / (1.f - d.p_post) : 1.f;
    float pg1[NV][4], pb1[NV][4], pbf[NV][4], pg2[NV][4], pb2[NV][4], pbb[NV][4], pW[NV][4][4], ga[NV][4], gb[NV][4];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int c = (v * 64 + lane) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            pg1[v][i] = pb1[v][i] = pbf[v][i] = pg2[v][i] = pb2[v][i] = pbb[v][i] = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) pW[v][i][j] = 0.f;
        }
        if (c < H) {
            load4(g1 + c, ga[v]);
            load4(g2 + c, gb[v]);
        }
    }
    for (int row = blockIdx.x * WPB + wid; row < M; row += gridDim.x * WPB) {
        const int64_t rb = (int64_t)row * H;
        const float m1 = stats[4 * row], r1 = stats[4 * row + 1], m2 = stats[4 * row + 2], r2 = stats[4 * row + 3];
        float bx[4];
        load4(boxes + (int64_t)row * 4, bx);
        float dyn[NV][4], x1[NV][4], x2[NV][4];
        float a1 = 0.f, a2 = 0.f, c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int c = (v * 64 + lane) * 4;
            if (c < H) {
                float za[4], zb[4];
                load4(dy + rb + c, dyn[v]);
                load4(z1 + rb + c, za);
                load4(z2 + rb + c, zb);
                if (d.p_post > 0.f) {
                    float s4[4];
                    dropout_scale4(d.p_post, ik_post, seed, off, d.s_post, (uint64_t)(rb + c), s4);
#pragma unroll
                    for (int i = 0; i < 4; ++i) dyn[v][i] *= s4[i];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    dyn[v][i] *= 0.5f;
                    x1[v][i] = (za[i] - m1) * r1;
                    x2[v][i] = (zb[i] - m2) * r2;
                    pg1[v][i] += dyn[v][i] * x1[v][i];
                    pb1[v][i] += dyn[v][i];
                    pg2[v][i] += dyn[v][i] * x2[v][i];
                    pb2[v][i] += dyn[v][i];
                    const float t1 = dyn[v][i] * ga[v][i], t2 = dyn[v][i] * gb[v][i];
                    a1 += t1;
                    c1 += t1 * x1[v][i];
                    a2 += t2;
                    c2 += t2 * x2[v][i];
                }
            }
        }
        a1 = wave_sum(a1) / (float)H;
        c1 = wave_sum(c1) / (float)H;
        a2 = wave_sum(a2) / (float)H;
        c2 = wave_sum(c2) / (float)H;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int c = (v * 64 + lane) * 4;
            if (c < H) {
                float d1[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    d1[i] = r1 * (dyn[v][i] * ga[v][i] - a1 - x1[v][i] * c1);
                    const float d2 = r2 * (dyn[v][i] * gb[v][i] - a2 - x2[v][i] * c2);
                    pbf[v][i] += d1[i];
                    pbb[v][i] += d2;
#pragma unroll
                    for (int j = 0; j < 4; ++j) pW[v][i][j] += d2 * bx[j];
                }
                store4(du + rb + c, d1);
            }
        }
    }
    extern __shared__ __attribute__((aligned(16))) float red_lds[];
    float* wsb = ws + (int64_t)blockIdx.x * 10 * H;
    block_store_partial<NV>(pbf, red_lds, wsb, H, lane, wid);
    block_store_partial<NV>(pg1, red_lds, wsb + H, H, lane, wid);
    block_store_partial<NV>(pb1, red_lds, wsb + 2 * H, H, lane, wid);
    block_store_partial<NV>(pbb, red_lds, wsb + 3 * H, H, lane, wid);
    block_store_partial<NV>(pg2, red_lds, wsb + 4 * H, H, lane, wid);
    block_store_partial<NV>(pb2, red_lds, wsb + 5 * H, H, lane, wid);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float pj[NV][4];
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int i = 0; i < 4; ++i) pj[v][i] = pW[v][i][j];
        block_store_partial<NV>(pj, red_lds, wsb + (6 + j) * H, H, lane, wid);
    }
}

// ------------------------------------------------------------------------------- column sum
// partial column sums (bias gradients): workgroup = 64 columns x 4 row lanes over a 64-row chunk
template <typename T>
__global__ __launch_bounds__(NT) void colsum_kernel(const T* __restrict__ x, float* ws, int M, int N, int64_t ld,
                                                    int rows_per_block) {
    __shared__ float red[4][64];
    const int ci = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + ci;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (col < N) {
        int r = r0 + sl;
        for (; r + 12 < r1; r += 16) {
            a0 += to_f32(x[(int64_t)r * ld + col]);
            a1 += to_f32(x[(int64_t)(r + 4) * ld + col]);
            a2 += to_f32(x[(int64_t)(r + 8) * ld + col]);
            a3 += to_f32(x[(int64_t)(r + 12) * ld + col]);
        }
        for (; r < r1; r += 4) a0 += to_f32(x[(int64_t)r * ld + col]);
    }
    red[sl][ci] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (sl == 0 && col < N) ws[(int64_t)blockIdx.y * N + col] = (red[0][ci] + red[1][ci]) + (red[2][ci] + red[3][ci]);
}

inline int rows_grid(int M, int cap) { return std::min(ceil_div(M, WPB), cap); }

#define DISPATCH_NV(H, ...)                                          \
    do {                                                             \
        const int nv_ = ceil_div(H, 256);                            \
        if (nv_ <= 1) { constexpr int NV = 1; __VA_ARGS__; }         \
        else if (nv_ <= 2) { constexpr int NV = 2; __VA_ARGS__; }    \
        else if (nv_ <= 3) { constexpr int NV = 3; __VA_ARGS__; }    \
        else if (nv_ <= 4) { constexpr int NV = 4; __VA_ARGS__; }    \
        else if (nv_ <= 6) { constexpr int NV = 6; __VA_ARGS__; }    \
        else { constexpr int NV = 8; __VA_ARGS__; }                  \
    } while (0)

inline int check_row_shape(const char* who, int M, int H) {
    XGGM_REQUIRE(M > 0 && H > 0, "%s: empty input M=%d H=%d", who, M, H);
    XGGM_REQUIRE(H % 4 == 0 && H <= 2048, "%s: H=%d must be a multiple of 4 and <= 2048", who, H);
    return XGGM_OK;
}

template <typename T>
int ln_fwd_grouped(const xggm_ln_fwd_problem* probs, int n, int H, float eps, float p_pre, float p_post, const uint64_t* rng,
                   int accumulate, float out_scale, hipStream_t st) {
    XGGM_REQUIRE(probs && n > 0, "xggm_ln_fwd: no problems");
    XGGM_REQUIRE((p_pre == 0.f && p_post == 0.f) || rng, "xggm_ln_fwd: dropout needs an rng state");
    XGGM_REQUIRE(p_pre >= 0.f && p_pre < 1.f && p_post >= 0.f && p_post < 1.f, "xggm_ln_fwd: bad dropout p");
    for (int i0 = 0; i0 < n; i0 += MAX_SEG) {
        LnFwdGroup G;
        G.n = std::min(MAX_SEG, n - i0);
        int total = 0;
        for (int i = 0; i < G.n; ++i) {
            const xggm_ln_fwd_problem& q = probs[i0 + i];
            if (int e = check_row_shape("xggm_ln_fwd", q.M, H)) return e;
            XGGM_REQUIRE(q.in && q.gamma && q.beta && q.out, "xggm_ln_fwd: null pointer");
            XGGM_REQUIRE(q.in_slabs >= 0 && q.in_slabs <= 8, "xggm_ln_fwd: in_slabs = %d (0..8)", q.in_slabs);
            XGGM_REQUIRE(q.in_slabs == 0 || (q.z_out && q.z_out != q.in), "xggm_ln_fwd: split-K input needs its own z_out");
            G.s[i] = q;
            G.start[i] = total;
            total += std::min(ceil_div(q.M, LN_FWD_W), 4096);
        }
        G.start[G.n] = total;
        const PrefetchArgs pf = xggm_take_prefetch();
        DISPATCH_NV(H, hipLaunchKernelGGL((ln_fwd_kernel<T, NV>), dim3(total + pf.blocks), dim3(LN_FWD_W * 64), 0, st, G, H, eps, p_pre,
                                           p_post, rng, accumulate, out_scale, pf));
        if (int e = xggm_check_launch("xggm_ln_fwd")) return e;
    }
    return XGGM_OK;
}

template <typename T>
int ln_sum_fwd(const xggm_ln_sum_args* x, hipStream_t st) {
    XGGM_REQUIRE(x && x->n >= 1 && x->n <= 4 && x->out, "xggm_ln_sum_fwd: 1..4 terms and an output");
    if (int e = check_row_shape("xggm_ln_sum_fwd", x->M, x->H)) return e;
    XGGM_REQUIRE(x->p_post >= 0.f && x->p_post < 1.f && (x->p_post == 0.f || x->rng), "xggm_ln_sum_fwd: bad dropout p / no rng state");
    LnSumArgs a;
    a.n = x->n;
    a.M = x->M;
    a.out = x->out;
    for (int k = 0; k < 4; ++k) {
        const bool on = k < x->n;
        XGGM_REQUIRE(!on || (x->in[k] && x->gamma[k] && x->beta[k]), "xggm_ln_sum_fwd: term %d has a null pointer", k);
        XGGM_REQUIRE(!on || x->in[k] != x->out, "xggm_ln_sum_fwd: the output may not be one of the terms");
        a.in[k] = on ? x->in[k] : nullptr;
        a.gamma[k] = on ? x->gamma[k] : nullptr;
        a.beta[k] = on ? x->beta[k] : nullptr;
        a.stats[k] = on ? x->stats[k] : nullptr;
        a.sid_post[k] = on ? x->sid_post[k] : 0u;
    }
    const int H = x->H;
    const int grid = std::min(ceil_div(x->M, LN_FWD_W), 4096);
    DISPATCH_NV(H, hipLaunchKernelGGL((ln_sum_fwd_kernel<T, NV>), dim3(grid), dim3(LN_FWD_W * 64), 0, st, a, H, x->eps, x->p_post, x->rng));
    return xggm_check_launch("xggm_ln_sum_fwd");
}

template <typename T>
int ln_fwd(const void* in, const float* bias, const void* residual, const float* gamma, const float* beta, void* out,
           void* z_out, float* stats, int M, int H, float eps, float p_pre, float p_post, const uint64_t* rng,
           uint32_t s_pre, uint32_t s_post, int accumulate, float out_scale, hipStream_t st) {
    const xggm_ln_fwd_problem q{in, bias, residual, gamma, beta, out, z_out, stats, M, s_pre, s_post, 0, nullptr, nullptr, nullptr};
    return ln_fwd_grouped<T>(&q, 1, H, eps, p_pre, p_post, rng, accumulate, out_scale, st);
}

inline size_t bwd_ws_bytes(int M, int H, int K) { return sizeof(float) * (size_t)rows_grid(M, 512) * K * H; }
// workgroups of a LayerNorm backward = partial [3, H] rows its reduce job reads: at most 256 (one per CU).  With 512 the
// launches of >= 2048 rows (the reference batch of 92, the 64 x 64 stress configuration) left twice the partial rows for
// the batched reduction: 256 measured -0.15 ms per iteration at 92 samples and -20 us per C4 pass, nothing at 32 samples
// (224 workgroups either way); 128 / 192 no better (tools/exp_ln_bwd_grid.sh, XGGM_LN_BWD_GRID overrides)
inline int ln_bwd_grid(int M) {
    static const int cap = getenv("XGGM_LN_BWD_GRID") ? std::max(1, atoi(getenv("XGGM_LN_BWD_GRID"))) : 256;
    return std::min(ceil_div(M, LN_BWD_W), cap);
}
inline size_t ln_bwd_ws_bytes(int M, int H) { return sizeof(float) * (size_t)ln_bwd_grid(M) * 3 * H; }

inline void launch_reduce(const float* ws, int nblk, int K, int H, const ReduceTargets& tg, hipStream_t st) {
    hipLaunchKernelGGL(partial_reduce_kernel, dim3(ceil_div(K * H, 64)), dim3(NT), 0, st, ws, nblk, K, H, tg);
}

template <typename T>
int ln_bwd_grouped(const xggm_ln_bwd_problem* probs, int n, int H, float p_pre, float p_post, const uint64_t* rng,
                   float out_scale, hipStream_t st) {
    XGGM_REQUIRE(probs && n > 0, "xggm_ln_bwd: no problems");
    XGGM_REQUIRE((p_pre == 0.f && p_post == 0.f) || rng, "xggm_ln_bwd: dropout needs an rng state");
    for (int i0 = 0; i0 < n; i0 += MAX_SEG) {
        LnBwdGroup G;
        G.n = std::min(MAX_SEG, n - i0);
        int total = 0;
        for (int i = 0; i < G.n; ++i) {
            const xggm_ln_bwd_problem& q = probs[i0 + i];
            if (int e = check_row_shape("xggm_ln_bwd", q.M, H)) return e;
            XGGM_REQUIRE(q.dy && q.z && q.stats && q.gamma, "xggm_ln_bwd: null pointer");
            XGGM_REQUIRE(q.ws && q.ws_bytes >= ln_bwd_ws_bytes(q.M, H), "xggm_ln_bwd: workspace of %zu bytes needed, got %zu",
                         ln_bwd_ws_bytes(q.M, H), (size_t)q.ws_bytes);
            G.s[i] = q;
            G.start[i] = total;
            total += ln_bwd_grid(q.M);
        }
        G.start[G.n] = total;
        DISPATCH_NV(H, {
            static bool big_lds = false;  // rows of up to 2048 floats x 8 waves exceed the 48 KB default
            if (!big_lds) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ln_bwd_kernel<T, NV>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)std::min<size_t>(sizeof(float) * LN_BWD_W * 256 * NV * 3, 144 * 1024));
                big_lds = true;
            }
        });
        const PrefetchArgs pf = xggm_take_prefetch();
        DISPATCH_NV(H, hipLaunchKernelGGL((ln_bwd_kernel<T, NV>), dim3(total + pf.blocks), dim3(LN_BWD_W * 64), sizeof(float) * LN_BWD_W * H * (ln_bwd_fused_tail(H) ? 3 : 1), st, G, H,
                                           p_pre, p_post, rng, out_scale, pf));
        if (int e = xggm_check_launch("xggm_ln_bwd")) return e;
        for (int i = 0; i < G.n; ++i) {
            const xggm_ln_bwd_problem& q = G.s[i];
            if (q.dgamma || q.dbeta || q.dbias) {
                ReduceTargets tg{};
                tg.t[0] = q.dgamma; tg.t[1] = q.dbeta; tg.t[2] = q.dbias;
                tg.stride[0] = tg.stride[1] = tg.stride[2] = 1;
                launch_reduce(q.ws, G.start[i + 1] - G.start[i], 3, H, tg, st);
                if (int e = xggm_check_launch("xggm_ln_bwd(reduce)")) return e;
            }
        }
    }
    return XGGM_OK;
}

template <typename T>
int ln_bwd(const void* dy, const void* z, const float* stats, const float* gamma, void* d_in, void* d_res, float* dgamma,
           float* dbeta, float* dbias, int M, int H, float p_pre, float p_post, const uint64_t* rng, uint32_t s_pre,
           uint32_t s_post, float out_scale, int accumulate_dres, const void* gelu_aux, float* ws, size_t ws_bytes,
           hipStream_t st) {
    const xggm_ln_bwd_problem q{dy, z, stats, gamma, d_in, d_res, dgamma, dbeta, dbias, gelu_aux, ws, ws_bytes, M, s_pre, s_post,
                                accumulate_dres};
    return ln_bwd_grouped<T>(&q, 1, H, p_pre, p_post, rng, out_scale, st);
}

template <typename T>
int embed_fwd(const int64_t* ids, const int64_t* seg, const void* word, const void* pos, const void* type,
              const float* gamma, const float* beta, void* out, void* z_out, float* stats, int M, int Tlen, int H,
              float eps, float p, const uint64_t* rng, uint32_t sid, void* out8, const float* qscale, float* amax,
              int amax_slots, const xggm_side_jobs* side, hipStream_t st) {
    if (int e = check_row_shape("xggm_embed_fwd", M, H)) return e;
    SideJobs sj;
    sj.n = 0;
    sj.main_blocks = rows_grid(M, 4096);
    int side_blocks = 0;
    if (side) {
        XGGM_REQUIRE(side->n >= 0 && side->n <= XGGM_SIDE_MAX, "xggm_embed_fwd_side: %d side jobs (at most %d)", side->n, XGGM_SIDE_MAX);
        for (int k = 0; k < side->n; ++k) {
            const auto& j = side->job[k];
            XGGM_REQUIRE((j.kind == XGGM_SIDE_ADDITIVE_MASK || j.kind == XGGM_SIDE_CAST_BF16) && j.src && j.dst && j.count > 0,
                         "xggm_embed_fwd_side: bad side job %d (kind %d, %lld elements)", k, j.kind, (long long)j.count);
            sj.kind[k] = j.kind;
            sj.src[k] = j.src;
            sj.dst[k] = j.dst;
            sj.count[k] = j.count;
            sj.blk0[k] = side_blocks;
            side_blocks += (int)std::min<int64_t>(ceil_div64(j.count, (int64_t)NT * 8), 256);
        }
        sj.n = side->n;
        sj.blk0[sj.n] = side_blocks;
    }
    XGGM_REQUIRE(!out8 || (sizeof(T) == 2 && reinterpret_cast<uintptr_t>(out8) % 4 == 0),
                 "xggm_embed_fwd: the e4m3 output copy needs bf16 storage and a 4-byte aligned buffer");
    const Emit8 e8{reinterpret_cast<unsigned char*>(out8), qscale, amax, amax_slots > 1 ? amax_slots : 1};
    XGGM_REQUIRE(ids && word && pos && type && gamma && beta && out && z_out && stats, "xggm_embed_fwd: null pointer");
    XGGM_REQUIRE(Tlen > 0 && M % Tlen == 0, "xggm_embed_fwd: M=%d is not a multiple of T=%d", M, Tlen);
    XGGM_REQUIRE(p == 0.f || rng, "xggm_embed_fwd: dropout needs an rng state");
    DropArgs d{0.f, p, rng, 0, sid};
    DISPATCH_NV(H, hipLaunchKernelGGL((embed_fwd_kernel<T, NV>), dim3(sj.main_blocks + side_blocks), dim3(NT), 0, st, ids, seg,
                                       (const T*)word, (const T*)pos, (const T*)type, gamma, beta, (T*)out, (T*)z_out, stats,
                                       M, Tlen, H, eps, d, e8, sj));
    return xggm_check_launch("xggm_embed_fwd");
}

template <typename T>
int embed_bwd(const int64_t* ids, const int64_t* seg, const void* dy, const void* z, const float* stats, const float* gamma,
              void* dz_ws, float* dword, float* dpos, float* dtype, float* dgamma, float* dbeta, int M, int Tlen, int H,
              float p, const uint64_t* rng, uint32_t sid, float* ws, size_t ws_bytes, RowList rl, hipStream_t st) {
    if (int e = check_row_shape("xggm_embed_bwd", M, H)) return e;
    XGGM_REQUIRE((rl.ids != nullptr) == (rl.sq != nullptr) && (rl.ids != nullptr) == (rl.n != nullptr),
                 "xggm_embed_bwd_listed: the row list is ids + squares + count, or nothing");
    XGGM_REQUIRE(ids && dy && z && stats && gamma && dz_ws && dword && dpos && dtype, "xggm_embed_bwd: null pointer");
    if (int e = ln_bwd<T>(dy, z, stats, gamma, dz_ws, nullptr, dgamma, dbeta, nullptr, M, H, 0.f, p, rng, 0, sid, 1.0f, 0,
                          nullptr, ws, ws_bytes, st))
        return e;
    XGGM_REQUIRE(Tlen > 0 && M % Tlen == 0, "xggm_embed_bwd: M=%d is not a multiple of T=%d", M, Tlen);
    const int tasks = M + (seg ? M : 0) + Tlen;  // one wave per candidate owner, see embed_scatter_kernel
    DISPATCH_NV(H, hipLaunchKernelGGL((embed_scatter_kernel<T, NV>), dim3(ceil_div(tasks, WPB)), dim3(NT), 0, st, ids, seg,
                                       (const T*)dz_ws, dword, dpos, dtype, M, Tlen, H, rl));
    return xggm_check_launch("xggm_embed_bwd(scatter)");
}

template <typename T>
int visn_fwd(const void* u, const float* bf, const void* boxes, const float* Wb, const float* bb, const float* g1,
             const float* b1, const float* g2, const float* b2, void* out, void* z1, void* z2, float* stats, int M, int H,
             float eps, float p, const uint64_t* rng, uint32_t sid, void* out8, const float* qscale, float* amax,
             int amax_slots, hipStream_t st) {
    if (int e = check_row_shape("xggm_visn_embed_fwd", M, H)) return e;
    XGGM_REQUIRE(!out8 || (sizeof(T) == 2 && reinterpret_cast<uintptr_t>(out8) % 4 == 0),
                 "xggm_visn_embed_fwd: the e4m3 output copy needs bf16 storage and a 4-byte aligned buffer");
    const Emit8 e8{reinterpret_cast<unsigned char*>(out8), qscale, amax, amax_slots > 1 ? amax_slots : 1};
    XGGM_REQUIRE(u && bf && boxes && Wb && bb && g1 && b1 && g2 && b2 && out && z1 && z2 && stats,
                 "xggm_visn_embed_fwd: null pointer");
    XGGM_REQUIRE(p == 0.f || rng, "xggm_visn_embed_fwd: dropout needs an rng state");
    DropArgs d{0.f, p, rng, 0, sid};
    DISPATCH_NV(H, hipLaunchKernelGGL((visn_embed_fwd_kernel<T, NV>), dim3(rows_grid(M, 4096)), dim3(NT), 0, st,
                                       (const T*)u, bf, (const T*)boxes, Wb, bb, g1, b1, g2, b2, (T*)out, (T*)z1, (T*)z2,
                                       stats, M, H, eps, d, e8));
    return xggm_check_launch("xggm_visn_embed_fwd");
}

template <typename T>
int visn_bwd(const void* dy, const void* z1, const void* z2, const float* stats, const void* boxes, const float* g1,
             const float* g2, void* du, float* dbf, float* dg1, float* db1, float* dWb, float* dbb, float* dg2, float* db2,
             int M, int H, float p, const uint64_t* rng, uint32_t sid, float* ws, size_t ws_bytes, hipStream_t st) {
    if (int e = check_row_shape("xggm_visn_embed_bwd", M, H)) return e;
    XGGM_REQUIRE(dy && z1 && z2 && stats && boxes && g1 && g2 && du && dbf && dg1 && db1 && dWb && dbb && dg2 && db2,
                 "xggm_visn_embed_bwd: null pointer");
    XGGM_REQUIRE(H <= 1024, "xggm_visn_embed_bwd: H=%d > 1024", H);
    XGGM_REQUIRE(ws && ws_bytes >= bwd_ws_bytes(M, H, 10), "xggm_visn_embed_bwd: workspace of %zu bytes needed, got %zu",
                 bwd_ws_bytes(M, H, 10), ws_bytes);
    DropArgs d{0.f, p, rng, 0, sid};
    const int nv_ = ceil_div(H, 256);
    const int grid = rows_grid(M, 512);
    const size_t lds = sizeof(float) * WPB * H;
    if (nv_ <= 1) {
        hipLaunchKernelGGL((visn_embed_bwd_kernel<T, 1>), dim3(grid), dim3(NT), lds, st, (const T*)dy, (const T*)z1,
                           (const T*)z2, stats, (const T*)boxes, g1, g2, (T*)du, ws, M, H, d);
    } else if (nv_ <= 3) {
        hipLaunchKernelGGL((visn_embed_bwd_kernel<T, 3>), dim3(grid), dim3(NT), lds, st, (const T*)dy, (const T*)z1,
                           (const T*)z2, stats, (const T*)boxes, g1, g2, (T*)du, ws, M, H, d);
    } else {
        hipLaunchKernelGGL((visn_embed_bwd_kernel<T, 4>), dim3(grid), dim3(NT), lds, st, (const T*)dy, (const T*)z1,
                           (const T*)z2, stats, (const T*)boxes, g1, g2, (T*)du, ws, M, H, d);
    }
    if (int e = xggm_check_launch("xggm_visn_embed_bwd")) return e;
    ReduceTargets tg{};
    float* t[6] = {dbf, dg1, db1, dbb, dg2, db2};
    for (int k = 0; k < 6; ++k) { tg.t[k] = t[k]; tg.stride[k] = 1; }
    for (int j = 0; j < 4; ++j) { tg.t[6 + j] = dWb + j; tg.stride[6 + j] = 4; }
    launch_reduce(ws, grid, 10, H, tg, st);
    return xggm_check_launch("xggm_visn_embed_bwd(reduce)");
}

constexpr int CS_ROWS = 64;
template <typename T> int colsum(const void* x, float* out, int M, int N, int64_t ld, float* ws, size_t ws_bytes,
                                 hipStream_t st) {
    XGGM_REQUIRE(x && out && M > 0 && N > 0 && ld >= N, "xggm_colsum: bad arguments M=%d N=%d", M, N);
    const int chunks = ceil_div(M, CS_ROWS);
    XGGM_REQUIRE(ws && ws_bytes >= sizeof(float) * (size_t)chunks * N, "xggm_colsum: workspace of %zu bytes needed, got %zu",
                 sizeof(float) * (size_t)chunks * N, ws_bytes);
    hipLaunchKernelGGL((colsum_kernel<T>), dim3(ceil_div(N, 64), chunks), dim3(NT), 0, st, (const T*)x, ws, M, N, ld,
                       CS_ROWS);
    if (int e = xggm_check_launch("xggm_colsum")) return e;
    ReduceTargets tg{};
    tg.t[0] = out;
    tg.stride[0] = 1;
    launch_reduce(ws, chunks, 1, N, tg, st);
    return xggm_check_launch("xggm_colsum(reduce)");
}

size_t ws_ln(int M, int H) { return ln_bwd_ws_bytes(M, H); }
size_t ws_visn(int M, int H) { return bwd_ws_bytes(M, H, 10); }
size_t ws_colsum(int M, int N) { return sizeof(float) * (size_t)ceil_div(M, CS_ROWS) * N; }
}  // namespace

#define ROW_API(SUF, T)                                                                                                     \
    extern "C" int xggm_ln_fwd_##SUF(const void* in, const float* bias, const void* residual, const float* gamma,          \
                                     const float* beta, void* out, void* z_out, float* stats, int M, int H, float eps,     \
                                     float p_pre, float p_post, const uint64_t* rng, uint32_t s_pre, uint32_t s_post,      \
                                     int accumulate, float out_scale, hipStream_t st) {                                   \
        return ln_fwd<T>(in, bias, residual, gamma, beta, out, z_out, stats, M, H, eps, p_pre, p_post, rng, s_pre, s_post, \
                         accumulate, out_scale, st);                                                                       \
    }                                                                                                                       \
    extern "C" int xggm_ln_bwd_##SUF(const void* dy, const void* z, const float* stats, const float* gamma, void* d_in,    \
                                     void* d_res, float* dgamma, float* dbeta, float* dbias, int M, int H, float p_pre,    \
                                     float p_post, const uint64_t* rng, uint32_t s_pre, uint32_t s_post, float out_scale,  \
                                     int accumulate_dres, const void* gelu_aux, float* ws, size_t ws_bytes,               \
                                     hipStream_t st) {                                                                    \
        return ln_bwd<T>(dy, z, stats, gamma, d_in, d_res, dgamma, dbeta, dbias, M, H, p_pre, p_post, rng, s_pre, s_post,  \
                         out_scale, accumulate_dres, gelu_aux, ws, ws_bytes, st);                                          \
    }                                                                                                                       \
    extern "C" int xggm_ln_fwd_grouped_##SUF(const xggm_ln_fwd_problem* probs, int n, int H, float eps, float p_pre,       \
                                             float p_post, const uint64_t* rng, int accumulate, float out_scale,          \
                                             hipStream_t st) {                                                            \
        return ln_fwd_grouped<T>(probs, n, H, eps, p_pre, p_post, rng, accumulate, out_scale, st);                         \
    }                                                                                                                       \
    extern "C" int xggm_ln_bwd_grouped_##SUF(const xggm_ln_bwd_problem* probs, int n, int H, float p_pre, float p_post,    \
                                             const uint64_t* rng, float out_scale, hipStream_t st) {                      \
        return ln_bwd_grouped<T>(probs, n, H, p_pre, p_post, rng, out_scale, st);                                          \
    }                                                                                                                       \
    extern "C" int xggm_ln_sum_fwd_##SUF(const xggm_ln_sum_args* args, hipStream_t st) { return ln_sum_fwd<T>(args, st); }  \
    extern "C" int xggm_embed_fwd_##SUF(const int64_t* ids, const int64_t* seg, const void* word, const void* pos,         \
                                        const void* type, const float* gamma, const float* beta, void* out, void* z_out,   \
                                        float* stats, int M, int Tlen, int H, float eps, float p, const uint64_t* rng,    \
                                        uint32_t sid, void* out8, const float* qscale, float* amax, int amax_slots,       \
                                        hipStream_t st) {                                                                 \
        return embed_fwd<T>(ids, seg, word, pos, type, gamma, beta, out, z_out, stats, M, Tlen, H, eps, p, rng, sid, out8, \
                            qscale, amax, amax_slots, nullptr, st);                                                        \
    }                                                                                                                       \
    extern "C" int xggm_embed_fwd_side_##SUF(const int64_t* ids, const int64_t* seg, const void* word, const void* pos,    \
                                             const void* type, const float* gamma, const float* beta, void* out,           \
                                             void* z_out, float* stats, int M, int Tlen, int H, float eps, float p,         \
                                             const uint64_t* rng, uint32_t sid, void* out8, const float* qscale,           \
                                             float* amax, int amax_slots, const xggm_side_jobs* side, hipStream_t st) {    \
        return embed_fwd<T>(ids, seg, word, pos, type, gamma, beta, out, z_out, stats, M, Tlen, H, eps, p, rng, sid, out8, \
                            qscale, amax, amax_slots, side, st);                                                           \
    }                                                                                                                       \
    extern "C" int xggm_embed_bwd_##SUF(const int64_t* ids, const int64_t* seg, const void* dy, const void* z,             \
                                        const float* stats, const float* gamma, void* dz_ws, float* dword, float* dpos,    \
                                        float* dtype, float* dgamma, float* dbeta, int M, int Tlen, int H, float p,        \
                                        const uint64_t* rng, uint32_t sid, float* ws, size_t ws_bytes, hipStream_t st) {  \
        return embed_bwd<T>(ids, seg, dy, z, stats, gamma, dz_ws, dword, dpos, dtype, dgamma, dbeta, M, Tlen, H, p, rng,   \
                            sid, ws, ws_bytes, RowList{nullptr, nullptr, nullptr}, st);                                    \
    }                                                                                                                       \
    extern "C" int xggm_embed_bwd_listed_##SUF(const int64_t* ids, const int64_t* seg, const void* dy, const void* z,      \
                                               const float* stats, const float* gamma, void* dz_ws, float* dword,          \
                                               float* dpos, float* dtype, float* dgamma, float* dbeta, int M, int Tlen,    \
                                               int H, float p, const uint64_t* rng, uint32_t sid, float* ws,              \
                                               size_t ws_bytes, int64_t* row_ids, float* row_sq, int* row_n,               \
                                               hipStream_t st) {                                                           \
        return embed_bwd<T>(ids, seg, dy, z, stats, gamma, dz_ws, dword, dpos, dtype, dgamma, dbeta, M, Tlen, H, p, rng,   \
                            sid, ws, ws_bytes, RowList{row_ids, row_sq, row_n}, st);                                       \
    }                                                                                                                       \
    extern "C" int xggm_visn_embed_fwd_##SUF(const void* u, const float* bf, const void* boxes, const float* Wb,           \
                                             const float* bb, const float* g1, const float* b1, const float* g2,           \
                                             const float* b2, void* out, void* z1, void* z2, float* stats, int M, int H,   \
                                             float eps, float p, const uint64_t* rng, uint32_t sid, void* out8,           \
                                             const float* qscale, float* amax, int amax_slots, hipStream_t st) {          \
        return visn_fwd<T>(u, bf, boxes, Wb, bb, g1, b1, g2, b2, out, z1, z2, stats, M, H, eps, p, rng, sid, out8, qscale,   \
                           amax, amax_slots, st);                                                                          \
    }                                                                                                                       \
    extern "C" int xggm_visn_embed_bwd_##SUF(const void* dy, const void* z1, const void* z2, const float* stats,           \
                                             const void* boxes, const float* g1, const float* g2, void* du, float* dbf,    \
                                             float* dg1, float* db1, float* dWb, float* dbb, float* dg2, float* db2, int M,\
                                             int H, float p, const uint64_t* rng, uint32_t sid, float* ws,                \
                                             size_t ws_bytes, hipStream_t st) {                                           \
        return visn_bwd<T>(dy, z1, z2, stats, boxes, g1, g2, du, dbf, dg1, db1, dWb, dbb, dg2, db2, M, H, p, rng, sid, ws, \
                           ws_bytes, st);                                                                                  \
    }                                                                                                                       \
    extern "C" int xggm_colsum_##SUF(const void* x, float* out, int M, int N, int64_t ld, float* ws, size_t ws_bytes,      \
                                     hipStream_t st) {                                                                    \
        return colsum<T>(x, out, M, N, ld, ws, ws_bytes, st);                                                              \
    }

ROW_API(f32, float)
ROW_API(bf16, bf16)

extern "C" int xggm_partial_reduce_batch(const xggm_reduce_job* jobs, int n, hipStream_t stream) {
    XGGM_REQUIRE(jobs && n > 0, "xggm_partial_reduce_batch: no jobs");
    for (int i0 = 0; i0 < n; i0 += MAX_JOBS) {
        BatchJobs bj;
        bj.n = std::min(MAX_JOBS, n - i0);
        int total = 0;
        for (int i = 0; i < bj.n; ++i) {
            const xggm_reduce_job& j = jobs[i0 + i];
            XGGM_REQUIRE(j.ws && j.nblk > 0 && j.K > 0 && j.K <= 3 && j.H > 0, "xggm_partial_reduce_batch: bad job %d", i0 + i);
            bj.j[i] = j;
            bj.start[i] = total;
            XGGM_REQUIRE(j.H % 4 == 0 && reinterpret_cast<uintptr_t>(j.ws) % 16 == 0,
                         "xggm_partial_reduce_batch: job %d needs H %% 4 == 0 and a 16-byte aligned workspace", i0 + i);
            total += ceil_div(j.K * j.H, RB_COLS);
        }
        bj.start[bj.n] = total;
        hipLaunchKernelGGL(partial_reduce_batch_kernel, dim3(total), dim3(NT), 0, stream, bj);
        if (int e = xggm_check_launch("xggm_partial_reduce_batch")) return e;
    }
    return XGGM_OK;
}

// workspace sizes (bytes) of the backward row kernels: K partial rows per workgroup
extern "C" size_t xggm_ln_bwd_workspace_bytes(int M, int H) { return ws_ln(M, H); }
extern "C" size_t xggm_visn_embed_bwd_workspace_bytes(int M, int H) { return ws_visn(M, H); }
extern "C" size_t xggm_colsum_workspace_bytes(int M, int N) { return ws_colsum(M, N); }
