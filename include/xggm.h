/* xggm.h -- C ABI of the MI355X-native X-GGM training-step hot path (libxggm_hip.so).
 *
 * The reference (jingjing12110/X-GGM) is pure PyTorch and has no native interface; this
 * header is the boundary SURVEY.md section 8(b) specifies: one extern "C" entry point per
 * fused op and direction.  Each declaration cites the reference call site it replaces
 * (paths relative to /root/reference).
 *
 * Conventions
 *  - Every pointer is a DEVICE pointer unless said otherwise; tensors are row-major and
 *    contiguous except where a stride argument exists.  The caller owns all memory,
 *    including workspaces; nothing here allocates, frees or synchronises -- work is only
 *    enqueued on `stream` (graph-capture safe).
 *  - `_f32` / `_bf16` in a symbol name is the STORAGE type "T" of activations and of the
 *    GEMM operands.  Arithmetic is fp32 (bf16 MFMA accumulates in fp32).  Parameters that
 *    are vectors (biases, LayerNorm gains), parameter gradients, adjacency-shaped tensors
 *    [B,N,N], losses and optimiser state are always fp32.
 *  - Return value: 0 = ok; non-zero = error, message via xggm_last_error() (thread-local).
 *    Reentrant and thread-safe: called from the host thread in forward and from the
 *    autograd thread in backward.
 *  - Randomness: `rng` points to two device uint64 {seed, offset}; dropout masks and
 *    Gaussian draws are pure functions of (seed, offset, stream id `sid`, element index), so
 *    backward regenerates the forward mask.  xggm_rng_advance() bumps the offset once per
 *    pass.  p == 0 (or a supplied `randn`) ignores `rng`.
 */
#ifndef XGGM_H
#define XGGM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* xggm_stream_t; /* == hipStream_t */

#define XGGM_VERSION 100

/* activation codes of xggm_gemm_* */
#define XGGM_ACT_NONE 0
#define XGGM_ACT_GELU 1      /* erf-GELU, src/lxrt/modeling.py:116-124 */
#define XGGM_ACT_SIGMOID 2   /* encoder_adj, src/vqa/vqacpv2_model.py:91-94 */
#define XGGM_ACT_TANH 3      /* BertPooler, src/lxrt/modeling.py:614-620 */
#define XGGM_ACT_GELU_GRAD 4 /* C = acc * gelu'(aux): backward of BertIntermediate fused into dgrad */

/* matrix modes of xggm_aggregate_* */
#define XGGM_AGG_PLAIN 0
#define XGGM_AGG_TRANSPOSE 1
#define XGGM_AGG_SYMMETRIZE 2

int xggm_version(void);
const char* xggm_last_error(void);
/* HOST: queue a byte range (16-byte aligned; at most four per launch) that the NEXT xggm_ln_fwd_grouped_* /
 * xggm_ln_bwd_grouped_* / bf16 xggm_attn_fwd_grouped / xggm_attn_bwd_grouped launch reads beside its own work and
 * discards: the weights of the Linear products that follow (src/lxrt/modeling.py:344-347, 384-388, 428-445) are then in
 * the Infinity Cache when those products ask for them.  Speed only -- nothing is written, no result depends on it;
 * XGGM_PREFETCH=0 in the environment makes this a no-op. */
int xggm_prefetch_next(const void* ptr, size_t bytes);

/* ---- dense products --------------------------------------------------------------------
 * C[z][m][n] = epilogue( alpha * sum_k A(z,m,k) * B(z,k,n) ),
 *   A(z,m,k) = A[z*a_bs + m*a_rs + k*a_ks],  B(z,k,n) = B[z*b_bs + n*b_ns + k*b_ks],
 *   C at C[z*c_bs + m*ldc + n].
 * epilogue: v = alpha*acc + bias[n]; preact (if given) <- v; v = act(v) (GELU_GRAD: v *
 * gelu'(aux)); v += residual; C = accumulate ? C + v : v; C is T, or float when c_f32.
 * colsum (or NULL; bias gradients): fp32 [batch][ceil(M/32)][N] PARTIAL column sums of v -- row r of batch z holds
 * sum over output rows 32r .. 32r+31 of v[z][m][n]; every element is written exactly once by exactly one workgroup,
 * in a fixed summation order (no floating-point atomics anywhere on the path: a pass gives the same bits whatever the
 * scheduling).  The caller adds the partial rows into the gradient with xggm_partial_reduce_batch (K = 1, H = N,
 * nblk = batch * ceil(M/32)).
 * Replaces nn.Linear forward/backward (src/lxrt/modeling.py:345-347, 385, 429, 442, 617;
 * src/module/gcn.py:28; src/vqa/vqacpv2_model.py:63-105) and torch.bmm(x, x^T)
 * (src/module/graph_generative_modeling.py:225). */
int xggm_gemm_f32(const void* A, const void* B, void* C, int M, int N, int K, int64_t a_rs, int64_t a_ks, int64_t b_ns,
                  int64_t b_ks, int64_t ldc, int batch, int64_t a_bs, int64_t b_bs, int64_t c_bs, const float* bias,
                  const void* residual, void* preact, const void* aux, float* colsum, int act, int c_f32, int accumulate,
                  float alpha, xggm_stream_t stream);
int xggm_gemm_bf16(const void* A, const void* B, void* C, int M, int N, int K, int64_t a_rs, int64_t a_ks, int64_t b_ns,
                   int64_t b_ks, int64_t ldc, int batch, int64_t a_bs, int64_t b_bs, int64_t c_bs, const float* bias,
                   const void* residual, void* preact, const void* aux, float* colsum, int act, int c_f32, int accumulate,
                   float alpha, xggm_stream_t stream);
/* Up to 4 independent products in ONE launch (forward of both modalities, dgrad + wgrad of a layer):
 * skinny problems that cannot fill 256 CUs alone share a grid.  `probs` is a HOST array; fields as
 * the arguments of xggm_gemm_*.  Falls back to one launch per problem for shapes the tuned kernel
 * does not take (unaligned strides, fp32).  A dimension that is not a multiple of 8 (2274 answers, 630 edges)
 * stays on the tuned path when the operand's leading stride is padded to a multiple of 8 (whatever the
 * padding holds). */
typedef struct xggm_gemm_problem {
    const void* A;
    const void* B;
    void* C;
    int M, N, K;
    int64_t a_rs, a_ks, b_ns, b_ks, ldc;
    int batch;
    int64_t a_bs, b_bs, c_bs;
    const float* bias;
    const void* residual;
    void* preact;
    const void* aux;
    float* colsum;
    int act, c_f32, accumulate;
    float alpha;
    /* or NULL.  fp32 outputs on the tuned bf16 kernels only: slot [batch][ceil(M/64)][ceil(N/64)] receives the sum of
     * squares of the values this launch STORED in that 64 x 64 block of C (after `accumulate`); every slot is written
     * by one workgroup, so a fixed-order sum of the slots is the squared norm of the weight gradient without another
     * pass over it (nn.utils.clip_grad_norm_, src/vqa/vqacpv2.py:175). */
    float* sqsum;
    /* e4m3 side of the mixed-precision forward (xggm_gemm_grouped_fp8e4m3 reads scale_a / scale_b; every bf16-output
     * kernel honours c8): scale_a, scale_b: device scalars, the reciprocals of the per-tensor quantisation scales of
     * A and B (NULL = 1).  c8 (or NULL): a second copy of the stored result as OCP e4m3 bytes, layout of C, each
     * value multiplied by *c8_qscale (NULL = 1) and saturated to +-448 -- the operand of the NEXT forward product,
     * written by its producer instead of by a quantisation pass; *c8_amax (or NULL) is raised to max |value| (one
     * atomic per workgroup; feeds the delayed scale update, xggm_fp8_scale_update). */
    const float* scale_a;
    const float* scale_b;
    void* c8;
    const float* c8_qscale;
    float* c8_amax;
    /* amax entries may be spread over `amax_slots` floats (a power of two <= 64; 0 reads as 1): workgroup w raises
     * c8_amax[w % amax_slots] and xggm_fp8_scale_update takes the maximum over the slots.  Hundreds of workgroups
     * raising ONE address are same-address device-scope atomics the kernel end waits for (measured: +2 us per LayerNorm
     * launch, +3.9 us per attention launch, 0.25 ms per iteration of the fp8 step). */
    int amax_slots;
} xggm_gemm_problem;
int xggm_gemm_grouped_f32(const xggm_gemm_problem* probs, int n, xggm_stream_t stream);
int xggm_gemm_grouped_bf16(const xggm_gemm_problem* probs, int n, xggm_stream_t stream);
/* fp8 forward product of the mixed-precision configuration (BASELINE.json configs[4]: "fp8 MFMA path for LXMERT
 * QKV/FFN GEMMs"; the Linear layers of src/lxrt/modeling.py:345-347 (query/key/value), :429 (intermediate),
 * :442 (output) -- the reference itself is fp32 only, `--fp16` is unused):
 *   C[m,n] = act(scale_a * scale_b * sum_k A(m,k) B(n,k) + bias[n]) (+ residual[m,n])
 * A [M, K] and B [N, K] hold OCP e4m3fn bytes (k contiguous; row strides a_rs / b_ns in elements, multiples of
 * 16, like K; 16-byte aligned bases); scale_a / scale_b: device scalars (null = 1), the reciprocals of the
 * per-tensor quantisation scales; fp32 accumulation; bias fp32 or null; C / residual / preact bf16 with row
 * stride ldc (c_f32 != 0: C fp32); act: XGGM_ACT_NONE .. XGGM_ACT_TANH as in xggm_gemm_bf16. */
int xggm_gemm_fp8e4m3(const void* A, const void* B, void* C, int M, int N, int K, int64_t a_rs, int64_t b_ns, int64_t ldc,
                      const float* scale_a, const float* scale_b, const float* bias, const void* residual, void* preact,
                      int act, int c_f32, xggm_stream_t stream);
/* The same product for up to 6 problems in one launch (the language and the vision stream of a layer, the two
 * directions of a cross-attention layer): every problem has e4m3 operands A [M, K] / B [N, K], k contiguous (a_ks =
 * b_ks = 1, strides in elements = bytes, multiples of 16 like K), its own scale_a / scale_b, and the bf16 epilogue of
 * xggm_gemm_grouped_bf16 (bias, GELU + pre-activation, residual, fp32 split-K slabs through batch / c_f32, c8).  Not
 * for the backward: dgrad / wgrad stay bf16. */
int xggm_gemm_grouped_fp8e4m3(const xggm_gemm_problem* probs, int n, xggm_stream_t stream);
/* Delayed per-tensor scaling of the e4m3 operands: one table (amax, history, qscale, dscale = 1 / qscale) whose
 * entries are weight operands and activation sites.  Protocol of the producers (LayerNorm / GELU epilogue /
 * attention output / BertAdam): an entry with qscale <= 0 is uncalibrated -- they quantise with 1 and record every
 * maximum; otherwise they quantise with qscale and record the maximum of the step in amax only when it exceeds half
 * (BertAdam: three quarters) of the representable range 448 / qscale: e4m3 is a floating-point format, so a stale
 * SMALLER maximum costs no precision and only a larger one saturates, and the same-address atomics stay rare.
 * This call, for the n entries the pointers address (amax, qscale, dscale: n floats; hist: n x hist_len):
 *   hist[i][*pos % hist_len] = amax[i]; amax[i] = 0; m = max_j hist[i][j];
 *   m > 0: qscale[i] = 448 / (margin * m)          (range = margin x the largest recent maximum)
 *   m == 0, shrink != 0, history just wrapped: qscale[i] *= 2   (nothing near the range for hist_len calls)
 *   dscale[i] = 1 / qscale[i];   bump != 0: *pos += 1.
 * Activation sites: once per pass, margin 1.25, shrink, bump.  Weight operands: for the parameter groups a pass
 * updates, BEFORE xggm_bertadam_ex writes their e4m3 copies with the new scale (margin 4/3, no shrink, no bump).
 * Everything is device-resident (graph replay); n <= 1024.  `amax_slots` (power of two <= 64, 0 = 1): entry i's maximum
 * is the largest of amax[i * amax_slots .. + amax_slots - 1] (all cleared). */
int xggm_fp8_scale_update(float* amax, float* hist, float* qscale, float* dscale, int64_t* pos, int n, int hist_len,
                          float margin, int shrink, int bump, int amax_slots, xggm_stream_t stream);
/* y[i] = e4m3fn(clamp(x[i] * *qscale, -448, 448)), round-to-nearest-even; x fp32 / bf16, n % 8 == 0; qscale: device
 * scalar or null (1); amax: device scalar or null, raised to max |x| (atomic; the caller zeroes it): the next
 * step's scale without another pass over x. */
int xggm_quantize_fp8e4m3_f32(const void* x, void* y, int64_t n, const float* qscale, float* amax, xggm_stream_t stream);
int xggm_quantize_fp8e4m3_bf16(const void* x, void* y, int64_t n, const float* qscale, float* amax, xggm_stream_t stream);
/* HOST: tile of grouped launches (0 heuristic, 1: 64x64, 2: 128x64, 3: 128x128, 4: 128x128 on 8 waves) */
int xggm_gemm_set_group_tile(int v);
/* HOST: 1 = run bf16 GEMMs on the generic 64x64 kernel, 0 = tuned kernels (default); A/B tests */
int xggm_gemm_set_generic(int on);
/* HOST: pin the tuned bf16 kernel variant of single launches (1: 64x64 depth 2, 2: 64x64 depth 4,
 * 3: 128x64 depth 2, 5: 128x128 depth 2, 6/7/8: 64x64 / 128x64 / 128x128 depth 1; 0: heuristic;
 * | 0x100 turns the XCD-aware tile order off) */
int xggm_gemm_set_tile(int variant);
/* out[n] += sum_m x[m*ld + n]  (bias gradients; `out` must hold the running value) */
int xggm_colsum_f32(const void* x, float* out, int M, int N, int64_t ld, float* ws, size_t ws_bytes,
                    xggm_stream_t stream);
int xggm_colsum_bf16(const void* x, float* out, int M, int N, int64_t ld, float* ws, size_t ws_bytes,
                     xggm_stream_t stream);
size_t xggm_colsum_workspace_bytes(int M, int N);

/* ---- attention core: src/lxrt/modeling.py:355-373 (BertAttention.forward after the
 * projections).  q/k/v/out rows of sample b start at row b*S of a matrix with the given row
 * stride (so fused-QKV buffers are addressed in place); head h occupies columns
 * [64h, 64h+64).  mask: additive [B,Sk] fp32 or NULL.  Sq, Sk <= 64, head_dim == 64.
 * backward: dbq/dbk/dbv (NULL ok; dbk and dbv go together): fp32 [B][heads*64] PARTIAL rows with batch stride db_bs
 * elements -- row b receives the column sums of dq/dk/dv over sample b's rows, each element written once by the
 * (sample, head) workgroup that owns it, in a fixed order.  Summed over b into the query/key/value bias gradients by
 * xggm_partial_reduce_batch (ws = dbq laid out [B][3][H]: K = 3, nblk = B) -- no floating-point atomics. */
int xggm_attn_fwd_f32(const void* q, const void* k, const void* v, const float* mask, void* out, int B, int heads, int Sq,
                      int Sk, int head_dim, int64_t q_rs, int64_t k_rs, int64_t v_rs, int64_t o_rs, float scale, float p,
                      const uint64_t* rng, uint32_t sid, xggm_stream_t stream);
int xggm_attn_fwd_bf16(const void* q, const void* k, const void* v, const float* mask, void* out, int B, int heads, int Sq,
                       int Sk, int head_dim, int64_t q_rs, int64_t k_rs, int64_t v_rs, int64_t o_rs, float scale, float p,
                       const uint64_t* rng, uint32_t sid, xggm_stream_t stream);
int xggm_attn_bwd_f32(const void* q, const void* k, const void* v, const float* mask, const void* d_out, void* dq, void* dk,
                      void* dv, int B, int heads, int Sq, int Sk, int head_dim, int64_t q_rs, int64_t k_rs, int64_t v_rs,
                      int64_t o_rs, int64_t dq_rs, int64_t dk_rs, int64_t dv_rs, float scale, float p, const uint64_t* rng,
                      uint32_t sid, float* dbq, float* dbk, float* dbv, int64_t db_bs, xggm_stream_t stream);
int xggm_attn_bwd_bf16(const void* q, const void* k, const void* v, const float* mask, const void* d_out, void* dq,
                       void* dk, void* dv, int B, int heads, int Sq, int Sk, int head_dim, int64_t q_rs, int64_t k_rs,
                       int64_t v_rs, int64_t o_rs, int64_t dq_rs, int64_t dk_rs, int64_t dv_rs, float scale, float p,
                       const uint64_t* rng, uint32_t sid, float* dbq, float* dbk, float* dbv, int64_t db_bs,
                       xggm_stream_t stream);

/* Grouped form: independent attention problems (the language and the vision stream of a layer, or
 * the two directions of a cross-attention layer, src/lxrt/modeling.py:485-516) in one launch; two
 * problems share a grid, more are launched in consecutive pairs.  Fields as the arguments above;
 * the backward fields are ignored by the forward entry points. */
typedef struct xggm_attn_problem {
    const void* q;
    const void* k;
    const void* v;
    const float* mask;
    void* out;
    int B, heads, Sq, Sk;
    int64_t q_rs, k_rs, v_rs, o_rs;
    float scale, p;
    uint32_t sid;
    const void* d_out;
    void* dq;
    void* dk;
    void* dv;
    int64_t dq_rs, dk_rs, dv_rs;
    float* dbq;
    float* dbk;
    float* dbv;
    int64_t db_bs; /* batch stride (floats) of the dbq / dbk / dbv partial rows, see above */
    /* forward, bf16 storage only: out8 (or NULL) = e4m3 copy of `out` scaled by *qscale (NULL = 1), rows of o_rs
     * bytes -- the A operand of the fp8 output projection; *amax (or NULL) raised to max |out| */
    void* out8;
    const float* qscale;
    float* amax;
    int amax_slots; /* see xggm_gemm_problem.amax_slots */
} xggm_attn_problem;
int xggm_attn_fwd_grouped_f32(const xggm_attn_problem* probs, int n, int head_dim, const uint64_t* rng, xggm_stream_t stream);
int xggm_attn_fwd_grouped_bf16(const xggm_attn_problem* probs, int n, int head_dim, const uint64_t* rng, xggm_stream_t stream);
int xggm_attn_bwd_grouped_f32(const xggm_attn_problem* probs, int n, int head_dim, const uint64_t* rng, xggm_stream_t stream);
int xggm_attn_bwd_grouped_bf16(const xggm_attn_problem* probs, int n, int head_dim, const uint64_t* rng, xggm_stream_t stream);

/* HOST: 1 = run bf16 attention on the scalar kernels (A/B tests), 0 = matrix-core kernels */
int xggm_attn_set_scalar(int on);

/* ---- row kernels -----------------------------------------------------------------------
 * out = [out +] out_scale * drop_post( LN( drop_pre(in + bias) + residual ; gamma, beta, eps) )
 * in/residual/out/z_out: T [M,H]; z_out (may alias `in`) receives the LN input, stats [M,2]
 * = {mean, rstd}.  BertAttOutput/BertOutput: src/lxrt/modeling.py:384-388, 441-445; GCNConv
 * LN: src/module/gcn.py:29; GNN read-outs with dropout(.5) and sum: src/module/gcn.py:70-77;
 * head tails: src/vqa/vqacpv2_model.py:63-105.  H % 4 == 0, H <= 2048. */
int xggm_ln_fwd_f32(const void* in, const float* bias, const void* residual, const float* gamma, const float* beta,
                    void* out, void* z_out, float* stats, int M, int H, float eps, float p_pre, float p_post,
                    const uint64_t* rng, uint32_t sid_pre, uint32_t sid_post, int accumulate, float out_scale,
                    xggm_stream_t stream);
int xggm_ln_fwd_bf16(const void* in, const float* bias, const void* residual, const float* gamma, const float* beta,
                     void* out, void* z_out, float* stats, int M, int H, float eps, float p_pre, float p_post,
                     const uint64_t* rng, uint32_t sid_pre, uint32_t sid_post, int accumulate, float out_scale,
                     xggm_stream_t stream);
/* dy = grad of `out`.  d_in: grad of `in` (NULL ok); d_res: grad of `residual` (NULL ok,
 * accumulate_dres adds to it); dgamma/dbeta/dbias (NULL ok) are ACCUMULATED (+=).
 * gelu_aux (T [M,H], NULL ok): `in` was gelu(u) of a Linear -> d_in (and dbias) are further
 * multiplied by gelu'(u), i.e. they become the gradients of u and of that Linear's bias. */
int xggm_ln_bwd_f32(const void* dy, const void* z, const float* stats, const float* gamma, void* d_in, void* d_res,
                    float* dgamma, float* dbeta, float* dbias, int M, int H, float p_pre, float p_post, const uint64_t* rng,
                    uint32_t sid_pre, uint32_t sid_post, float out_scale, int accumulate_dres, const void* gelu_aux,
                    float* ws, size_t ws_bytes, xggm_stream_t stream);
int xggm_ln_bwd_bf16(const void* dy, const void* z, const float* stats, const float* gamma, void* d_in, void* d_res,
                     float* dgamma, float* dbeta, float* dbias, int M, int H, float p_pre, float p_post,
                     const uint64_t* rng, uint32_t sid_pre, uint32_t sid_post, float out_scale, int accumulate_dres,
                     const void* gelu_aux, float* ws, size_t ws_bytes, xggm_stream_t stream);
/* Grouped forms: up to 4 independent row sets with the same H, eps, dropout rates and scaling (the
 * language and the vision stream of one LXMERT layer, src/lxrt/modeling.py:494-516) in ONE launch;
 * more than 4 problems are launched in consecutive groups.  Fields as the arguments above. */
typedef struct xggm_ln_fwd_problem {
    const void* in;
    const float* bias;
    const void* residual;
    const float* gamma;
    const float* beta;
    void* out;
    void* z_out;
    float* stats;
    int M;
    uint32_t sid_pre, sid_post;
    /* in_slabs > 0: `in` is fp32 and holds in_slabs partial sums [in_slabs][M][H] (split-K products of a long-K
     * GEMM, xggm_gemm_* with batch = in_slabs, c_f32 = 1): the row kernel adds them in order on the way in, so
     * the GEMM gets in_slabs times the workgroups and no reduction pass exists.  0: `in` is T [M][H]. */
    int in_slabs;
    /* out8 (or NULL; bf16 storage only): e4m3 copy of `out` scaled by *qscale (NULL = 1), [M][H] bytes -- the A operand
     * of the following fp8 product (QKV / FFN input); *amax (or NULL) is raised to max |out| (one atomic per workgroup). */
    void* out8;
    const float* qscale;
    float* amax;
    int amax_slots; /* see xggm_gemm_problem.amax_slots */
} xggm_ln_fwd_problem;
typedef struct xggm_ln_bwd_problem {
    const void* dy;
    const void* z;
    const float* stats;
    const float* gamma;
    void* d_in;
    void* d_res;
    float* dgamma; /* all three NULL: sums stay in ws for xggm_partial_reduce_batch */
    float* dbeta;
    float* dbias;
    const void* gelu_aux;
    float* ws;
    size_t ws_bytes;
    int M;
    uint32_t sid_pre, sid_post;
    int accumulate_dres;
} xggm_ln_bwd_problem;
int xggm_ln_fwd_grouped_f32(const xggm_ln_fwd_problem* probs, int n, int H, float eps, float p_pre, float p_post,
                            const uint64_t* rng, int accumulate, float out_scale, xggm_stream_t stream);
int xggm_ln_fwd_grouped_bf16(const xggm_ln_fwd_problem* probs, int n, int H, float eps, float p_pre, float p_post,
                             const uint64_t* rng, int accumulate, float out_scale, xggm_stream_t stream);
int xggm_ln_bwd_grouped_f32(const xggm_ln_bwd_problem* probs, int n, int H, float p_pre, float p_post,
                            const uint64_t* rng, float out_scale, xggm_stream_t stream);
int xggm_ln_bwd_grouped_bf16(const xggm_ln_bwd_problem* probs, int n, int H, float p_pre, float p_post,
                             const uint64_t* rng, float out_scale, xggm_stream_t stream);
/* out = sum over n <= 4 terms of dropout_p_post(LayerNorm(in[k]; gamma[k], beta[k], eps)) in ONE launch: the
 * jump-knowledge read-out of the graph blocks (src/module/gcn.py:70-77, src/module/gin.py:80-87).  The sum is held in
 * fp32 and rounded once.  stats[k] (or NULL): [M][2] (mean, rstd) of term k for xggm_ln_bwd_* (whose `z` is in[k] itself:
 * the terms have no bias, residual or input dropout).  `out` may not alias a term. */
typedef struct xggm_ln_sum_args {
    const void* in[4];
    const float* gamma[4];
    const float* beta[4];
    float* stats[4];
    uint32_t sid_post[4];
    void* out;
    int n, M, H;
    float eps, p_post;
    const uint64_t* rng;
} xggm_ln_sum_args;
int xggm_ln_sum_fwd_f32(const xggm_ln_sum_args* args, xggm_stream_t stream);
int xggm_ln_sum_fwd_bf16(const xggm_ln_sum_args* args, xggm_stream_t stream);
/* workspaces (bytes) of the backward row kernels: they hold one partial row per workgroup and
 * reduced vector, summed by a second kernel instead of contended atomics */
size_t xggm_ln_bwd_workspace_bytes(int M, int H);
size_t xggm_visn_embed_bwd_workspace_bytes(int M, int H);
/* Deferred second stage.  xggm_ln_bwd called with dgamma = dbeta = dbias = NULL only fills its
 * workspace ws[nblk][3][H] (nblk = workspace_bytes / (12 H)); the sums of many such workspaces
 * are then added to their gradients by ONE launch at the end of the backward pass instead of one
 * tiny launch per LayerNorm (58 per pass in LXMERT 9/5/5).  target[k] += sum_b ws[b][k][:]. */
typedef struct xggm_reduce_job {
    const float* ws;
    int nblk, K, H; /* K <= 3 partial vectors of H floats per workgroup */
    float* target[3]; /* NULL = vector not wanted */
} xggm_reduce_job;
int xggm_partial_reduce_batch(const xggm_reduce_job* jobs, int n, xggm_stream_t stream);
/* BertEmbeddings: src/lxrt/modeling.py:298-313.  ids/seg: int64 [M] (M = B*Tlen), tables T.
 * forward: out8 (or NULL; bf16 storage only) = e4m3 copy of `out` scaled by *qscale, *amax raised to max |out|
 * (amax_slots as in xggm_gemm_problem): the operand of the first fp8 product, written by its producer.
 * backward: the table gradients are GATHERED by an owner per table row (the first batch row that looked it up adds
 * the dz rows of every look-up in row order): no atomics, a fixed summation order; row 0 (padding_idx) gets none. */
int xggm_embed_fwd_f32(const int64_t* ids, const int64_t* seg, const void* word, const void* pos, const void* type,
                       const float* gamma, const float* beta, void* out, void* z_out, float* stats, int M, int Tlen, int H,
                       float eps, float p, const uint64_t* rng, uint32_t sid, void* out8, const float* qscale, float* amax,
                       int amax_slots, xggm_stream_t stream);
int xggm_embed_fwd_bf16(const int64_t* ids, const int64_t* seg, const void* word, const void* pos, const void* type,
                        const float* gamma, const float* beta, void* out, void* z_out, float* stats, int M, int Tlen, int H,
                        float eps, float p, const uint64_t* rng, uint32_t sid, void* out8, const float* qscale, float* amax,
                        int amax_slots, xggm_stream_t stream);
/* xggm_embed_fwd_* with up to XGGM_SIDE_MAX pieces of a pass's input glue done by workgroups appended to its grid instead
 * of launches of their own: XGGM_SIDE_ADDITIVE_MASK dst[i] (fp32) = (1 - src[i] (int64)) * -10000 (xggm_additive_mask,
 * src/lxrt/modeling.py:919-928); XGGM_SIDE_CAST_BF16 dst[i] (bf16) = src[i] (fp32) (xggm_cast_f32_to_bf16: the visual
 * features / boxes, src/lxrt/modeling.py:546-550).  The embedding kernel itself reads none of the outputs. */
#define XGGM_SIDE_MAX 3
#define XGGM_SIDE_ADDITIVE_MASK 1
#define XGGM_SIDE_CAST_BF16 2
typedef struct xggm_side_jobs {
    int n;
    struct {
        int kind;
        const void* src;
        void* dst;
        int64_t count;
    } job[XGGM_SIDE_MAX];
} xggm_side_jobs;
int xggm_embed_fwd_side_f32(const int64_t* ids, const int64_t* seg, const void* word, const void* pos, const void* type,
                            const float* gamma, const float* beta, void* out, void* z_out, float* stats, int M, int Tlen, int H,
                            float eps, float p, const uint64_t* rng, uint32_t sid, void* out8, const float* qscale, float* amax,
                            int amax_slots, const xggm_side_jobs* side, xggm_stream_t stream);
int xggm_embed_fwd_side_bf16(const int64_t* ids, const int64_t* seg, const void* word, const void* pos, const void* type,
                             const float* gamma, const float* beta, void* out, void* z_out, float* stats, int M, int Tlen, int H,
                             float eps, float p, const uint64_t* rng, uint32_t sid, void* out8, const float* qscale, float* amax,
                             int amax_slots, const xggm_side_jobs* side, xggm_stream_t stream);
int xggm_embed_bwd_f32(const int64_t* ids, const int64_t* seg, const void* dy, const void* z, const float* stats,
                       const float* gamma, void* dz_ws, float* dword, float* dpos, float* dtype, float* dgamma,
                       float* dbeta, int M, int Tlen, int H, float p, const uint64_t* rng, uint32_t sid, float* ws,
                       size_t ws_bytes, xggm_stream_t stream);
int xggm_embed_bwd_bf16(const int64_t* ids, const int64_t* seg, const void* dy, const void* z, const float* stats,
                        const float* gamma, void* dz_ws, float* dword, float* dpos, float* dtype, float* dgamma,
                        float* dbeta, int M, int Tlen, int H, float p, const uint64_t* rng, uint32_t sid, float* ws,
                        size_t ws_bytes, xggm_stream_t stream);
/* xggm_embed_bwd_* that also LISTS what it did to the word table's gradient (M <= the capacity of the three buffers):
 * row_ids[r] = ids[r]; row_sq[r] = |dword[ids[r]]|^2 after the add where batch row r is the owner of that table row, 0
 * elsewhere (so sum(row_sq[0 .. M)) is the table's contribution to clip_grad_norm_ when the table was zero before);
 * *row_n = M.  A caller that keeps the invariant "every non-zero row of dword is listed" clears the table with
 * xggm_zero_ranges_rows_f32 and never reads its 94 MB for the norm. */
int xggm_embed_bwd_listed_f32(const int64_t* ids, const int64_t* seg, const void* dy, const void* z, const float* stats,
                              const float* gamma, void* dz_ws, float* dword, float* dpos, float* dtype, float* dgamma,
                              float* dbeta, int M, int Tlen, int H, float p, const uint64_t* rng, uint32_t sid, float* ws,
                              size_t ws_bytes, int64_t* row_ids, float* row_sq, int* row_n, xggm_stream_t stream);
int xggm_embed_bwd_listed_bf16(const int64_t* ids, const int64_t* seg, const void* dy, const void* z, const float* stats,
                               const float* gamma, void* dz_ws, float* dword, float* dpos, float* dtype, float* dgamma,
                               float* dbeta, int M, int Tlen, int H, float p, const uint64_t* rng, uint32_t sid, float* ws,
                               size_t ws_bytes, int64_t* row_ids, float* row_sq, int* row_n, xggm_stream_t stream);
/* VisualFeatEncoder tail: src/lxrt/modeling.py:546-556.  u = feat @ W_f^T (T, from
 * xggm_gemm); boxes T [M,4]; W_b fp32 [H,4].  z1 may alias u.  stats [M,4]. */
int xggm_visn_embed_fwd_f32(const void* u, const float* bf, const void* boxes, const float* Wb, const float* bb,
                            const float* g1, const float* b1, const float* g2, const float* b2, void* out, void* z1,
                            void* z2, float* stats, int M, int H, float eps, float p, const uint64_t* rng, uint32_t sid,
                            void* out8, const float* qscale, float* amax, int amax_slots, xggm_stream_t stream);
int xggm_visn_embed_fwd_bf16(const void* u, const float* bf, const void* boxes, const float* Wb, const float* bb,
                             const float* g1, const float* b1, const float* g2, const float* b2, void* out, void* z1,
                             void* z2, float* stats, int M, int H, float eps, float p, const uint64_t* rng, uint32_t sid,
                             void* out8, const float* qscale, float* amax, int amax_slots, xggm_stream_t stream);
int xggm_visn_embed_bwd_f32(const void* dy, const void* z1, const void* z2, const float* stats, const void* boxes,
                            const float* g1, const float* g2, void* du, float* dbf, float* dg1, float* db1, float* dWb,
                            float* dbb, float* dg2, float* db2, int M, int H, float p, const uint64_t* rng, uint32_t sid,
                            float* ws, size_t ws_bytes, xggm_stream_t stream);
int xggm_visn_embed_bwd_bf16(const void* dy, const void* z1, const void* z2, const float* stats, const void* boxes,
                             const float* g1, const float* g2, void* du, float* dbf, float* dg1, float* db1, float* dWb,
                             float* dbb, float* dg2, float* db2, int M, int H, float p, const uint64_t* rng, uint32_t sid,
                             float* ws, size_t ws_bytes, xggm_stream_t stream);

/* ---- graph-generative kernels (adjacency tensors fp32 [B,N,N], N <= 64) -----------------
 * out = [out +] self_w * x + scale * (1 + *scale_ptr) * M' @ x; x/out T [B,N,H].
 * GCNConv aggregate src/module/gcn.py:28; GIN src/module/gin.py:32. */
int xggm_aggregate_f32(const float* M, const void* x, void* out, int B, int N, int H, int mode, float scale,
                       const float* scale_ptr, float self_w, int accumulate, xggm_stream_t stream);
int xggm_aggregate_bf16(const float* M, const void* x, void* out, int B, int N, int H, int mode, float scale,
                        const float* scale_ptr, float self_w, int accumulate, xggm_stream_t stream);
/* GCNConv's tail in one launch (bf16 storage): out = LayerNorm(res + M @ y; gamma, beta, eps) per sample, y = x W^T
 * taken first by a plain product -- LN(x + W (A x)) = LN(x + A (x W^T)), src/module/gcn.py:22-29.  y, res, out, z_out:
 * bf16 [B, N, H], H in {64, 128, 256, 768}; z_out (or NULL) = the rounded pre-normalisation rows and stats (or NULL) =
 * [B*N][2] (mean, rstd), both as xggm_ln_fwd_* leaves them for xggm_ln_bwd_*.  res may be y's own input x; out / z_out
 * may not alias y. */
int xggm_agg_residual_ln_bf16(const float* M, const void* y, const void* res, const float* gamma, const float* beta, void* out,
                              void* z_out, float* stats, int B, int N, int H, float eps, xggm_stream_t stream);
/* *out += sum_{b,i,c} dh[b,i,c] * (M @ x)[b,i,c]   (gradient of GIN's eps); ws: see XGGM_SUM_WS_FLOATS */
int xggm_agg_dot_f32(const float* M, const void* x, const void* dh, float* out, int B, int N, int H, float* ws,
                     xggm_stream_t stream);
int xggm_agg_dot_bf16(const float* M, const void* x, const void* dh, float* out, int B, int N, int H, float* ws,
                      xggm_stream_t stream);
/* adjacency regeneration from S = x x^T: src/module/graph_generative_modeling.py:225-228.
 * adj[i][j] = sigmoid(S[i][j] / max_r S[r][i]), zero diagonal; colmax/argmax [B,N] saved
 * (argmax = first index of the column maximum, as torch.max). */
int xggm_adj_regen_fwd(const float* S, float* adj, float* colmax, int32_t* argmax, int B, int N, xggm_stream_t stream);
int xggm_adj_regen_bwd(const float* d_adj, const float* S, const float* adj, const float* colmax, const int32_t* argmax,
                       float* dS, int B, int N, xggm_stream_t stream);
/* adjacency initialisation: src/vqa/vqacpv2.py:195-202 + src/module/graph_utils.py:162-168.
 * e fp32 [B, N(N-1)/2] (NULL = zeros); entry k goes to (i,j), the k-th element of the strict
 * upper triangle in row-major order, and to (j,i).  Noise: sigma * randn[b,min,max] mirrored
 * (randn fp32 [B,N,N] or NULL = Philox draw); gradlog = -noise / sigma^2 (NULL ok). */
int xggm_adj_init_fwd(const float* e, const float* randn, float* adj, float* gradlog, int B, int N, float sigma,
                      const uint64_t* rng, uint32_t sid, xggm_stream_t stream);
int xggm_adj_init_bwd(const float* d_adj, float* d_e, int B, int N, xggm_stream_t stream);
/* HOST helper: the (i,j) of entry k (bit-exactness tests of the index map) */
int xggm_triu_index(int k, int N, int* i_out, int* j_out);
/* add_feature_noise_v2: src/module/graph_utils.py:144-149; gradlog fp32 */
int xggm_feature_noise_f32(const void* x, const float* randn, void* out, float* gradlog, int64_t n, float sigma,
                           const uint64_t* rng, uint32_t sid, xggm_stream_t stream);
int xggm_feature_noise_bf16(const void* x, const float* randn, void* out, float* gradlog, int64_t n, float sigma,
                            const uint64_t* rng, uint32_t sid, xggm_stream_t stream);
/* out[B,2H] = [x, tanh(mean_n nodes)]: src/vqa/vqacpv2.py:216-218 */
int xggm_pool_concat_fwd_f32(const void* x, const void* nodes, void* out, int B, int N, int H, xggm_stream_t stream);
int xggm_pool_concat_fwd_bf16(const void* x, const void* nodes, void* out, int B, int N, int H, xggm_stream_t stream);
int xggm_pool_concat_bwd_f32(const void* d_out, const void* out, void* dx, void* dnodes, int B, int N, int H,
                             int accumulate_dx, xggm_stream_t stream);
int xggm_pool_concat_bwd_bf16(const void* d_out, const void* out, void* dx, void* dnodes, int B, int N, int H,
                              int accumulate_dx, xggm_stream_t stream);
/* x.unsqueeze(1).repeat(1,N,1) and its backward: src/vqa/vqacpv2.py:228 */
int xggm_bcast_rows_f32(const void* x, void* out, int B, int N, int H, xggm_stream_t stream);
int xggm_bcast_rows_bf16(const void* x, void* out, int B, int N, int H, xggm_stream_t stream);
int xggm_sum_rows_f32(const void* g, void* out, int B, int N, int H, xggm_stream_t stream);
int xggm_sum_rows_bf16(const void* g, void* out, int B, int N, int H, xggm_stream_t stream);

/* ---- graph attention (GATConv, src/module/gat.py:25-49) ---------------------------------------
 * s fp32 [B*N,2] = h [a1 a2] (from xggm_gemm, c_f32); att[i][j] = softmax_j(adj_ij == 0 ? -9e15 :
 * LeakyReLU_alpha(s1_i + s2_j)).  backward: ds (T [B*N,2]) from d_att. */
int xggm_gat_att_fwd(const float* s, const float* adj, float* att, int B, int N, float alpha, xggm_stream_t stream);
int xggm_gat_att_bwd_f32(const float* d_att, const float* att, const float* s, const float* adj, void* ds, int B, int N,
                         float alpha, xggm_stream_t stream);
int xggm_gat_att_bwd_bf16(const float* d_att, const float* att, const float* s, const float* adj, void* ds, int B, int N,
                          float alpha, xggm_stream_t stream);
/* out[m*ld_out + c] = elu(x[m*D + c]) (head concat = column slice), and its backward from y */
int xggm_elu_fwd_f32(const void* x, void* out, int M, int D, int64_t ld_out, xggm_stream_t stream);
int xggm_elu_fwd_bf16(const void* x, void* out, int M, int D, int64_t ld_out, xggm_stream_t stream);
int xggm_elu_bwd_f32(const void* dy, const void* y, void* dx, int M, int D, int64_t ld, xggm_stream_t stream);
int xggm_elu_bwd_bf16(const void* dy, const void* y, void* dx, int M, int D, int64_t ld, xggm_stream_t stream);
/* F.dropout(x, p) with the Philox mask of (rng, sid); applying it to a gradient is its backward */
int xggm_dropout_f32(const void* x, void* out, int64_t n, float p, const uint64_t* rng, uint32_t sid,
                     xggm_stream_t stream);
int xggm_dropout_bf16(const void* x, void* out, int64_t n, float p, const uint64_t* rng, uint32_t sid,
                      xggm_stream_t stream);

/* ---- losses (scalars are device fp32; *loss must hold the running value, usually 0) ------
 * Sums over the whole grid are taken WITHOUT floating-point atomics: every workgroup leaves its partial in `ws`, the
 * workgroup that finishes last adds the partials in index order and updates *loss (same bits whatever the
 * scheduling).  `ws`: XGGM_SUM_WS_FLOATS floats of device memory owned by the caller, ws[0] == 0 at launch (the
 * kernels leave it 0, so one zeroed buffer serves every launch of a stream).
 * loss_func: src/vqa/vqacpv2.py:48-51.  *loss += coef * sum (s-g)^2; ds = gout*2*coef*(s-g) */
#define XGGM_SUM_WS_FLOATS 4104
int xggm_dsm_loss_fwd_f32(const void* s, const float* g, float* loss, int64_t n, float coef, float* ws,
                          xggm_stream_t stream);
int xggm_dsm_loss_fwd_bf16(const void* s, const float* g, float* loss, int64_t n, float coef, float* ws,
                           xggm_stream_t stream);
int xggm_dsm_loss_bwd_f32(const void* s, const float* g, const float* gout, void* ds, int64_t n, float coef,
                          xggm_stream_t stream);
int xggm_dsm_loss_bwd_bf16(const void* s, const float* g, const float* gout, void* ds, int64_t n, float coef,
                           xggm_stream_t stream);
/* compute_kl_loss: src/vqa/vqacpv2.py:54-61.  rows x W; *loss += coef * sum_rows f (NULL
 * skips); dx/dy (NULL ok) = [+] *gout * coef * df. */
int xggm_symkl_f32(const void* x, const void* y, float* loss, const float* gout, void* dx, void* dy, int rows, int W,
                   float coef, int accumulate, float* ws, xggm_stream_t stream);
int xggm_symkl_bf16(const void* x, const void* y, float* loss, const float* gout, void* dx, void* dy, int rows, int W,
                    float coef, int accumulate, float* ws, xggm_stream_t stream);
/* nn.BCEWithLogitsLoss()(logit, target) * A: src/vqa/vqacpv2.py:131,173; logits fp32 */
int xggm_bce_fwd(const float* logit, const float* target, float* loss, int64_t n, float coef, float* ws,
                 xggm_stream_t stream);
int xggm_bce_bwd_f32(const float* logit, const float* target, const float* gout, void* dlogit, int64_t n, float coef,
                     xggm_stream_t stream);
int xggm_bce_bwd_bf16(const float* logit, const float* target, const float* gout, void* dlogit, int64_t n, float coef,
                      xggm_stream_t stream);

/* ---- preprocessing that feeds the path ----------------------------------------------------
 * adj_true of every sample: data/preprocess/vqa/compute_adjacency.py:38-45 (compute_cosin_sim_v2) + :90.
 *   c[i][j] = cos(cls[i], attr[j]) for j >= i, 0 below the diagonal  (torch.cosine_similarity, eps 1e-6: each
 *   norm clamped from below by eps);  a = c + c^T (the diagonal counts twice);  adj = a / max(a).
 * cls, attr: fp32 [n_img][N][D] (BERT pooled embeddings of the objects' class and attribute names), adj: fp32
 * [n_img][N][N]; N <= 64, D % 4 == 0. */
int xggm_cosine_adjacency_f32(const float* cls, const float* attr, float* adj, int n_img, int N, int D, float eps,
                              xggm_stream_t stream);

/* ---- optimiser ---------------------------------------------------------------------------
 * *out += sum g^2 over a flat fp32 range (clip_grad_norm_, src/vqa/vqacpv2.py:175).  The sum is taken in a
 * fixed order (per-workgroup partials in `ws`, added up by a second one-workgroup launch), so data-parallel
 * replicas that hold identical gradients compute bit-identical norms and stay bit-identical.  `ws`: caller-owned
 * scratch of XGGM_SQNORM_WS_FLOATS floats (contents irrelevant before, undefined after). */
#define XGGM_SQNORM_WS_FLOATS 4100
int xggm_sqnorm_f32(const float* g, int64_t n, float* out, float* ws, xggm_stream_t stream);
/* The same sum over up to 16 ranges [offsets[i], offsets[i] + lengths[i]) of ONE fp32 buffer (HOST arrays; offsets
 * multiples of 4) in two launches, partials added in (range, slice) order: *out = (overwrite ? 0 : *out) + sum;
 * *norm (or NULL) = sqrt(*out) -- the total norm nn.utils.clip_grad_norm_ returns.  n == 0 only seeds / finishes.
 * square == 0 sums the VALUES instead of their squares: the gradient-norm slots of the weight-gradient GEMMs
 * (xggm_gemm_problem.sqsum) already hold sums of squares and are added to the running sum by a second call.
 * mul: the new *out is multiplied by it before the root (1; 1 / world^2 turns the norm of SUMMED data-parallel
 * gradients into the norm of their average). */
int xggm_sqnorm_multi_f32(const float* base, const int64_t* offsets, const int64_t* lengths, int n, float* out, float* norm,
                          float* ws, int overwrite, int square, float mul, xggm_stream_t stream);
/* BertAdam.step (src/lxrt/optimization.py:159-193) fused with the clip scale
 * min(1, max_norm/(sqrt(*sqnorm)+1e-6)) and the bf16 shadow-weight write. */
int xggm_bertadam_f32(float* p, const float* g, float* m, float* v, void* shadow_bf16, int64_t n, const float* sqnorm,
                      float max_norm, float lr, const float* lr_scale, float b1, float b2, float eps, float weight_decay,
                      xggm_stream_t stream);
/* The same update with everything a deployment variant needs, as one argument block (HOST struct, copied):
 *  - g_bf16 != 0: `g` holds bf16 gradients (the data-parallel wire arena the weight-gradient GEMMs write and
 *    RCCL reduces in place: no fp32 copy of the matrix gradients exists);
 *  - lr_dev (or NULL): device scalar that replaces `lr` (edits of param_groups[i]['lr'] reach replayed graphs);
 *  - shadow8 (or NULL): e4m3 copy of the updated weights for the fp8 forward products, written in the same pass.
 *    Per-tensor scales: w8_id[(elem0 + i) >> 8] (uint16 per 256-element chunk of the arena, 0 = this chunk has no
 *    e4m3 copy) selects the entry of w8_qscale the value is multiplied by, and w8_amax[id] is raised to max |p_new|
 *    for the next scale update (xggm_fp8_scale_update).  `elem0`: arena offset of p[0], a multiple of 256 when
 *    shadow8 is given. */
typedef struct xggm_adam_args {
    float* p;
    const void* g;
    float* m;
    float* v;
    void* shadow_bf16;
    int64_t n;
    const float* sqnorm;
    float max_norm;
    float lr;
    const float* lr_dev;
    const float* lr_scale;
    float b1, b2, eps, weight_decay;
    int g_bf16;
    void* shadow8;
    const uint16_t* w8_id;
    const float* w8_qscale;
    float* w8_amax;
    int64_t elem0;
    float g_scale; /* > 0: every gradient is multiplied by it (1 / world: the exchange SUMS over the data-parallel ranks
                      and the average is taken here); <= 0 reads as 1 */
    int w8_amax_slots; /* entry id of w8_amax starts at w8_amax[id * max(1, w8_amax_slots)]; see xggm_gemm_problem.amax_slots */
} xggm_adam_args;
int xggm_bertadam_ex(const xggm_adam_args* args, xggm_stream_t stream);
/* the same update for n spans (HOST array; every arena group a pass updates, or a rank's slices of them under the sharded
 * update) in ONE launch: src/lxrt/optimization.py:159-193 loops over param_groups, each span carries its own group's
 * lr / schedule value / weight decay.  All spans of a call share the gradient type (g_bf16). */
int xggm_bertadam_multi(const xggm_adam_args* args, int n, xggm_stream_t stream);
/* *out += sum g^2 of a flat bf16 range (the wire arena), same fixed summation order as xggm_sqnorm_f32 */
int xggm_sqnorm_bf16(const void* g, int64_t n, float* out, float* ws, xggm_stream_t stream);
/* *lr_scale = warmup_linear(*step / t_total, warmup); *step += 1 (optimization.py:42-48) */
int xggm_sched_step(int64_t* step, float* lr_scale, int64_t t_total, float warmup, xggm_stream_t stream);
/* the same for n <= 16 distinct counters steps[index[i]] / lr_scale[index[i]] of one table in one launch (all
 * parameter groups of one optimiser step); index, t_total, warmup: HOST arrays of n entries */
int xggm_sched_step_multi(int64_t* steps, float* lr_scale, const int* index, const int64_t* t_total, const float* warmup,
                          int n, xggm_stream_t stream);

/* The tail of a pass in two launches instead of seven: nn.utils.clip_grad_norm_'s norm (src/vqa/vqacpv2.py:175) over up
 * to 24 ranges in all -- `n` ranges of the gradient buffer `g` (squared) and `n_slots` ranges of the slot table the
 * weight-gradient products filled (xggm_gemm_problem.sqsum: summed as they are) -- with partials added in (range, slice)
 * order (fixed: replicas stay bit-identical); *out = sum * mul, *norm (or NULL) = sqrt(*out).  With `tail` the finishing
 * workgroup also performs xggm_sched_step_multi's schedule step for tail->n counters (BertAdam.step's bookkeeping,
 * src/lxrt/optimization.py:170-180) and, with tail->rng, xggm_rng_advance(rng, rng_by).  offsets / lengths / tail's
 * arrays: HOST arrays; offsets multiples of 4; ws: XGGM_SQNORM_WS_FLOATS floats. */
typedef struct {
    int64_t* steps;
    float* lr_scale;
    const int* index;
    const int64_t* t_total;
    const float* warmup;
    int n;
    uint64_t* rng; /* or NULL */
    uint64_t rng_by;
} xggm_pass_tail;
int xggm_clip_norm_f32(const float* g, const int64_t* offsets, const int64_t* lengths, int n, const float* slots,
                       const int64_t* slot_offsets, const int64_t* slot_lengths, int n_slots, float* out, float* norm, float* ws,
                       float mul, const xggm_pass_tail* tail, xggm_stream_t stream);
/* the same pair of launches over up to 24 ranges of a BF16 buffer (the data-parallel wire arena, offsets multiples of 8):
 * *out = ((accumulate ? *out : 0) + sum g^2) * mul -- accumulate = 1 extends a sum another call (or an all-reduce over the
 * ranks, sharded update) has left in *out; mul = 1 / world^2 turns the norm of SUMMED gradients into that of their mean. */
int xggm_clip_norm_bf16(const void* g, const int64_t* offsets, const int64_t* lengths, int n, float* out, float* norm, float* ws,
                        int accumulate, float mul, const xggm_pass_tail* tail, xggm_stream_t stream);

/* out = in with the diagonal of every [N, N] matrix zeroed: adj_true.triu(1) + adj_true.tril(-1), src/vqa/vqacpv2.py:188 */
int xggm_zero_diag_f32(const float* in, float* out, int B, int N, xggm_stream_t stream);
/* out[i] = (1 - mask[i]) * -10000: additive attention mask from the int64 token mask, src/lxrt/modeling.py:919-928 */
int xggm_additive_mask(const int64_t* mask, float* out, int64_t n, xggm_stream_t stream);
/* *out = *a + *b + *c + *d (NULL terms skipped): the sum of the loss terms of a pass, src/vqa/vqacpv2.py:220-221 */
int xggm_add_scalars_f32(const float* a, const float* b, const float* c, const float* d, float* out, xggm_stream_t stream);
/* out = a + b (+ c (+ d)), n elements, summed in fp32 and rounded once; out may alias an input.  Replaces the sums
 * torch's autograd engine forms (at::add) where a tensor of the training loop feeds several consumers
 * (reference: x, feat_seq[1], node_feats, adj_noise in src/vqa/vqacpv2.py:195-251). */
int xggm_add_n_f32(const float* a, const float* b, const float* c, const float* d, float* out, int64_t n, xggm_stream_t stream);
int xggm_add_n_bf16(const void* a, const void* b, const void* c, const void* d, void* out, int64_t n, xggm_stream_t stream);
/* Rows of a [V, H] table by index (H % 4 == 0; indices outside [0, V) are skipped): dst[i, :] = bf16(src[idx[i], :]) and
 * dst[idx[i], :] = src[i, :] (duplicate indices must carry identical rows).  Data parallelism exchanges only the rows of
 * the word-embedding gradient (nn.Embedding(30522, 768), src/lxrt/modeling.py:283) that some rank touched. */
int xggm_gather_rows_bf16(const float* src, const int64_t* idx, void* dst, int n, int H, int64_t V, xggm_stream_t stream);
int xggm_scatter_rows_bf16(const void* src, const int64_t* idx, void* dst, int n, int H, int64_t V, xggm_stream_t stream);
/* dst [rows, ld] bf16 = cast(src [rows, n], fp32 if src_f32 else bf16), columns n .. ld - 1 zero: the padded row stride
 * the backward products of an odd-width output run on (answer logits of src/vqa/vqacpv2_model.py:63-70). */
int xggm_pad_rows_bf16(const void* src, int src_f32, void* dst, int rows, int n, int ld, xggm_stream_t stream);
/* zero up to 16 element ranges [offset[i], offset[i] + length[i]) of one fp32 buffer in ONE launch: the
 * atomically accumulated gradient ranges of all parameter groups at the start of a backward pass.
 * offsets / lengths are HOST arrays (copied into the kernel arguments); both multiples of 4. */
int xggm_zero_ranges_f32(float* base, const int64_t* offsets, const int64_t* lengths, int n, xggm_stream_t stream);
/* the same fill (n may be 0) and, in the same launch, rows row_ids[0 .. min(*row_n, row_cap)) of the fp32 table
 * [R, H] zeroed (row_ids / row_n: DEVICE buffers, as xggm_embed_bwd_listed_* left them; H a multiple of 4) */
int xggm_zero_ranges_rows_f32(float* base, const int64_t* offsets, const int64_t* lengths, int n, float* table,
                              const int64_t* row_ids, const int* row_n, int row_cap, int64_t R, int H, xggm_stream_t stream);

/* ---- small element-wise kernels ---------------------------------------------------------
 * out = scale * (1 + *scale_ptr) * x   (GIN's (1+eps), src/module/gin.py:32) */
int xggm_scale_f32(const void* x, void* out, int64_t n, float scale, const float* scale_ptr, xggm_stream_t stream);
int xggm_scale_bf16(const void* x, void* out, int64_t n, float scale, const float* scale_ptr, xggm_stream_t stream);
/* out(T) = dy(fp32) * y(fp32) * (1 - y): backward of encoder_adj's Sigmoid (vqacpv2_model.py:93) */
int xggm_sigmoid_bwd_f32(const float* dy, const float* y, void* out, int64_t n, xggm_stream_t stream);
int xggm_sigmoid_bwd_bf16(const float* dy, const float* y, void* out, int64_t n, xggm_stream_t stream);
/* out(T) = dy(T) * (1 - y(T)^2): backward of BertPooler's tanh (src/lxrt/modeling.py:619) */
int xggm_tanh_bwd_f32(const void* dy, const void* y, void* out, int64_t n, xggm_stream_t stream);
int xggm_tanh_bwd_bf16(const void* dy, const void* y, void* out, int64_t n, xggm_stream_t stream);
/* out(T) = x(fp32): loader tensors (feats, boxes) and fp32 logit gradients entering the T path */
int xggm_cast_from_f32_f32(const float* x, void* out, int64_t n, xggm_stream_t stream);
int xggm_cast_from_f32_bf16(const float* x, void* out, int64_t n, xggm_stream_t stream);

/* ---- utilities ---------------------------------------------------------------------------*/
int xggm_rng_advance(uint64_t* rng, uint64_t by, xggm_stream_t stream);
int xggm_cast_f32_to_bf16(const float* x, void* out, int64_t n, xggm_stream_t stream);
/* test hooks: the dropout keep-scale (0 or 1/(1-p)) and N(0,1) draw of elements 0..n-1 */
int xggm_dropout_mask(float* out, int64_t n, float p, const uint64_t* rng, uint32_t sid, xggm_stream_t stream);
int xggm_normal(float* out, int64_t n, const uint64_t* rng, uint32_t sid, xggm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* XGGM_H */
