"""Thin tensor-level wrappers over the C ABI (one Python function per kernel family).

No arithmetic happens here: each function checks shapes on the host -- a kernel that
faults can reset the whole GPU node -- and enqueues one HIP kernel on the current stream.
"""
import torch

from . import _lib
from ._lib import call, ptr, stream

ACT_NONE, ACT_GELU, ACT_SIGMOID, ACT_TANH, ACT_GELU_GRAD = 0, 1, 2, 3, 4
AGG_PLAIN, AGG_TRANSPOSE, AGG_SYMMETRIZE = 0, 1, 2
F32 = torch.float32
BF16 = torch.bfloat16


def sfx(dt):
    if dt == F32:
        return "f32"
    if dt == BF16:
        return "bf16"
    raise TypeError("xggm_amd supports float32 and bfloat16 activations, got %s" % dt)


def _chk(t, dt=None, name="tensor"):
    if not t.is_cuda:
        raise RuntimeError("xggm_amd: %s must live on the GPU (no CPU fallback)" % name)
    if dt is not None and t.dtype != dt:
        raise TypeError("xggm_amd: %s has dtype %s, expected %s" % (name, t.dtype, dt))
    return t


def _c(t, dt=None, name="tensor"):
    _chk(t, dt, name)
    if not t.is_contiguous():
        raise RuntimeError("xggm_amd: %s must be contiguous" % name)
    return t


# ----------------------------------------------------------------------------- GEMM
def gemm_raw(dt, A, B, C, M, N, K, a_rs, a_ks, b_ns, b_ks, ldc, batch=1, a_bs=0, b_bs=0, c_bs=0,
             bias=None, residual=None, preact=None, aux=None, act=ACT_NONE, c_f32=False,
             accumulate=False, alpha=1.0, colsum=None):
    """``colsum``: fp32 [batch * ceil(M/32), N] partial rows (see xggm.h), overwritten"""
    if colsum is not None:
        assert colsum.dtype == F32 and colsum.numel() >= batch * colsum_rows(M) * N
    call("xggm_gemm_" + sfx(dt), ptr(A), ptr(B), ptr(C), M, N, K, a_rs, a_ks, b_ns, b_ks, ldc, batch,
         a_bs, b_bs, c_bs, ptr(bias), ptr(residual), ptr(preact), ptr(aux), ptr(colsum), act, int(c_f32),
         int(accumulate), float(alpha), stream())


def _rows(x):
    """(M, K, row stride) of a 2-D view whose last dim is contiguous."""
    if x.dim() != 2 or x.stride(1) != 1:
        raise RuntimeError("xggm_amd: expected a 2-D operand with unit inner stride")
    return x.shape[0], x.shape[1], x.stride(0)


def linear_fwd(x, w, bias=None, act=ACT_NONE, want_preact=False, out_f32=False, residual=None):
    """y[M,N] = act(x[M,K] @ w[N,K]^T + bias) (+ residual).  Returns (y, preact|None)."""
    M, K, a_rs = _rows(_chk(x))
    N, K2 = w.shape
    if K2 != K or w.dtype != x.dtype or not w.is_contiguous():
        raise RuntimeError("linear_fwd: weight %s/%s does not match input %s/%s"
                           % (tuple(w.shape), w.dtype, tuple(x.shape), x.dtype))
    if bias is not None:
        _c(bias, F32, "bias")
        assert bias.numel() == N
    y = torch.empty((M, N), device=x.device, dtype=F32 if out_f32 else x.dtype)
    pre = torch.empty((M, N), device=x.device, dtype=x.dtype) if want_preact else None
    if residual is not None:
        assert residual.shape == y.shape and residual.dtype == x.dtype and residual.is_contiguous()
    gemm_raw(x.dtype, x, w, y, M, N, K, a_rs, 1, K, 1, N, bias=bias, preact=pre, act=act,
             c_f32=out_f32, residual=residual)
    return y, pre


# ----------------------------------------------------------------------------- fp8 forward (config C5)
FP8 = torch.float8_e4m3fn
FP8_MAX = 448.0


def quantize_fp8(x, qscale=None, amax=None, out=None):
    """x (fp32 / bf16, contiguous, numel % 8 == 0) -> e4m3fn(clamp(x * qscale, +-448)); ``qscale``: 1-element fp32
    device tensor or None (1); ``amax``: 1-element fp32 device tensor raised to max |x|, or None; ``out``: a
    contiguous 1-byte tensor of the same number of elements to write into."""
    _c(x)
    if x.dtype not in (F32, BF16):
        raise RuntimeError("quantize_fp8: fp32 or bf16 input expected, got %s" % x.dtype)
    if out is not None:
        _c(out)
        assert out.numel() == x.numel() and out.element_size() == 1
    y = out if out is not None else torch.empty(x.shape, device=x.device, dtype=FP8)
    call("xggm_quantize_fp8e4m3_" + sfx(x.dtype), ptr(x), ptr(y), x.numel(), ptr(qscale), ptr(amax), stream())
    return y


def fp8_scale_for(amax):
    """(quantisation scale, its reciprocal) of a tensor whose max |x| is ``amax`` (1-element fp32 device tensor):
    amax lands on the largest e4m3 value; an all-zero tensor gets scale 1."""
    q = torch.where(amax > 0, FP8_MAX / amax, torch.ones_like(amax))
    return q, 1.0 / q


def linear_fwd_fp8(x8, w8, sx, sw, bias=None, act=ACT_NONE, want_preact=False, out_f32=False, residual=None):
    """y[M,N] = act(sx * sw * (x8[M,K] @ w8[N,K]^T) + bias) (+ residual) with e4m3 operands; bf16 (or fp32) result.
    ``sx`` / ``sw``: 1-element fp32 device tensors, the reciprocals of the operands' quantisation scales."""
    M, K, a_rs = _rows(x8)
    N, K2 = w8.shape
    if x8.dtype != FP8 or w8.dtype != FP8 or K2 != K or w8.stride(1) != 1:
        raise RuntimeError("linear_fwd_fp8: e4m3 operands [M,K] / [N,K] expected")
    y = torch.empty((M, N), device=x8.device, dtype=F32 if out_f32 else BF16)
    pre = torch.empty((M, N), device=x8.device, dtype=BF16) if want_preact else None
    if residual is not None:
        assert residual.shape == y.shape and residual.dtype == BF16 and residual.is_contiguous()
    call("xggm_gemm_fp8e4m3", ptr(x8), ptr(w8), ptr(y), M, N, K, a_rs, w8.stride(0), N, ptr(sx), ptr(sw), ptr(bias),
         ptr(residual), ptr(pre), act, int(out_f32), stream())
    return y, pre


def linear_dgrad(dy, w, residual=None, gelu_aux=None):
    """dx[M,K] = dy[M,N] @ w[N,K] (+ residual), optionally times gelu'(aux) (aux, dx same shape)."""
    M, N, a_rs = _rows(_chk(dy))
    N2, K = w.shape
    if N2 != N or w.dtype != dy.dtype or not w.is_contiguous():
        raise RuntimeError("linear_dgrad: weight %s does not match grad %s" % (tuple(w.shape), tuple(dy.shape)))
    dx = torch.empty((M, K), device=dy.device, dtype=dy.dtype)
    if residual is not None:
        assert residual.shape == dx.shape and residual.dtype == dy.dtype and residual.is_contiguous()
    if gelu_aux is not None:
        assert gelu_aux.shape == dx.shape and gelu_aux.dtype == dy.dtype and gelu_aux.is_contiguous()
    # B(k=n', n=k') = w[n', k']: column index contiguous, reduction index strided by K
    gemm_raw(dy.dtype, dy, w, dx, M, K, N, a_rs, 1, 1, K, K, residual=residual, aux=gelu_aux,
             act=ACT_GELU_GRAD if gelu_aux is not None else ACT_NONE)
    return dx


def linear_wgrad(dy, x, gw, accumulate):
    """gw[N,K] (+)= dy[M,N]^T @ x[M,K]; gw fp32, or bf16 (the data-parallel wire arena: the gradient is born in the
    type it crosses the links in)."""
    M, N, dy_rs = _rows(_chk(dy))
    M2, K, x_rs = _rows(_chk(x))
    if M2 != M or x.dtype != dy.dtype:
        raise RuntimeError("linear_wgrad: %s vs %s" % (tuple(dy.shape), tuple(x.shape)))
    _c(gw)
    assert tuple(gw.shape) == (N, K) and (gw.dtype == F32 or gw.dtype == dy.dtype)
    # A(m=n, k=m') = dy[m', n]: row index contiguous; B(k=m', n=k) = x[m', k]
    gemm_raw(dy.dtype, dy, x, gw, N, K, M, 1, dy_rs, 1, x_rs, K, c_f32=gw.dtype == F32, accumulate=accumulate)


def _ws(nbytes, device):
    """scratch for the two-stage reductions (caller-owned, as the C ABI requires)."""
    return torch.empty((nbytes + 3) // 4, device=device, dtype=F32), nbytes


# ---- grouped launches: several independent products in one grid -------------------------------
import ctypes as _ct
import os


class GemmProblem(_ct.Structure):
    """mirror of ``xggm_gemm_problem`` (include/xggm.h)"""
    _fields_ = [("A", _ct.c_void_p), ("B", _ct.c_void_p), ("C", _ct.c_void_p),
                ("M", _ct.c_int), ("N", _ct.c_int), ("K", _ct.c_int),
                ("a_rs", _ct.c_int64), ("a_ks", _ct.c_int64), ("b_ns", _ct.c_int64), ("b_ks", _ct.c_int64),
                ("ldc", _ct.c_int64), ("batch", _ct.c_int),
                ("a_bs", _ct.c_int64), ("b_bs", _ct.c_int64), ("c_bs", _ct.c_int64),
                ("bias", _ct.c_void_p), ("residual", _ct.c_void_p), ("preact", _ct.c_void_p), ("aux", _ct.c_void_p),
                ("colsum", _ct.c_void_p), ("act", _ct.c_int), ("c_f32", _ct.c_int), ("accumulate", _ct.c_int), ("alpha", _ct.c_float),
                ("sqsum", _ct.c_void_p),
                ("scale_a", _ct.c_void_p), ("scale_b", _ct.c_void_p), ("c8", _ct.c_void_p), ("c8_qscale", _ct.c_void_p),
                ("c8_amax", _ct.c_void_p), ("amax_slots", _ct.c_int)]


def _problem(A, B, C, M, N, K, a_rs, a_ks, b_ns, b_ks, ldc, bias=None, residual=None, preact=None, aux=None,
             act=ACT_NONE, c_f32=False, accumulate=False, colsum=None):
    p = GemmProblem(ptr(A), ptr(B), ptr(C), M, N, K, a_rs, a_ks, b_ns, b_ks, ldc, 1, 0, 0, 0, ptr(bias),
                    ptr(residual), ptr(preact), ptr(aux), ptr(colsum), act, int(c_f32), int(accumulate), 1.0)
    # the descriptor holds raw pointers and may be launched later (grouped with other problems): it keeps its
    # operands alive, otherwise the caching allocator can hand their memory to the next torch.empty
    p.keep = (A, B, C, bias, residual, preact, aux, colsum)
    return p


def p_fwd(x, w, bias=None, act=ACT_NONE, want_preact=False, out_f32=False, emit8=None):
    """problem for y = act(x @ w^T + bias); returns (problem, y, preact).  ``emit8`` = (y8, qscale, amax): also
    write y as e4m3 (see p_fwd8)."""
    M, K, a_rs = _rows(_chk(x))
    N = w.shape[0]
    assert w.shape[1] == K and w.dtype == x.dtype and w.is_contiguous()
    y = torch.empty((M, N), device=x.device, dtype=F32 if out_f32 else x.dtype)
    pre = torch.empty((M, N), device=x.device, dtype=x.dtype) if want_preact else None
    p = _problem(x, w, y, M, N, K, a_rs, 1, K, 1, N, bias=bias, preact=pre, act=act, c_f32=out_f32)
    if emit8 is not None:
        y8, q, amax = emit8
        assert y8.shape == y.shape and y8.element_size() == 1 and y8.is_contiguous() and x.dtype == BF16 and not out_f32
        p.c8, p.c8_qscale, p.c8_amax = ptr(y8), ptr(q), ptr(amax)
        p.amax_slots = amax.numel() if amax is not None else 0  # a table entry spread over several floats (fp8.AMAX_SLOTS)
        p.keep = p.keep + (y8, q, amax)
    return p, y, pre


def p_fwd_splitk(x, w, S):
    """problem for the S partial products of y = x @ w^T over K / S wide slices of the reduction, fp32 [S, M, N]
    (the batch dimension of the GEMM walks the slices); summed by the consumer (``LnFwdReq``).  For long-K
    products with few tiles (FFN output: K = 3072, N = 768): S times the workgroups, 1 / S of the k-loop each."""
    M, K, a_rs = _rows(_chk(x))
    N = w.shape[0]
    assert w.shape[1] == K and w.dtype == x.dtype and w.is_contiguous()
    if K % (64 * S):
        raise RuntimeError("p_fwd_splitk: K = %d is not a multiple of 64 * %d" % (K, S))
    part = torch.empty((S, M, N), device=x.device, dtype=F32)
    p = _problem(x, w, part, M, N, K // S, a_rs, 1, K, 1, N, c_f32=True)
    p.batch, p.a_bs, p.b_bs, p.c_bs = S, K // S, K // S, M * N
    return p, part


def p_fwd8(x8, w8, sx, sw, bias=None, act=ACT_NONE, want_preact=False, emit8=None, split=0):
    """problem for y = act(sx * sw * (x8 @ w8^T) + bias) with e4m3 operands x8 [M, K], w8 [N, K] (1-byte tensors, k
    contiguous); ``sx`` / ``sw``: 1-element fp32 device tensors (reciprocal quantisation scales).  ``emit8`` =
    (y8, qscale, amax): the producer also writes y as e4m3 (operand of the next fp8 product).  ``split`` > 1: the
    partial products over K / split wide slices as fp32 [split, M, N] (see p_fwd_splitk).
    Returns (problem, y, preact)."""
    M, K, a_rs = _rows(_chk(x8))
    N = w8.shape[0]
    assert w8.shape[1] == K and x8.element_size() == 1 and w8.element_size() == 1 and w8.stride(1) == 1
    if split > 1:
        if K % (128 * split):
            raise RuntimeError("p_fwd8: K = %d is not a multiple of 128 * %d" % (K, split))
        y = torch.empty((split, M, N), device=x8.device, dtype=F32)
        p = _problem(x8, w8, y, M, N, K // split, a_rs, 1, w8.stride(0), 1, N, c_f32=True)
        p.batch, p.a_bs, p.b_bs, p.c_bs = split, K // split, K // split, M * N
        pre = None
    else:
        y = torch.empty((M, N), device=x8.device, dtype=BF16)
        pre = torch.empty((M, N), device=x8.device, dtype=BF16) if want_preact else None
        p = _problem(x8, w8, y, M, N, K, a_rs, 1, w8.stride(0), 1, N, bias=bias, preact=pre, act=act)
    p.scale_a, p.scale_b = ptr(sx), ptr(sw)
    p.keep = p.keep + (sx, sw)
    if emit8 is not None:
        y8, q, amax = emit8
        assert y8.shape == y.shape and y8.element_size() == 1 and y8.is_contiguous()
        p.c8, p.c8_qscale, p.c8_amax = ptr(y8), ptr(q), ptr(amax)
        p.amax_slots = amax.numel() if amax is not None else 0  # a table entry spread over several floats (fp8.AMAX_SLOTS)
        p.keep = p.keep + (y8, q, amax)
    return p, y, pre


GROUP_MAX = 6  # MAX_GROUP of gemm.hip: problems one grouped launch takes


def gemm_group8(problems):
    """launch up to GROUP_MAX e4m3 forward products in one grid (more: consecutive groups); the measured tile table
    applies as for the bf16 groups (signatures start with ``e4m3|``)."""
    for i in range(0, len(problems), GROUP_MAX):
        chunk = problems[i:i + GROUP_MAX]
        arr = (GemmProblem * len(chunk))(*chunk)
        pin = 0
        if TILE_TABLE or TILE_HOOK is not None:
            sig = gemm_signature("e4m3", chunk)
            pin = TILE_TABLE.get(sig, 0)
            if TILE_HOOK is not None:
                pin = TILE_HOOK("e4m3", chunk, sig, arr) or pin
        if pin:
            _lib.lib.xggm_gemm_set_group_tile(pin)
        try:
            call("xggm_gemm_grouped_fp8e4m3", _ct.cast(arr, _ct.c_void_p), len(chunk), stream())
        finally:
            if pin:
                _lib.lib.xggm_gemm_set_group_tile(0)


def p_dgrad(dy, w, residual=None, gelu_aux=None, colsum=None, into=None, defer=None):
    """problem for dx = dy @ w (+ residual) (* gelu'(aux)); ``colsum`` (fp32 [K]) += column sums of dx
    (the bias gradient of the Linear that produced the activation) -- the epilogue leaves one partial row per 32
    output rows, a reduce job adds them to ``colsum`` in a fixed order: appended to ``defer`` (the backward pass's
    list, see functional.Runtime.defer_list) or, without one, run by ``gemm_group`` right behind the launch;
    ``into``: an existing gradient of the same input, the result is ADDED to it (second consumer of one tensor);
    returns (problem, dx)."""
    M, N, a_rs = _rows(_chk(dy))
    K = w.shape[1]
    assert w.shape[0] == N and w.dtype == dy.dtype and w.is_contiguous()
    if into is not None:
        assert tuple(into.shape) == (M, K) and into.dtype == dy.dtype and into.is_contiguous()
        assert gelu_aux is None and colsum is None
        return _problem(dy, w, into, M, K, N, a_rs, 1, 1, K, K, residual=residual, accumulate=True), into
    dx = torch.empty((M, K), device=dy.device, dtype=dy.dtype)
    part = job = None
    if colsum is not None:
        _c(colsum, F32, "colsum")
        assert colsum.numel() == K
        if K % 4:
            raise RuntimeError("p_dgrad: column sums need a width that is a multiple of 4 (got %d)" % K)
        part = torch.empty((colsum_rows(M), K), device=dy.device, dtype=F32)
        job = (part, colsum_rows(M), 1, K, (colsum,))
    p = _problem(dy, w, dx, M, K, N, a_rs, 1, 1, K, K, residual=residual, aux=gelu_aux,
                 act=ACT_GELU_GRAD if gelu_aux is not None else ACT_NONE, colsum=part)
    if job is not None:
        if defer is not None:
            defer.append(job)
        else:
            p.post = job
    return p, dx


def p_wgrad(dy, x, gw, accumulate, sqsum=None):
    """problem for gw (+)= dy^T @ x (fp32).  ``sqsum``: fp32 [ceil(N/64) * ceil(K/64)] slots that receive the sums of
    squares of the stored gradient per 64 x 64 block (tuned bf16 kernels only), or None."""
    M, N, dy_rs = _rows(_chk(dy))
    M2, K, x_rs = _rows(_chk(x))
    assert M2 == M and x.dtype == dy.dtype and tuple(gw.shape) == (N, K) and gw.is_contiguous()
    assert gw.dtype == F32 or gw.dtype == dy.dtype  # bf16: the data-parallel wire arena
    p = _problem(dy, x, gw, N, K, M, 1, dy_rs, 1, x_rs, K, c_f32=gw.dtype == F32, accumulate=accumulate)
    if sqsum is not None:
        _c(sqsum, F32, "sqsum slots")
        assert sqsum.numel() == ((N + 63) // 64) * ((K + 63) // 64)
        p.sqsum = ptr(sqsum)
        p.keep = p.keep + (sqsum,)
    return p


# ---- measured tile choice -----------------------------------------------------------------------
# The library picks the tile of a grouped launch from a fitted cost model (gemm.hip: pick_group_tile).  Where a
# measurement on the hardware disagrees, the measurement wins: ``gemm_tiles_gfx950.json`` (written by tools/tune_gemm.py
# from in-step timings of every distinct launch of the training iteration under every tile) maps a launch SIGNATURE --
# shapes and operand layouts of its problems -- to the tile that was fastest; unknown signatures keep the model's choice.
# The tile changes speed, not values: every tile walks k in the same order and the column sums are taken per 32 output
# rows whatever the tile, so products, gradients and bias gradients are bit-identical under every tile (tests); only the
# 64 x 64 norm slots add their squares in a tile-dependent -- fixed -- order, i.e. the clip norm can differ in its last
# bit between two TABLES.  The table is part of the configuration: one table, one set of bits.
def _load_tile_table():
    import json
    import os
    path = os.environ.get("XGGM_TILE_TABLE") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_tiles_gfx950.json")
    if os.environ.get("XGGM_TILE_TABLE") == "0" or not os.path.exists(path):
        return {}
    try:
        return {k: int(v) for k, v in json.load(open(path)).get("tiles", {}).items()}
    except (ValueError, OSError):
        return {}


TILE_TABLE = _load_tile_table()
TILE_HOOK = None  # tools/tune_gemm.py: callable(dt, chunk, signature) -> tile pin (or None) run in front of every launch


def gemm_signature(dt, chunk):
    """what the tile choice of a grouped launch may depend on: per problem M x N x K (x batch), which operands have
    the reduction index contiguous, fp32 output, the epilogue's extras"""
    parts = []
    for p in chunk:
        ext = ("f" if p.c_f32 else "") + ("a" if p.accumulate else "") + ("r" if p.residual else "") + \
              ("g" if p.act == ACT_GELU_GRAD else ("G" if p.act == ACT_GELU else "")) + ("c" if p.colsum else "")
        parts.append("%dx%dx%d%s:%d%d%s" % (p.M, p.N, p.K, ("b%d" % p.batch) if p.batch != 1 else "", int(p.a_ks == 1),
                                            int(p.b_ks == 1), ext))
    return (dt if isinstance(dt, str) else sfx(dt)) + "|" + "+".join(parts)


def gemm_group(dt, problems):
    """launch up to GROUP_MAX independent products in one grid (more: consecutive groups)."""
    for i in range(0, len(problems), GROUP_MAX):
        chunk = problems[i:i + GROUP_MAX]
        arr = (GemmProblem * len(chunk))(*chunk)
        pin = 0
        if TILE_TABLE or TILE_HOOK is not None:
            sig = gemm_signature(dt, chunk)
            pin = TILE_TABLE.get(sig, 0)
            if TILE_HOOK is not None:
                pin = TILE_HOOK(dt, chunk, sig, arr) or pin
        if pin:
            _lib.lib.xggm_gemm_set_group_tile(pin)
        try:
            call("xggm_gemm_grouped_" + sfx(dt), _ct.cast(arr, _ct.c_void_p), len(chunk), stream())
        finally:
            if pin:
                _lib.lib.xggm_gemm_set_group_tile(0)
        reduce_batch([p.post for p in chunk if getattr(p, "post", None) is not None])  # column sums nobody deferred


def colsum(x, out):
    """out[n] += sum_m x[m, n]  (fp32 accumulator)."""
    M, N, ld = _rows(_chk(x))
    _c(out, F32, "colsum out")
    assert out.numel() == N
    ws, nb = _ws(_lib.lib.xggm_colsum_workspace_bytes(M, N), x.device)
    call("xggm_colsum_" + sfx(x.dtype), ptr(x), ptr(out), M, N, ld, ptr(ws), nb, stream())


def bmm_nt(a, b, out_f32=True, into=None):
    """out[z] = a[z] @ b[z]^T for contiguous a [Z,M,K], b [Z,N,K] -> [Z,M,N].  ``into``: an existing [Z,M,N] result the
    product is ADDED to (returned)."""
    Z, M, K = a.shape
    Z2, N, K2 = b.shape
    assert Z == Z2 and K == K2 and a.dtype == b.dtype
    _c(a), _c(b)
    if into is not None:
        _c(into, F32 if out_f32 else a.dtype)
        assert tuple(into.shape) == (Z, M, N)
        out = into
    else:
        out = torch.empty((Z, M, N), device=a.device, dtype=F32 if out_f32 else a.dtype)
    gemm_raw(a.dtype, a, b, out, M, N, K, K, 1, K, 1, N, batch=Z, a_bs=M * K, b_bs=N * K, c_bs=M * N,
             c_f32=out_f32, accumulate=into is not None)
    return out


# ----------------------------------------------------------------------------- attention
def attn_fwd(q, k, v, mask, B, heads, Sq, Sk, p, rng, sid):
    """q/k/v: 2-D row views [B*S, heads*64] with unit inner stride (may be slices of a fused
    QKV buffer).  Returns ctx [B*Sq, heads*64]."""
    d = 64
    H = heads * d
    for t, S in ((q, Sq), (k, Sk), (v, Sk)):
        _chk(t)
        if t.dim() != 2 or t.shape[0] != B * S or t.shape[1] != H or t.stride(1) != 1:
            raise RuntimeError("attn_fwd: bad operand shape %s" % (tuple(t.shape),))
    if mask is not None:
        _c(mask, F32, "mask")
        assert tuple(mask.shape) == (B, Sk)
    out = torch.empty((B * Sq, H), device=q.device, dtype=q.dtype)
    call("xggm_attn_fwd_" + sfx(q.dtype), ptr(q), ptr(k), ptr(v), ptr(mask), ptr(out), B, heads, Sq, Sk, d,
         q.stride(0), k.stride(0), v.stride(0), H, 0.125, float(p), ptr(rng), sid, stream())
    return out


def _bias_partials(B, H, dev, dbq, dbk, dbv):
    """workspace [B][3][H] the attention backward leaves the per-sample column sums of dq / dk / dv in, and the reduce
    job that adds them into the q / k / v bias gradients (fixed order: no atomics)"""
    if dbq is None and dbk is None:
        return None, None
    for t in (dbq, dbk, dbv):
        if t is not None:
            _c(t, F32, "bias gradient")
            assert t.numel() == H
    ws = torch.empty((B, 3, H), device=dev, dtype=F32)
    return ws, (ws, B, 3, H, (dbq, dbk, dbv))


def attn_bwd(q, k, v, mask, d_out, dq, dk, dv, B, heads, Sq, Sk, p, rng, sid, dbq=None, dbk=None, dbv=None):
    """dbq/dbk/dbv: fp32 [heads*64] accumulators of the query/key/value bias gradients"""
    d = 64
    H = heads * d
    _c(d_out)
    assert tuple(d_out.shape) == (B * Sq, H) and d_out.dtype == q.dtype
    for t, S in ((dq, Sq), (dk, Sk), (dv, Sk)):
        if t.dim() != 2 or t.shape[0] != B * S or t.shape[1] != H or t.stride(1) != 1 or t.dtype != q.dtype:
            raise RuntimeError("attn_bwd: bad gradient buffer %s" % (tuple(t.shape),))
    ws, job = _bias_partials(B, H, q.device, dbq, dbk, dbv)
    call("xggm_attn_bwd_" + sfx(q.dtype), ptr(q), ptr(k), ptr(v), ptr(mask), ptr(d_out), ptr(dq), ptr(dk),
         ptr(dv), B, heads, Sq, Sk, d, q.stride(0), k.stride(0), v.stride(0), H, dq.stride(0),
         dk.stride(0), dv.stride(0), 0.125, float(p), ptr(rng), sid,
         ptr(ws[0, 0]) if (ws is not None and dbq is not None) else None,
         ptr(ws[0, 1]) if (ws is not None and dbk is not None) else None,
         ptr(ws[0, 2]) if (ws is not None and dbv is not None) else None, 3 * H, stream())
    if job is not None:
        reduce_batch([job])


class AttnProblem(_ct.Structure):
    """mirror of ``xggm_attn_problem`` (include/xggm.h)"""
    _fields_ = [("q", _ct.c_void_p), ("k", _ct.c_void_p), ("v", _ct.c_void_p), ("mask", _ct.c_void_p), ("out", _ct.c_void_p),
                ("B", _ct.c_int), ("heads", _ct.c_int), ("Sq", _ct.c_int), ("Sk", _ct.c_int),
                ("q_rs", _ct.c_int64), ("k_rs", _ct.c_int64), ("v_rs", _ct.c_int64), ("o_rs", _ct.c_int64),
                ("scale", _ct.c_float), ("p", _ct.c_float), ("sid", _ct.c_uint32),
                ("d_out", _ct.c_void_p), ("dq", _ct.c_void_p), ("dk", _ct.c_void_p), ("dv", _ct.c_void_p),
                ("dq_rs", _ct.c_int64), ("dk_rs", _ct.c_int64), ("dv_rs", _ct.c_int64),
                ("dbq", _ct.c_void_p), ("dbk", _ct.c_void_p), ("dbv", _ct.c_void_p), ("db_bs", _ct.c_int64),
                ("out8", _ct.c_void_p), ("qscale", _ct.c_void_p), ("amax", _ct.c_void_p), ("amax_slots", _ct.c_int)]


class AttnFwdReq:
    """attention core forward handed to ``functional.drive`` (see LnFwdReq).  Result: ``out``."""

    def __init__(self, q, k, v, mask, B, heads, Sq, Sk, p, rng, sid, emit8=None):
        """``emit8`` = (qscale, amax): 1-element fp32 device tensors; the context is then ALSO written as e4m3
        (``out8``), the operand of the fp8 output projection"""
        d = 64
        H = heads * d
        for t, S in ((q, Sq), (k, Sk), (v, Sk)):
            _chk(t)
            if t.dim() != 2 or t.shape[0] != B * S or t.shape[1] != H or t.stride(1) != 1:
                raise RuntimeError("attn_fwd: bad operand shape %s" % (tuple(t.shape),))
        if mask is not None:
            _c(mask, F32, "mask")
            assert tuple(mask.shape) == (B, Sk)
        self.key = ("attn_fwd", q.dtype)
        self.rng = rng
        self.out = torch.empty((B * Sq, H), device=q.device, dtype=q.dtype)
        self.out8 = None
        self.keep = (q, k, v, mask)
        self.prob = AttnProblem(ptr(q), ptr(k), ptr(v), ptr(mask), ptr(self.out), B, heads, Sq, Sk, q.stride(0), k.stride(0),
                                v.stride(0), H, 0.125, float(p), sid, None, None, None, None, 0, 0, 0, None, None, None)
        if emit8 is not None:
            self.out8 = torch.empty((B * Sq, H), device=q.device, dtype=torch.uint8)
            self.prob.out8, self.prob.qscale, self.prob.amax = ptr(self.out8), ptr(emit8[0]), ptr(emit8[1])
            self.prob.amax_slots = emit8[1].numel() if emit8[1] is not None else 0
            self.keep = self.keep + tuple(emit8)


class AttnBwdReq:
    """attention core backward; gradients are written into the caller's dq/dk/dv views."""

    def __init__(self, q, k, v, mask, d_out, dq, dk, dv, B, heads, Sq, Sk, p, rng, sid, dbq=None, dbk=None, dbv=None,
                 defer=None):
        """``dbq / dbk / dbv``: the fp32 [heads*64] gradients of the query / key / value biases (+=); the kernel leaves
        per-sample partial rows and a reduce job adds them: appended to ``defer`` (the running backward's list) or kept in
        ``self.post`` for ``launch_row_requests`` to run right behind the launch"""
        d = 64
        H = heads * d
        _c(d_out)
        assert tuple(d_out.shape) == (B * Sq, H) and d_out.dtype == q.dtype
        for t, S in ((dq, Sq), (dk, Sk), (dv, Sk)):
            if t.dim() != 2 or t.shape[0] != B * S or t.shape[1] != H or t.stride(1) != 1 or t.dtype != q.dtype:
                raise RuntimeError("attn_bwd: bad gradient buffer %s" % (tuple(t.shape),))
        self.key = ("attn_bwd", q.dtype)
        self.rng = rng
        ws, job = _bias_partials(B, H, q.device, dbq, dbk, dbv)
        self.post = None
        if job is not None:
            if defer is not None:
                defer.append(job)
            else:
                self.post = job
        self.keep = (q, k, v, mask, d_out, dq, dk, dv, dbq, dbk, dbv, ws)
        self.prob = AttnProblem(ptr(q), ptr(k), ptr(v), ptr(mask), None, B, heads, Sq, Sk, q.stride(0), k.stride(0),
                                v.stride(0), H, 0.125, float(p), sid, ptr(d_out), ptr(dq), ptr(dk), ptr(dv), dq.stride(0),
                                dk.stride(0), dv.stride(0),
                                ptr(ws[0, 0]) if (ws is not None and dbq is not None) else None,
                                ptr(ws[0, 1]) if (ws is not None and dbk is not None) else None,
                                ptr(ws[0, 2]) if (ws is not None and dbv is not None) else None, 3 * H)


ROW_REQUESTS = ()  # filled below: the request classes ``functional.drive`` recognises


# ----------------------------------------------------------------------------- row kernels
def ln_fwd(x, bias, residual, gamma, beta, eps, p_pre=0.0, p_post=0.0, rng=None, sid_pre=0, sid_post=0,
           out=None, accumulate=False, out_scale=1.0, save=True):
    """Returns (out, z, stats).  ``z`` overwrites ``x`` in place when ``save``."""
    _c(x)
    M, H = x.shape
    _c(gamma, F32, "gamma"), _c(beta, F32, "beta")
    assert gamma.numel() == H and beta.numel() == H
    if bias is not None:
        _c(bias, F32, "bias")
        assert bias.numel() == H
    if residual is not None:
        _c(residual, x.dtype, "residual")
        assert residual.shape == x.shape
    if out is None:
        out = torch.empty_like(x)
    else:
        assert out.shape == x.shape and out.dtype == x.dtype and out.is_contiguous()
    stats = torch.empty((M, 2), device=x.device, dtype=F32) if save else None
    z = x if save else None
    call("xggm_ln_fwd_" + sfx(x.dtype), ptr(x), ptr(bias), ptr(residual), ptr(gamma), ptr(beta), ptr(out),
         ptr(z), ptr(stats), M, H, float(eps), float(p_pre), float(p_post), ptr(rng), sid_pre, sid_post,
         int(accumulate), float(out_scale), stream())
    return out, z, stats


class LnFwdProblem(_ct.Structure):
    """mirror of ``xggm_ln_fwd_problem`` (include/xggm.h)"""
    _fields_ = [("inp", _ct.c_void_p), ("bias", _ct.c_void_p), ("residual", _ct.c_void_p), ("gamma", _ct.c_void_p),
                ("beta", _ct.c_void_p), ("out", _ct.c_void_p), ("z_out", _ct.c_void_p), ("stats", _ct.c_void_p),
                ("M", _ct.c_int), ("sid_pre", _ct.c_uint32), ("sid_post", _ct.c_uint32), ("in_slabs", _ct.c_int),
                ("out8", _ct.c_void_p), ("qscale", _ct.c_void_p), ("amax", _ct.c_void_p), ("amax_slots", _ct.c_int)]


class LnBwdProblem(_ct.Structure):
    """mirror of ``xggm_ln_bwd_problem`` (include/xggm.h)"""
    _fields_ = [("dy", _ct.c_void_p), ("z", _ct.c_void_p), ("stats", _ct.c_void_p), ("gamma", _ct.c_void_p),
                ("d_in", _ct.c_void_p), ("d_res", _ct.c_void_p), ("dgamma", _ct.c_void_p), ("dbeta", _ct.c_void_p),
                ("dbias", _ct.c_void_p), ("gelu_aux", _ct.c_void_p), ("ws", _ct.c_void_p), ("ws_bytes", _ct.c_size_t),
                ("M", _ct.c_int), ("sid_pre", _ct.c_uint32), ("sid_post", _ct.c_uint32), ("accumulate_dres", _ct.c_int)]


class LnFwdReq:
    """a residual-LayerNorm forward a block generator hands to ``functional.drive``: requests of the
    same round (language + vision stream) are launched together.  Results: ``out``, ``z``, ``stats``."""

    def __init__(self, x, bias, residual, gamma, beta, eps, p_pre=0.0, rng=None, sid_pre=0, dtype=None, emit8=None):
        """``x``: [M, H] activations, or the fp32 split-K partial sums [S, M, H] of ``p_fwd_splitk`` (then
        ``dtype`` = the activation type of out / z / residual).  ``emit8`` = (qscale, amax): ``out`` is also
        written as e4m3 (``out8``), the operand of the next fp8 product."""
        _c(x)
        slabs = 0
        if x.dim() == 3:
            slabs, M, H = x.shape
            if x.dtype != F32 or dtype is None:
                raise RuntimeError("LnFwdReq: split-K input is fp32 [S, M, H] and needs the activation dtype")
        else:
            M, H = x.shape
            dtype = x.dtype
        _c(gamma, F32, "gamma"), _c(beta, F32, "beta")
        assert gamma.numel() == H and beta.numel() == H
        if bias is not None:
            _c(bias, F32, "bias")
            assert bias.numel() == H
        if residual is not None:
            _c(residual, dtype, "residual")
            assert tuple(residual.shape) == (M, H)
        self.key = ("ln_fwd", dtype, H, float(eps), float(p_pre))
        self.rng = rng
        self.out = torch.empty((M, H), device=x.device, dtype=dtype)
        self.z = torch.empty((M, H), device=x.device, dtype=dtype) if slabs else x  # pre-LN sum: in place when it can
        self.stats = torch.empty((M, 2), device=x.device, dtype=F32)
        self.keep = (x, bias, residual, gamma, beta)
        self.prob = LnFwdProblem(ptr(x), ptr(bias), ptr(residual), ptr(gamma), ptr(beta), ptr(self.out), ptr(self.z),
                                 ptr(self.stats), M, sid_pre, 0, slabs)
        self.out8 = None
        if emit8 is not None:
            assert dtype == BF16
            self.out8 = torch.empty((M, H), device=x.device, dtype=torch.uint8)
            self.prob.out8, self.prob.qscale, self.prob.amax = ptr(self.out8), ptr(emit8[0]), ptr(emit8[1])
            self.prob.amax_slots = emit8[1].numel() if emit8[1] is not None else 0
            self.keep = self.keep + tuple(emit8)


class LnBwdReq:
    """backward of LnFwdReq; ``defer`` as in ``ln_bwd``.  Results: ``d_in``, ``d_res``."""

    def __init__(self, dy, z, stats, gamma, dgamma, dbeta, dbias, p_pre=0.0, rng=None, sid_pre=0, defer=None):
        _c(dy), _c(z, dy.dtype)
        M, H = dy.shape
        assert z.shape == dy.shape and tuple(stats.shape) == (M, 2)
        for t in (dgamma, dbeta, dbias):
            if t is not None:
                _c(t, F32, "param grad")
                assert t.numel() == H
        self.key = ("ln_bwd", dy.dtype, H, float(p_pre))
        self.rng = rng
        self.d_in = torch.empty_like(dy)
        self.d_res = torch.empty_like(dy)
        ws, nb = _ws(_lib.lib.xggm_ln_bwd_workspace_bytes(M, H), dy.device)
        now = (dgamma, dbeta, dbias)
        if defer is not None and any(t is not None for t in now):
            defer.append((ws, nb // (12 * H), 3, H, now))
            now = (None, None, None)
        self.keep = (dy, z, stats, gamma, ws)
        self.prob = LnBwdProblem(ptr(dy), ptr(z), ptr(stats), ptr(gamma), ptr(self.d_in), ptr(self.d_res), ptr(now[0]),
                                 ptr(now[1]), ptr(now[2]), None, ptr(ws), nb, M, sid_pre, 0, 0)


class LnSumArgs(_ct.Structure):
    """mirror of ``xggm_ln_sum_args`` (include/xggm.h)"""
    _fields_ = [("inp", _ct.c_void_p * 4), ("gamma", _ct.c_void_p * 4), ("beta", _ct.c_void_p * 4), ("stats", _ct.c_void_p * 4),
                ("sid_post", _ct.c_uint32 * 4), ("out", _ct.c_void_p), ("n", _ct.c_int), ("M", _ct.c_int), ("H", _ct.c_int),
                ("eps", _ct.c_float), ("p_post", _ct.c_float), ("rng", _ct.c_void_p)]


def ln_sum_fwd(xs, gammas, betas, eps, p_post=0.0, rng=None, sids=None):
    """sum_k dropout(LayerNorm(xs[k])) for up to four [M, H] terms in ONE launch (the read-out of the graph blocks).
    Returns (out, [stats_k]); the saved pre-normalisation rows are ``xs`` themselves."""
    n = len(xs)
    assert 1 <= n <= 4 and len(gammas) == n and len(betas) == n
    M, H = xs[0].shape
    a = LnSumArgs()
    stats = []
    for k, (x, g, b) in enumerate(zip(xs, gammas, betas)):
        _c(x, xs[0].dtype), _c(g, F32, "gamma"), _c(b, F32, "beta")
        assert tuple(x.shape) == (M, H) and g.numel() == H and b.numel() == H
        st = torch.empty((M, 2), device=x.device, dtype=F32)
        stats.append(st)
        a.inp[k], a.gamma[k], a.beta[k], a.stats[k] = ptr(x), ptr(g), ptr(b), ptr(st)
        a.sid_post[k] = int(sids[k]) if sids is not None else 0
    out = torch.empty_like(xs[0])
    a.out, a.n, a.M, a.H, a.eps, a.p_post, a.rng = ptr(out), n, M, H, float(eps), float(p_post), ptr(rng)
    call("xggm_ln_sum_fwd_" + sfx(xs[0].dtype), _ct.byref(a), stream())
    return out, stats


def ln_bwd_group(items, p_pre=0.0, p_post=0.0, rng=None, out_scale=1.0):
    """several LayerNorm backwards with one H and one pair of dropout rates in ONE launch (xggm_ln_bwd_grouped_*, four
    problems per launch).  ``items``: dicts of the per-problem arguments of ``ln_bwd`` (dy, z, stats, gamma, dgamma, dbeta,
    dbias, and optionally want_din, want_dres, d_res, sid_pre, sid_post, gelu_aux, defer).  Returns [(d_in, d_res)]."""
    probs, outs, keep = [], [], []
    dt, H = items[0]["dy"].dtype, items[0]["dy"].shape[1]
    for it in items:
        dy, z, stats = it["dy"], it["z"], it["stats"]
        _c(dy, dt), _c(z, dt)
        M = dy.shape[0]
        assert dy.shape[1] == H and z.shape == dy.shape and tuple(stats.shape) == (M, 2)
        now = (it.get("dgamma"), it.get("dbeta"), it.get("dbias"))
        for t in now:
            if t is not None:
                _c(t, F32, "param grad")
                assert t.numel() == H
        d_in = torch.empty_like(dy) if it.get("want_din", True) else None
        d_res = it.get("d_res")
        acc = d_res is not None
        if it.get("want_dres", False) and d_res is None:
            d_res = torch.empty_like(dy)
        if d_res is not None:
            assert d_res.shape == dy.shape and d_res.dtype == dt and d_res.is_contiguous()
        ws, nb = _ws(_lib.lib.xggm_ln_bwd_workspace_bytes(M, H), dy.device)
        defer = it.get("defer")
        if defer is not None and any(t is not None for t in now):
            defer.append((ws, nb // (12 * H), 3, H, now))
            now = (None, None, None)
        probs.append(LnBwdProblem(ptr(dy), ptr(z), ptr(stats), ptr(it["gamma"]), ptr(d_in), ptr(d_res), ptr(now[0]), ptr(now[1]),
                                  ptr(now[2]), ptr(it.get("gelu_aux")), ptr(ws), nb, M, it.get("sid_pre", 0),
                                  it.get("sid_post", 0), int(acc)))
        outs.append((d_in, d_res))
        keep.append(ws)
    arr = (LnBwdProblem * len(probs))(*probs)
    call("xggm_ln_bwd_grouped_" + sfx(dt), _ct.cast(arr, _ct.c_void_p), len(probs), H, float(p_pre), float(p_post), ptr(rng),
         float(out_scale), stream())
    return outs


def prefetch_next(t):
    """queue the bytes of tensor ``t`` (a contiguous slice of a weight buffer) for the next LayerNorm / attention launch to
    read beside its own work and discard (xggm_prefetch_next): the products behind that launch then find their weights
    in the Infinity Cache"""
    if t is not None and t.numel():
        a, n = t.data_ptr(), t.numel() * t.element_size()
        pad = (-a) % 16
        if n > pad + 16:
            call("xggm_prefetch_next", a + pad, n - pad)


def launch_row_requests(reqs):
    """launch LnFwdReq / LnBwdReq objects, grouping those with equal key (same kind, dtype, H, eps, p)."""
    groups = {}
    for r in reqs:
        groups.setdefault(r.key, []).append(r)
    for key, rs in groups.items():
        for r in rs:
            for t in getattr(r, "prefetch", ()):
                prefetch_next(t)
        if key[0] in ("attn_fwd", "attn_bwd"):
            arr = (AttnProblem * len(rs))(*[r.prob for r in rs])
            call("xggm_%s_grouped_%s" % (key[0], sfx(key[1])), _ct.cast(arr, _ct.c_void_p), len(rs), 64, ptr(rs[0].rng),
                 stream())
            reduce_batch([r.post for r in rs if getattr(r, "post", None) is not None])
        elif key[0] == "ln_fwd":
            _, dt, H, eps, p = key
            arr = (LnFwdProblem * len(rs))(*[r.prob for r in rs])
            call("xggm_ln_fwd_grouped_" + sfx(dt), _ct.cast(arr, _ct.c_void_p), len(rs), H, eps, p, 0.0, ptr(rs[0].rng), 0,
                 1.0, stream())
        else:
            _, dt, H, p = key
            arr = (LnBwdProblem * len(rs))(*[r.prob for r in rs])
            call("xggm_ln_bwd_grouped_" + sfx(dt), _ct.cast(arr, _ct.c_void_p), len(rs), H, p, 0.0, ptr(rs[0].rng), 1.0,
                 stream())


ROW_REQUESTS = (LnFwdReq, LnBwdReq, AttnFwdReq, AttnBwdReq)


class ReduceJob(_ct.Structure):
    """mirror of ``xggm_reduce_job`` (include/xggm.h)"""
    _fields_ = [("ws", _ct.c_void_p), ("nblk", _ct.c_int), ("K", _ct.c_int), ("H", _ct.c_int),
                ("target", _ct.c_void_p * 3)]


REDUCE_JOBS_PER_LAUNCH = 72  # MAX_JOBS of rowops.hip: jobs one launch of xggm_partial_reduce_batch takes


def reduce_batch(jobs):
    """second stage of the two-stage parameter-gradient sums: ``jobs`` = [(ws, nblk, K, H, targets)] with ``ws`` fp32
    [nblk][K][H] partial rows and ``targets`` a tuple of K fp32 [H] gradient vectors (None = not wanted):
    targets[k] += sum_b ws[b][k][:], blocks added in index order.  The producers: LayerNorm backwards (K = 3: dgamma,
    dbeta, dbias), GEMM epilogue column sums (K = 1: a bias gradient), attention backwards (K = 3: q / k / v bias)."""
    if not jobs:
        return
    arr = (ReduceJob * len(jobs))()
    for a, (ws, nblk, K, H, tg) in zip(arr, jobs):
        assert 1 <= K <= 3 and len(tg) == K and ws.dtype == F32 and ws.numel() >= nblk * K * H
        a.ws, a.nblk, a.K, a.H = ptr(ws), nblk, K, H
        for k in range(3):
            a.target[k] = ptr(tg[k]) if k < K else None
    call("xggm_partial_reduce_batch", _ct.cast(arr, _ct.c_void_p), len(jobs), stream())


def colsum_rows(M):
    """partial rows the GEMM epilogue writes per column-sum request: one per block of 32 output rows (xggm.h)"""
    return (M + 31) // 32


def ln_bwd(dy, z, stats, gamma, dgamma, dbeta, dbias, want_din=True, want_dres=False, d_res=None,
           p_pre=0.0, p_post=0.0, rng=None, sid_pre=0, sid_post=0, out_scale=1.0, gelu_aux=None, defer=None):
    """Returns (d_in, d_res).  dgamma/dbeta/dbias (fp32, may be None) are accumulated.
    If ``d_res`` is given the residual gradient is ADDED into it.  ``gelu_aux`` = the
    pre-activation u when the LN input was gelu(u): d_in/dbias become grads of u.
    ``defer``: a list; the parameter-gradient sums are then left in the workspace and a job for
    ``reduce_batch`` is appended instead of running the second-stage kernel now."""
    _c(dy), _c(z, dy.dtype)
    M, H = dy.shape
    assert z.shape == dy.shape and tuple(stats.shape) == (M, 2)
    for t in (dgamma, dbeta, dbias):
        if t is not None:
            _c(t, F32, "param grad")
            assert t.numel() == H
    d_in = torch.empty_like(dy) if want_din else None
    acc = d_res is not None
    if want_dres and d_res is None:
        d_res = torch.empty_like(dy)
    if d_res is not None:
        assert d_res.shape == dy.shape and d_res.dtype == dy.dtype and d_res.is_contiguous()
    ws, nb = _ws(_lib.lib.xggm_ln_bwd_workspace_bytes(M, H), dy.device)
    now = (dgamma, dbeta, dbias)
    if defer is not None and any(t is not None for t in now):
        defer.append((ws, nb // (12 * H), 3, H, now))
        now = (None, None, None)
    call("xggm_ln_bwd_" + sfx(dy.dtype), ptr(dy), ptr(z), ptr(stats), ptr(gamma), ptr(d_in), ptr(d_res),
         ptr(now[0]), ptr(now[1]), ptr(now[2]), M, H, float(p_pre), float(p_post), ptr(rng), sid_pre, sid_post,
         float(out_scale), int(acc), ptr(gelu_aux), ptr(ws), nb, stream())
    return d_in, d_res


def _emit8_args(emit8, shape, device):
    """(out8, qscale ptr, amax ptr, slots) of a row kernel that also writes its output as e4m3; ``emit8`` = (qscale,
    amax) 1-element / slot-spread fp32 device tensors, or None"""
    if emit8 is None:
        return None, None, None, 0
    out8 = torch.empty(shape, device=device, dtype=torch.uint8)
    return out8, ptr(emit8[0]), ptr(emit8[1]), (emit8[1].numel() if emit8[1] is not None else 0)


SIDE_MAX, SIDE_ADDITIVE_MASK, SIDE_CAST_BF16 = 3, 1, 2


class _SideJob(_ct.Structure):
    _fields_ = [("kind", _ct.c_int), ("src", _ct.c_void_p), ("dst", _ct.c_void_p), ("count", _ct.c_int64)]


class SideJobs(_ct.Structure):
    """mirror of ``xggm_side_jobs`` (include/xggm.h)"""
    _fields_ = [("n", _ct.c_int), ("job", _SideJob * SIDE_MAX)]


def run_side_jobs(jobs):
    """the stand-alone launches of side jobs nobody carried: [(kind, src, dst)]"""
    for kind, src, dst in jobs:
        if kind == SIDE_ADDITIVE_MASK:
            call("xggm_additive_mask", ptr(src), ptr(dst), src.numel(), stream())
        else:
            call("xggm_cast_from_f32_bf16", ptr(src), ptr(dst), src.numel(), stream())


def embed_fwd(ids, seg, word, pos, typ, gamma, beta, eps, p, rng, sid, emit8=None, side=None):
    """``emit8`` = (qscale, amax): the output is also written as e4m3 (returned as a 4th value).  ``side``: up to three
    (kind, src, dst) pieces of the pass's input glue (SIDE_ADDITIVE_MASK: int64 mask -> fp32 additive mask;
    SIDE_CAST_BF16: fp32 -> bf16) done by workgroups appended to this launch (xggm_embed_fwd_side_*)."""
    B, T = ids.shape
    _c(ids, torch.int64, "input_ids")
    if seg is not None:
        _c(seg, torch.int64, "segment_ids")
        assert seg.shape == ids.shape
    H = word.shape[1]
    for t in (word, pos, typ):
        _c(t)
        assert t.shape[1] == H and t.dtype == word.dtype
    assert pos.shape[0] >= T, "sequence longer than the position table"
    out = torch.empty((B * T, H), device=word.device, dtype=word.dtype)
    z = torch.empty_like(out)
    stats = torch.empty((B * T, 2), device=word.device, dtype=F32)
    out8, q, am, slots = _emit8_args(emit8 if word.dtype == BF16 else None, (B * T, H), word.device)
    if side:
        assert len(side) <= SIDE_MAX
        sj = SideJobs()
        sj.n = len(side)
        for k, (kind, src, dst) in enumerate(side):
            _c(src, torch.int64 if kind == SIDE_ADDITIVE_MASK else F32), _c(dst, F32 if kind == SIDE_ADDITIVE_MASK else BF16)
            assert dst.numel() == src.numel()
            sj.job[k].kind, sj.job[k].src, sj.job[k].dst, sj.job[k].count = kind, ptr(src), ptr(dst), src.numel()
        call("xggm_embed_fwd_side_" + sfx(word.dtype), ptr(ids), ptr(seg), ptr(word), ptr(pos), ptr(typ), ptr(gamma),
             ptr(beta), ptr(out), ptr(z), ptr(stats), B * T, T, H, float(eps), float(p), ptr(rng), sid, ptr(out8), q, am, slots,
             _ct.byref(sj), stream())
    else:
        call("xggm_embed_fwd_" + sfx(word.dtype), ptr(ids), ptr(seg), ptr(word), ptr(pos), ptr(typ), ptr(gamma),
             ptr(beta), ptr(out), ptr(z), ptr(stats), B * T, T, H, float(eps), float(p), ptr(rng), sid, ptr(out8), q, am, slots,
             stream())
    if emit8 is not None:
        return out, z, stats, out8
    return out, z, stats


def embed_bwd(ids, seg, dy, z, stats, gamma, dword, dpos, dtyp, dgamma, dbeta, p, rng, sid, row_list=None):
    """``row_list`` = (ids int64 [cap], sq fp32 [cap], n int32 [1]), cap >= M: the kernel also lists the word-table rows it
    touched, |row|^2 of the gradient per owner and the count (xggm_embed_bwd_listed_*)"""
    B, T = ids.shape
    M, H = dy.shape
    _c(dy), _c(z, dy.dtype)
    for t in (dword, dpos, dtyp, dgamma, dbeta):
        _c(t, F32, "embedding grad")
    dz = torch.empty_like(dy)
    ws, nb = _ws(_lib.lib.xggm_ln_bwd_workspace_bytes(M, H), dy.device)
    if row_list is None:
        call("xggm_embed_bwd_" + sfx(dy.dtype), ptr(ids), ptr(seg), ptr(dy), ptr(z), ptr(stats), ptr(gamma),
             ptr(dz), ptr(dword), ptr(dpos), ptr(dtyp), ptr(dgamma), ptr(dbeta), M, T, H, float(p), ptr(rng), sid,
             ptr(ws), nb, stream())
        return
    rid, rsq, rn = row_list
    _c(rid, torch.int64), _c(rsq, F32), _c(rn, torch.int32)
    assert rid.numel() >= M and rsq.numel() >= M
    call("xggm_embed_bwd_listed_" + sfx(dy.dtype), ptr(ids), ptr(seg), ptr(dy), ptr(z), ptr(stats), ptr(gamma),
         ptr(dz), ptr(dword), ptr(dpos), ptr(dtyp), ptr(dgamma), ptr(dbeta), M, T, H, float(p), ptr(rng), sid,
         ptr(ws), nb, ptr(rid), ptr(rsq), ptr(rn), stream())


def visn_embed_fwd(u, bf, boxes, Wb, bb, g1, b1, g2, b2, eps, p, rng, sid, emit8=None):
    """``emit8`` = (qscale, amax): the output is also written as e4m3 (returned as a 5th value)"""
    _c(u), _c(boxes, u.dtype, "boxes")
    M, H = u.shape
    assert tuple(boxes.shape) == (M, 4) and tuple(Wb.shape) == (H, 4)
    for t in (bf, Wb, bb, g1, b1, g2, b2):
        _c(t, F32, "visn_fc parameter")
    out = torch.empty_like(u)
    z2 = torch.empty_like(u)
    stats = torch.empty((M, 4), device=u.device, dtype=F32)
    out8, q, am, slots = _emit8_args(emit8 if u.dtype == BF16 else None, (M, H), u.device)
    call("xggm_visn_embed_fwd_" + sfx(u.dtype), ptr(u), ptr(bf), ptr(boxes), ptr(Wb), ptr(bb), ptr(g1),
         ptr(b1), ptr(g2), ptr(b2), ptr(out), ptr(u), ptr(z2), ptr(stats), M, H, float(eps), float(p), ptr(rng),
         sid, ptr(out8), q, am, slots, stream())
    if emit8 is not None:
        return out, u, z2, stats, out8
    return out, u, z2, stats


def visn_embed_bwd(dy, z1, z2, stats, boxes, g1, g2, grads, p, rng, sid):
    """grads: dict with fp32 accumulators dbf, dg1, db1, dWb, dbb, dg2, db2.  Returns du."""
    _c(dy)
    M, H = dy.shape
    du = torch.empty_like(dy)
    ws, nb = _ws(_lib.lib.xggm_visn_embed_bwd_workspace_bytes(M, H), dy.device)
    call("xggm_visn_embed_bwd_" + sfx(dy.dtype), ptr(dy), ptr(z1), ptr(z2), ptr(stats), ptr(boxes), ptr(g1),
         ptr(g2), ptr(du), ptr(grads["dbf"]), ptr(grads["dg1"]), ptr(grads["db1"]), ptr(grads["dWb"]),
         ptr(grads["dbb"]), ptr(grads["dg2"]), ptr(grads["db2"]), M, H, float(p), ptr(rng), sid, ptr(ws), nb,
         stream())
    return du


# ----------------------------------------------------------------------------- graph kernels
def aggregate(Mx, x, mode=AGG_PLAIN, scale=1.0, scale_ptr=None, self_w=0.0, out=None):
    """out = [out +] self_w*x + scale*(1+*scale_ptr) * M' @ x.  Mx fp32 [B,N,N], x T [B,N,H]."""
    _c(Mx, F32, "adjacency"), _c(x)
    B, N, H = x.shape
    assert tuple(Mx.shape) == (B, N, N), "adjacency %s vs nodes %s" % (tuple(Mx.shape), tuple(x.shape))
    acc = out is not None
    if out is None:
        out = torch.empty_like(x)
    else:
        assert out.shape == x.shape and out.dtype == x.dtype and out.is_contiguous()
    call("xggm_aggregate_" + sfx(x.dtype), ptr(Mx), ptr(x), ptr(out), B, N, H, mode, float(scale),
         ptr(scale_ptr), float(self_w), int(acc), stream())
    return out


def agg_residual_ln(Mx, y, res, gamma, beta, eps):
    """LayerNorm(res + Mx @ y) per sample in ONE launch (bf16 [B, N, H], H in 64 / 128 / 256 / 768, N <= 64): GCNConv's
    tail once y = x W^T has been taken (xggm_agg_residual_ln_bf16).  Returns (out, z, stats) as ``ln_fwd`` does."""
    _c(Mx, F32, "adjacency"), _c(y, BF16), _c(res, BF16), _c(gamma, F32, "gamma"), _c(beta, F32, "beta")
    B, N, H = y.shape
    assert tuple(Mx.shape) == (B, N, N) and res.shape == y.shape and gamma.numel() == H and beta.numel() == H
    out, z = torch.empty_like(y), torch.empty_like(y)
    stats = torch.empty((B * N, 2), device=y.device, dtype=F32)
    call("xggm_agg_residual_ln_bf16", ptr(Mx), ptr(y), ptr(res), ptr(gamma), ptr(beta), ptr(out), ptr(z), ptr(stats), B, N, H,
         float(eps), stream())
    return out, z, stats


def agg_residual_ln_ok(x, N):
    """shapes the fused GCNConv tail is built for"""
    return x.dtype == BF16 and x.shape[-1] in (64, 128, 256, 768) and N <= 64


def agg_dot(Mx, x, dh, out):
    B, N, H = x.shape
    _c(Mx, F32), _c(x), _c(dh, x.dtype), _c(out, F32)
    assert tuple(Mx.shape) == (B, N, N) and dh.shape == x.shape
    call("xggm_agg_dot_" + sfx(x.dtype), ptr(Mx), ptr(x), ptr(dh), ptr(out), B, N, H, ptr(sum_ws(x.device)), stream())


def adj_regen_fwd(S):
    _c(S, F32, "S")
    B, N, _ = S.shape
    adj = torch.empty_like(S)
    colmax = torch.empty((B, N), device=S.device, dtype=F32)
    argmax = torch.empty((B, N), device=S.device, dtype=torch.int32)
    call("xggm_adj_regen_fwd", ptr(S), ptr(adj), ptr(colmax), ptr(argmax), B, N, stream())
    return adj, colmax, argmax


def adj_regen_bwd(d_adj, S, adj, colmax, argmax):
    _c(d_adj, F32, "d_adj")
    B, N, _ = S.shape
    assert d_adj.shape == S.shape
    dS = torch.empty_like(S)
    call("xggm_adj_regen_bwd", ptr(d_adj), ptr(S), ptr(adj), ptr(colmax), ptr(argmax), ptr(dS), B, N, stream())
    return dS


def adj_init_fwd(e, N, sigma, randn=None, rng=None, sid=0, B=None, want_gradlog=True):
    if e is not None:
        _c(e, F32, "encoder_adj output")
        B = e.shape[0]
        assert e.shape[1] == N * (N - 1) // 2, "encoder_adj width %d != N(N-1)/2" % e.shape[1]
    dev = e.device if e is not None else (randn.device if randn is not None else rng.device)
    if randn is not None:
        _c(randn, F32, "randn")
        assert tuple(randn.shape) == (B, N, N)
    adj = torch.empty((B, N, N), device=dev, dtype=F32)
    g = torch.empty_like(adj) if want_gradlog else None
    call("xggm_adj_init_fwd", ptr(e), ptr(randn), ptr(adj), ptr(g), B, N, float(sigma), ptr(rng), sid, stream())
    return adj, g


def adj_init_bwd(d_adj):
    _c(d_adj, F32, "d_adj")
    B, N, _ = d_adj.shape
    d_e = torch.empty((B, N * (N - 1) // 2), device=d_adj.device, dtype=F32)
    call("xggm_adj_init_bwd", ptr(d_adj), ptr(d_e), B, N, stream())
    return d_e


def feature_noise(x, sigma, randn=None, rng=None, sid=0):
    _c(x)
    if randn is not None:
        _c(randn, F32, "randn")
        assert randn.shape == x.shape
    out = torch.empty_like(x)
    g = torch.empty(x.shape, device=x.device, dtype=F32)
    call("xggm_feature_noise_" + sfx(x.dtype), ptr(x), ptr(randn), ptr(out), ptr(g), x.numel(), float(sigma),
         ptr(rng), sid, stream())
    return out, g


def pool_concat_fwd(x, nodes):
    _c(x), _c(nodes, x.dtype)
    B, N, H = nodes.shape
    assert tuple(x.shape) == (B, H)
    out = torch.empty((B, 2 * H), device=x.device, dtype=x.dtype)
    call("xggm_pool_concat_fwd_" + sfx(x.dtype), ptr(x), ptr(nodes), ptr(out), B, N, H, stream())
    return out


def pool_concat_bwd(d_out, out, N):
    _c(d_out), _c(out, d_out.dtype)
    B, H2 = out.shape
    H = H2 // 2
    dx = torch.empty((B, H), device=out.device, dtype=out.dtype)
    dn = torch.empty((B, N, H), device=out.device, dtype=out.dtype)
    call("xggm_pool_concat_bwd_" + sfx(out.dtype), ptr(d_out), ptr(out), ptr(dx), ptr(dn), B, N, H, 0, stream())
    return dx, dn


def bcast_rows(x, N):
    _c(x)
    B, H = x.shape
    out = torch.empty((B, N, H), device=x.device, dtype=x.dtype)
    call("xggm_bcast_rows_" + sfx(x.dtype), ptr(x), ptr(out), B, N, H, stream())
    return out


def sum_rows(g):
    _c(g)
    B, N, H = g.shape
    out = torch.empty((B, H), device=g.device, dtype=g.dtype)
    call("xggm_sum_rows_" + sfx(g.dtype), ptr(g), ptr(out), B, N, H, stream())
    return out


# ----------------------------------------------------------------------------- losses
def zeros_f32(n, device):
    """n zeroed floats from ONE custom launch (no framework fill kernel inside a captured pass)"""
    m = (n + 3) // 4 * 4
    t = torch.empty(m, device=device, dtype=F32)
    zero_ranges(t, [(0, m)])
    return t[:n]


def _scalar(out, device):
    """the 0-dim accumulator of a loss kernel: a caller-provided zeroed slot, or a fresh zeroed one"""
    return out.view(()) if out is not None else zeros_f32(1, device).view(())


SUM_WS_FLOATS = 4104  # XGGM_SUM_WS_FLOATS (include/xggm.h)
_SUM_WS = {}
_SUM_WS_HOME = {}


def sum_ws(device):
    """workspace of the grid-wide sums without atomics (losses, GIN's eps gradient): per-workgroup partials + an
    arrival counter every kernel leaves at zero, so ONE zeroed buffer per device serves every launch -- the package
    launches all its kernels on one stream, and launches of one stream run in order.  Created on first use;
    ``functional.Runtime`` touches it at construction so that it exists before any graph capture."""
    device = torch.device(device)
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    ws = _SUM_WS.get(device)
    if ws is None:
        ws = _SUM_WS[device] = torch.zeros(SUM_WS_FLOATS, device=device, dtype=F32)
        _SUM_WS_HOME[device] = torch.cuda.current_stream(device).cuda_stream
        return ws
    # launches on ANOTHER stream (an evaluation beside the training stream, tools/exp_two_streams.py) may overlap the
    # home stream's: two kernels sharing the arrival counter corrupt it for good (no workgroup ever draws the last ticket
    # again).  Every other stream gets a workspace of its own; a capture keeps the home one (its graph replays where the
    # trainer runs, and allocating inside a capture would tie the buffer to the graph's pool).
    sid = torch.cuda.current_stream(device).cuda_stream
    if sid == _SUM_WS_HOME[device] or torch.cuda.is_current_stream_capturing():
        return ws
    key = (device, sid)
    side = _SUM_WS.get(key)
    if side is None:
        side = _SUM_WS[key] = torch.zeros(SUM_WS_FLOATS, device=device, dtype=F32)
    return side


def dsm_fwd(s, g, coef, out=None):
    _c(s), _c(g, F32, "grad_log_noise")
    assert s.shape == g.shape
    loss = _scalar(out, s.device)
    call("xggm_dsm_loss_fwd_" + sfx(s.dtype), ptr(s), ptr(g), ptr(loss), s.numel(), float(coef), ptr(sum_ws(s.device)),
         stream())
    return loss


def dsm_bwd(s, g, gout, coef):
    ds = torch.empty_like(s)
    call("xggm_dsm_loss_bwd_" + sfx(s.dtype), ptr(s), ptr(g), ptr(gout), ptr(ds), s.numel(), float(coef), stream())
    return ds


def symkl_fwd(x, y, coef, out=None):
    _c(x), _c(y, x.dtype)
    assert x.shape == y.shape
    W = x.shape[-1]
    loss = _scalar(out, x.device)
    call("xggm_symkl_" + sfx(x.dtype), ptr(x), ptr(y), ptr(loss), None, None, None, x.numel() // W, W,
         float(coef), 0, ptr(sum_ws(x.device)), stream())
    return loss


def symkl_bwd(x, y, gout, coef, need_x, need_y):
    W = x.shape[-1]
    dx = torch.empty_like(x) if need_x else None
    dy = torch.empty_like(y) if need_y else None
    call("xggm_symkl_" + sfx(x.dtype), ptr(x), ptr(y), None, ptr(gout), ptr(dx), ptr(dy), x.numel() // W, W,
         float(coef), 0, None, stream())
    return dx, dy


def bce_fwd(logit, target, coef, out=None):
    _c(logit, F32, "logit"), _c(target, F32, "target")
    assert logit.shape == target.shape
    loss = _scalar(out, logit.device)
    call("xggm_bce_fwd", ptr(logit), ptr(target), ptr(loss), logit.numel(), float(coef), ptr(sum_ws(logit.device)), stream())
    return loss


def bce_bwd(logit, target, gout, coef, dt):
    dl = torch.empty(logit.shape, device=logit.device, dtype=dt)
    call("xggm_bce_bwd_" + sfx(dt), ptr(logit), ptr(target), ptr(gout), ptr(dl), logit.numel(), float(coef), stream())
    return dl


# ----------------------------------------------------------------------------- optimiser / utils
def zero_ranges(buf, ranges, rows=None):
    """zero buf[s:e] for up to 16 (s, e) element ranges per launch (all float4-aligned).  ``rows`` = (table [R, H] fp32,
    ids int64 [cap], n int32 [1]): the first launch also zeroes rows ids[:n] of the table (device-side list, as
    ``embed_bwd(..., row_list=...)`` left it)."""
    _c(buf, F32)
    rs = [(s, e) for s, e in ranges if e > s]
    for i in range(0, max(len(rs), 1 if rows is not None else 0), 16):
        chunk = rs[i:i + 16]
        offs = (_ct.c_int64 * max(len(chunk), 1))(*[s for s, _ in chunk])
        lens = (_ct.c_int64 * max(len(chunk), 1))(*[e - s for s, e in chunk])
        if rows is not None and i == 0:
            table, ids, n = rows
            _c(table, F32), _c(ids, torch.int64), _c(n, torch.int32)
            assert table.dim() == 2
            call("xggm_zero_ranges_rows_f32", ptr(buf), _ct.cast(offs, _ct.c_void_p), _ct.cast(lens, _ct.c_void_p), len(chunk),
                 ptr(table), ptr(ids), ptr(n), ids.numel(), table.shape[0], table.shape[1], stream())
        else:
            call("xggm_zero_ranges_f32", ptr(buf), _ct.cast(offs, _ct.c_void_p), _ct.cast(lens, _ct.c_void_p), len(chunk), stream())


_SQNORM_WS = {}


def sqnorm(g, out):
    """out += sum g^2 (fixed summation order: see xggm.h)."""
    _c(g, F32), _c(out, F32)
    ws = _SQNORM_WS.get(g.device)
    if ws is None:  # zeroed once; the kernel leaves its arrival counter at zero
        ws = _SQNORM_WS[g.device] = torch.zeros(4100, device=g.device, dtype=F32)
    call("xggm_sqnorm_f32", ptr(g), g.numel(), ptr(out), ptr(ws), stream())


def sqnorm_multi(buf, spans, out, norm=None, overwrite=True, square=True, mul=1.0):
    """out (1 fp32) = [out +] sum over the (start, end) element ranges ``spans`` of buf^2 (``square`` False: of buf),
    fixed summation order; ``norm`` (1 fp32 or None) = sqrt(out).  Any number of ranges (16 per launch pair; none:
    only seeds / finishes)."""
    _c(buf, F32), _c(out, F32)
    if norm is not None:
        _c(norm, F32)
    ws = _SQNORM_WS.get(buf.device)
    if ws is None:
        ws = _SQNORM_WS[buf.device] = torch.zeros(4100, device=buf.device, dtype=F32)
    rs = [(s, e) for s, e in spans if e > s]
    chunks = [rs[i:i + 16] for i in range(0, len(rs), 16)] or [[]]
    for ci, ch in enumerate(chunks):
        offs = (_ct.c_int64 * max(len(ch), 1))(*[s for s, _ in ch])
        lens = (_ct.c_int64 * max(len(ch), 1))(*[e - s for s, e in ch])
        last = ci == len(chunks) - 1
        call("xggm_sqnorm_multi_f32", ptr(buf), _ct.cast(offs, _ct.c_void_p), _ct.cast(lens, _ct.c_void_p), len(ch), ptr(out),
             ptr(norm) if last else None, ptr(ws), int(overwrite and ci == 0), int(square), float(mul) if last else 1.0, stream())


class _PassTail(_ct.Structure):
    _fields_ = [("steps", _ct.c_void_p), ("lr_scale", _ct.c_void_p), ("index", _ct.c_void_p), ("t_total", _ct.c_void_p),
                ("warmup", _ct.c_void_p), ("n", _ct.c_int), ("rng", _ct.c_void_p), ("rng_by", _ct.c_uint64)]


CLIP_NORM_MAX_SPANS = 24
CLIP_NORM_MAX_SCHED = 16


def _pass_tail(sched, rng):
    """(xggm_pass_tail or None, objects to keep alive) for clip_norm / clip_norm_bf16"""
    if sched is None and rng is None:
        return None, ()
    tail, keep = _PassTail(), []
    if sched is not None:
        steps, lr_scale, entries = sched
        _c(steps, torch.int64), _c(lr_scale, F32)
        n = len(entries)
        assert n <= CLIP_NORM_MAX_SCHED
        idx = (_ct.c_int * max(n, 1))(*[int(e[0]) for e in entries])
        tt = (_ct.c_int64 * max(n, 1))(*[int(e[1]) for e in entries])
        wu = (_ct.c_float * max(n, 1))(*[float(e[2]) for e in entries])
        keep += [idx, tt, wu]
        tail.steps, tail.lr_scale, tail.n = ptr(steps), ptr(lr_scale), n
        tail.index, tail.t_total, tail.warmup = (_ct.cast(a, _ct.c_void_p) for a in (idx, tt, wu))
    if rng is not None:
        tail.rng, tail.rng_by = ptr(rng[0]), int(rng[1])
    return tail, keep


def clip_norm_bf16(g, spans, out, norm=None, accumulate=False, mul=1.0, sched=None, rng=None):
    """out (1 fp32) = ((out if accumulate else 0) + sum over the (start, end) ranges of the bf16 buffer g of g^2) * mul,
    ``norm`` = sqrt(out): the norm pass over the data-parallel wire arena in two launches (at most 24 ranges, starts
    multiples of 8); ``sched`` / ``rng`` as in ``clip_norm``."""
    _c(g, BF16), _c(out, F32)
    rs = [(s, e) for s, e in spans if e > s]
    assert len(rs) <= CLIP_NORM_MAX_SPANS
    ws = _SQNORM_WS.get(g.device)
    if ws is None:
        ws = _SQNORM_WS[g.device] = torch.zeros(4100, device=g.device, dtype=F32)
    offs = (_ct.c_int64 * max(len(rs), 1))(*[s for s, _ in rs])
    lens = (_ct.c_int64 * max(len(rs), 1))(*[e - s for s, e in rs])
    tail, keep = _pass_tail(sched, rng)
    call("xggm_clip_norm_bf16", ptr(g), _ct.cast(offs, _ct.c_void_p), _ct.cast(lens, _ct.c_void_p), len(rs), ptr(out),
         ptr(norm) if norm is not None else None, ptr(ws), int(accumulate), float(mul),
         _ct.byref(tail) if tail is not None else None, stream())


def clip_norm(g, spans, slots, slot_spans, out, norm=None, mul=1.0, sched=None, rng=None):
    """out (1 fp32) = mul * (sum over the (start, end) ranges ``spans`` of g^2 + sum over ``slot_spans`` of slots), fixed
    summation order, ``norm`` (1 fp32 or None) = sqrt(out): the whole norm of a pass in two launches.  The finishing
    launch also takes ``sched`` = (steps, lr_scale, [(index, t_total, warmup)]) (sched_step_multi) and ``rng`` =
    (state, by) (rng_advance) along.  At most 24 ranges and 16 schedule entries."""
    _c(g, F32), _c(out, F32)
    rs = [(s, e) for s, e in spans if e > s]
    ss = [(s, e) for s, e in slot_spans if e > s]
    assert len(rs) + len(ss) <= CLIP_NORM_MAX_SPANS
    if ss:
        _c(slots, F32)
    ws = _SQNORM_WS.get(g.device)
    if ws is None:
        ws = _SQNORM_WS[g.device] = torch.zeros(4100, device=g.device, dtype=F32)

    def arrs(r):
        return ((_ct.c_int64 * max(len(r), 1))(*[s for s, _ in r]), (_ct.c_int64 * max(len(r), 1))(*[e - s for s, e in r]))

    o1, l1 = arrs(rs)
    o2, l2 = arrs(ss)
    tail, keep = _pass_tail(sched, rng)
    call("xggm_clip_norm_f32", ptr(g), _ct.cast(o1, _ct.c_void_p), _ct.cast(l1, _ct.c_void_p), len(rs),
         ptr(slots) if ss else None, _ct.cast(o2, _ct.c_void_p), _ct.cast(l2, _ct.c_void_p), len(ss), ptr(out),
         ptr(norm) if norm is not None else None, ptr(ws), float(mul), _ct.byref(tail) if tail is not None else None, stream())


def additive_mask(mask):
    """(1 - mask) * -10000 as fp32, for an int64 token mask [B, S]"""
    _c(mask, torch.int64, "attention mask")
    out = torch.empty(mask.shape, device=mask.device, dtype=F32)
    call("xggm_additive_mask", ptr(mask), ptr(out), mask.numel(), stream())
    return out


def zero_diag(adj):
    """adj.triu(1) + adj.tril(-1) for fp32 [B, N, N]"""
    _c(adj, F32, "adjacency")
    B, N, N2 = adj.shape
    assert N == N2
    out = torch.empty_like(adj)
    call("xggm_zero_diag_f32", ptr(adj), ptr(out), B, N, stream())
    return out


def add_n(terms, out=None):
    """sum of 2..4 same-shaped contiguous tensors (fp32 or bf16) in ONE launch, fp32 arithmetic; ``out`` may be one of
    them"""
    assert 2 <= len(terms) <= 4
    t0 = terms[0]
    for t in terms:
        assert t.is_cuda and t.dtype == t0.dtype and t.shape == t0.shape and t.is_contiguous()
    if out is None:
        out = torch.empty_like(t0)
    ps = [ptr(t) for t in terms] + [None] * (4 - len(terms))
    call("xggm_add_n_f32" if t0.dtype == F32 else "xggm_add_n_bf16", ps[0], ps[1], ps[2], ps[3], ptr(out), t0.numel(),
         stream())
    return out


def pad_rows(x, ld):
    """[rows, n] fp32 / bf16 -> bf16 [rows, ld] with zero columns n.. (one launch); returns the [rows, n] VIEW with row
    stride ld"""
    assert x.dim() == 2 and x.is_cuda and x.is_contiguous() and x.dtype in (F32, BF16) and ld >= x.shape[1]
    out = torch.empty((x.shape[0], ld), device=x.device, dtype=BF16)
    call("xggm_pad_rows_bf16", ptr(x), int(x.dtype == F32), ptr(out), x.shape[0], x.shape[1], ld, stream())
    return out[:, :x.shape[1]]


def gather_rows(table_f32, idx, out=None):
    """out[i] = bf16(table[idx[i]]): ``table`` fp32 [V, H] (contiguous), ``idx`` int64 [n] on the device"""
    V, H = table_f32.shape
    n = idx.numel()
    if out is None:
        out = torch.empty((n, H), device=table_f32.device, dtype=BF16)
    call("xggm_gather_rows_bf16", ptr(table_f32), ptr(idx), ptr(out), n, H, V, stream())
    return out


def scatter_rows(rows_bf16, idx, table_bf16):
    """table[idx[i]] = rows[i] (duplicate indices carry identical rows)"""
    V, H = table_bf16.shape
    call("xggm_scatter_rows_bf16", ptr(rows_bf16), ptr(idx), ptr(table_bf16), idx.numel(), H, V, stream())


def add_scalars(terms):
    """sum of up to four 0-dim fp32 device tensors"""
    ts = [_c(t, F32) for t in terms]
    assert 1 <= len(ts) <= 4
    out = torch.empty((), device=ts[0].device, dtype=F32)
    ps = [ptr(t) for t in ts] + [None] * (4 - len(ts))
    call("xggm_add_scalars_f32", ps[0], ps[1], ps[2], ps[3], ptr(out), stream())
    return out


def bertadam(p, g, m, v, shadow, sqn, max_norm, lr, lr_scale, b1, b2, eps, wd):
    for t in (p, g, m, v):
        _c(t, F32)
        assert t.numel() == p.numel()
    if shadow is not None:
        _c(shadow, BF16)
        assert shadow.numel() == p.numel()
    call("xggm_bertadam_f32", ptr(p), ptr(g), ptr(m), ptr(v), ptr(shadow), p.numel(), ptr(sqn), float(max_norm),
         float(lr), ptr(lr_scale), float(b1), float(b2), float(eps), float(wd), stream())


class AdamArgs(_ct.Structure):
    """mirror of ``xggm_adam_args`` (include/xggm.h)"""
    _fields_ = [("p", _ct.c_void_p), ("g", _ct.c_void_p), ("m", _ct.c_void_p), ("v", _ct.c_void_p),
                ("shadow_bf16", _ct.c_void_p), ("n", _ct.c_int64), ("sqnorm", _ct.c_void_p), ("max_norm", _ct.c_float),
                ("lr", _ct.c_float), ("lr_dev", _ct.c_void_p), ("lr_scale", _ct.c_void_p),
                ("b1", _ct.c_float), ("b2", _ct.c_float), ("eps", _ct.c_float), ("weight_decay", _ct.c_float),
                ("g_bf16", _ct.c_int), ("shadow8", _ct.c_void_p), ("w8_id", _ct.c_void_p), ("w8_qscale", _ct.c_void_p),
                ("w8_amax", _ct.c_void_p), ("elem0", _ct.c_int64), ("g_scale", _ct.c_float), ("w8_amax_slots", _ct.c_int)]


def bertadam_multi(jobs):
    """``jobs``: argument tuples of ``bertadam_ex`` (positional, then a dict of its keyword arguments): ONE launch for all
    of them (xggm_bertadam_multi), in chunks by gradient type"""
    if not jobs:
        return
    structs = [_adam_args(*a, **kw) for a, kw in jobs]
    for bf in (0, 1):
        chunk = [a for a in structs if a.g_bf16 == bf]
        if chunk:
            arr = (AdamArgs * len(chunk))(*chunk)
            call("xggm_bertadam_multi", _ct.cast(arr, _ct.c_void_p), len(chunk), stream())


def bertadam_ex(p, g, m, v, shadow, sqn, max_norm, lr, lr_scale, b1, b2, eps, wd, lr_dev=None, w8=None, elem0=0, g_scale=1.0):
    """the update with device-resident lr (``lr_dev``), bf16 gradients (``g.dtype``) and/or the e4m3 weight copy
    ``w8`` = (shadow8 slice, id table, qscale table, amax table); ``elem0``: arena offset of p[0]."""
    a = _adam_args(p, g, m, v, shadow, sqn, max_norm, lr, lr_scale, b1, b2, eps, wd, lr_dev=lr_dev, w8=w8, elem0=elem0,
                   g_scale=g_scale)
    call("xggm_bertadam_ex", _ct.byref(a), stream())


def _adam_args(p, g, m, v, shadow, sqn, max_norm, lr, lr_scale, b1, b2, eps, wd, lr_dev=None, w8=None, elem0=0, g_scale=1.0):
    for t in (p, m, v):
        _c(t, F32)
        assert t.numel() == p.numel()
    _c(g)
    assert g.numel() == p.numel() and g.dtype in (F32, BF16)
    a = AdamArgs(ptr(p), ptr(g), ptr(m), ptr(v), ptr(shadow), p.numel(), ptr(sqn), float(max_norm), float(lr), ptr(lr_dev),
                 ptr(lr_scale), float(b1), float(b2), float(eps), float(wd), int(g.dtype == BF16), None, None, None, None,
                 int(elem0), float(g_scale), 1)
    if w8 is not None:
        s8, ids, q, amax = w8[:4]
        assert s8.numel() == p.numel() and s8.element_size() == 1 and ids.dtype == torch.int16
        a.shadow8, a.w8_id, a.w8_qscale, a.w8_amax = ptr(s8), ptr(ids), ptr(q), ptr(amax)
        a.w8_amax_slots = int(w8[4]) if len(w8) > 4 else 1  # floats per entry of the amax table
    return a


def sqnorm_bf16(g, out):
    _c(g, BF16), _c(out, F32)
    ws = _SQNORM_WS.get(g.device)
    if ws is None:
        ws = _SQNORM_WS[g.device] = torch.zeros(4100, device=g.device, dtype=F32)
    call("xggm_sqnorm_bf16", ptr(g), g.numel(), ptr(out), ptr(ws), stream())


def fp8_scale_update(amax, hist, qscale, dscale, pos, i0, n, hist_len, margin, shrink, bump, slots=1):
    """entries [i0, i0 + n) of a scale table (xggm_fp8_scale_update); ``slots``: floats every amax entry is spread over"""
    for t in (amax, hist, qscale, dscale):
        _c(t, F32)
    _c(pos, torch.int64)
    call("xggm_fp8_scale_update", amax.data_ptr() + 4 * i0 * slots, hist.data_ptr() + 4 * i0 * hist_len,
         qscale.data_ptr() + 4 * i0, dscale.data_ptr() + 4 * i0, ptr(pos), n, hist_len, float(margin), int(shrink),
         int(bump), int(slots), stream())


def sched_step(step, lr_scale, t_total, warmup):
    _c(step, torch.int64), _c(lr_scale, F32)
    call("xggm_sched_step", ptr(step), ptr(lr_scale), int(t_total), float(warmup), stream())


def sched_step_multi(steps, lr_scale, entries):
    """``entries``: [(index, t_total, warmup)] -- one launch for all of them (groups of 16)."""
    _c(steps, torch.int64), _c(lr_scale, F32)
    for i in range(0, len(entries), 16):
        ch = entries[i:i + 16]
        n = len(ch)
        idx = (_ct.c_int * n)(*[int(e[0]) for e in ch])
        tt = (_ct.c_int64 * n)(*[int(e[1]) for e in ch])
        wu = (_ct.c_float * n)(*[float(e[2]) for e in ch])
        call("xggm_sched_step_multi", ptr(steps), ptr(lr_scale), _ct.cast(idx, _ct.c_void_p), _ct.cast(tt, _ct.c_void_p),
             _ct.cast(wu, _ct.c_void_p), n, stream())


def rng_advance(rng, by=1):
    call("xggm_rng_advance", ptr(rng), int(by), stream())


def scale(x, s=1.0, scale_ptr=None):
    """s * (1 + *scale_ptr) * x"""
    _c(x)
    out = torch.empty_like(x)
    call("xggm_scale_" + sfx(x.dtype), ptr(x), ptr(out), x.numel(), float(s), ptr(scale_ptr), stream())
    return out


def sigmoid_bwd(dy, y, dt):
    _c(dy, F32), _c(y, F32)
    out = torch.empty(y.shape, device=y.device, dtype=dt)
    call("xggm_sigmoid_bwd_" + sfx(dt), ptr(dy), ptr(y), ptr(out), y.numel(), stream())
    return out


def tanh_bwd(dy, y):
    _c(dy), _c(y, dy.dtype)
    out = torch.empty_like(y)
    call("xggm_tanh_bwd_" + sfx(y.dtype), ptr(dy), ptr(y), ptr(out), y.numel(), stream())
    return out


def cast_from_f32(x, dt):
    """fp32 -> T copy (identity copy for T = fp32 is skipped)."""
    _c(x, F32)
    if dt == F32:
        return x
    out = torch.empty(x.shape, device=x.device, dtype=dt)
    call("xggm_cast_from_f32_" + sfx(dt), ptr(x), ptr(out), x.numel(), stream())
    return out


def cast_bf16(x, out):
    _c(x, F32), _c(out, BF16)
    assert x.numel() == out.numel()
    if x.numel():
        call("xggm_cast_f32_to_bf16", ptr(x), ptr(out), x.numel(), stream())


def dropout_mask(n, p, rng, sid, device):
    out = torch.empty(n, device=device, dtype=F32)
    call("xggm_dropout_mask", ptr(out), n, float(p), ptr(rng), sid, stream())
    return out


def normal(n, rng, sid, device):
    out = torch.empty(n, device=device, dtype=F32)
    call("xggm_normal", ptr(out), n, ptr(rng), sid, stream())
    return out


def make_rng(seed, device):
    """device-resident {seed, offset} pair (int64 storage of the two uint64 words)."""
    return torch.tensor([int(seed) & 0x7FFFFFFFFFFFFFFF, 0], dtype=torch.int64, device=device)


def triu_index(k, N):
    import ctypes
    i, j = ctypes.c_int(), ctypes.c_int()
    _lib.check(_lib.lib.xggm_triu_index(k, N, ctypes.byref(i), ctypes.byref(j)), "xggm_triu_index")
    return i.value, j.value


# ----------------------------------------------------------------------------- graph attention
def gat_att_fwd(s, adj, alpha):
    _c(s, F32, "s"), _c(adj, F32, "adj")
    B, N, _ = adj.shape
    assert tuple(s.shape) == (B * N, 2)
    att = torch.empty_like(adj)
    call("xggm_gat_att_fwd", ptr(s), ptr(adj), ptr(att), B, N, float(alpha), stream())
    return att


def gat_att_bwd(d_att, att, s, adj, alpha, dt):
    _c(d_att, F32), _c(att, F32), _c(s, F32), _c(adj, F32)
    B, N, _ = adj.shape
    ds = torch.empty((B * N, 2), device=adj.device, dtype=dt)
    call("xggm_gat_att_bwd_" + sfx(dt), ptr(d_att), ptr(att), ptr(s), ptr(adj), ptr(ds), B, N, float(alpha), stream())
    return ds


def elu_fwd(x, out, col0):
    """out[:, col0:col0+D] = elu(x) for contiguous x [M,D] and a contiguous wider out [M,ld]."""
    _c(x), _c(out, x.dtype)
    M, D = x.shape
    ld = out.shape[1]
    assert out.shape[0] == M and col0 + D <= ld
    call("xggm_elu_fwd_" + sfx(x.dtype), ptr(x), out.data_ptr() + col0 * out.element_size(), M, D, ld, stream())


def elu_bwd(dy, y, col0, D):
    _c(dy), _c(y, dy.dtype)
    M, ld = y.shape
    assert dy.shape == y.shape and col0 + D <= ld
    dx = torch.empty((M, D), device=y.device, dtype=y.dtype)
    off = col0 * y.element_size()
    call("xggm_elu_bwd_" + sfx(y.dtype), dy.data_ptr() + off, y.data_ptr() + off, ptr(dx), M, D, ld, stream())
    return dx


def dropout(x, p, rng, sid):
    _c(x)
    out = torch.empty_like(x)
    call("xggm_dropout_" + sfx(x.dtype), ptr(x), ptr(out), x.numel(), float(p), ptr(rng), sid, stream())
    return out
