"""Per-kernel parity of the C-ABI entry points against the CPU oracle / fp64 torch on the
same seeded inputs.  fp32 storage: tolerance 1e-5..1e-4 relative (fp32 MFMA is an exact
fma chain, differences are summation order only).  bf16 storage: inputs are rounded to
bf16 first and the checker consumes the SAME rounded values, so what remains is the output
rounding: 1e-2 relative on tensors."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import rel_err  # noqa: E402


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from xggm_amd import ops as o
    return o


DEV = "cuda"
DTS = [torch.float32, torch.bfloat16]


def tol(dt, f32=2e-5, bf=1.2e-2):
    return f32 if dt == torch.float32 else bf


def rnd(shape, dt, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(shape, generator=g) * scale).to(dt)
    return x.to(DEV), x.double()


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("M,N,K", [(640, 768, 768), (1152, 2304, 768), (37, 50, 100), (32, 2274, 1536),
                                   (64, 64, 64), (1, 630, 768), (200, 96, 40)])
def test_linear_fwd_dgrad_wgrad(ops, dt, M, N, K):
    x, xr = rnd((M, K), dt, 1)
    w, wr = rnd((N, K), dt, 2, 0.05)
    b = torch.randn(N, generator=torch.Generator().manual_seed(3)).to(DEV)
    y, _ = ops.linear_fwd(x, w, b)
    ref = xr @ wr.t() + b.double().cpu()
    assert rel_err(y, ref) < tol(dt)
    # gelu + preact
    y2, pre = ops.linear_fwd(x, w, b, act=ops.ACT_GELU, want_preact=True)
    assert rel_err(pre, ref) < tol(dt)
    pr = pre.double().cpu()
    assert rel_err(y2, pr * 0.5 * (1 + torch.erf(pr / math.sqrt(2)))) < tol(dt)
    # f32 output with sigmoid
    y3, _ = ops.linear_fwd(x, w, b, act=ops.ACT_SIGMOID, out_f32=True)
    assert y3.dtype == torch.float32 and rel_err(y3, torch.sigmoid(ref)) < tol(dt, 2e-5, 2e-3)
    # dgrad
    dy, dyr = rnd((M, N), dt, 4)
    res, resr = rnd((M, K), dt, 5)
    dx = ops.linear_dgrad(dy, w, residual=res)
    assert rel_err(dx, dyr @ wr + resr) < tol(dt)
    # dgrad fused with GELU backward
    u, ur = rnd((M, K), dt, 6)
    dxg = ops.linear_dgrad(dy, w, gelu_aux=u)
    cdf = 0.5 * (1 + torch.erf(ur / math.sqrt(2)))
    pdf = torch.exp(-0.5 * ur * ur) / math.sqrt(2 * math.pi)
    assert rel_err(dxg, (dyr @ wr) * (cdf + ur * pdf)) < tol(dt)
    # ... and with the column sums (bias gradient) taken in the same epilogue
    cs = torch.full((K,), 2.0, device=DEV)
    pcs, dxc = ops.p_dgrad(dy, w, gelu_aux=u, colsum=cs)
    ops.gemm_group(dt, [pcs])
    assert rel_err(dxc, (dyr @ wr) * (cdf + ur * pdf)) < tol(dt)
    assert rel_err(cs, 2.0 + ((dyr @ wr) * (cdf + ur * pdf)).sum(0)) < tol(dt, 1e-4, 2e-3)
    # wgrad (fp32 out), overwrite then accumulate
    gw = torch.full((N, K), 7.0, device=DEV)
    ops.linear_wgrad(dy, x, gw, accumulate=False)
    assert rel_err(gw, dyr.t() @ xr) < tol(dt, 2e-5, 1e-4)
    ops.linear_wgrad(dy, x, gw, accumulate=True)
    assert rel_err(gw, 2 * (dyr.t() @ xr)) < tol(dt, 2e-5, 1e-4)
    # bias grad
    gb = torch.zeros(N, device=DEV)
    ops.colsum(dy, gb)
    assert rel_err(gb, dyr.sum(0)) < 1e-5


@pytest.mark.parametrize("M,N,K", [(32, 2274, 1536), (32, 630, 768), (70, 50, 64)])
def test_linear_backward_padded_odd_width(ops, M, N, K):
    """output widths that are not multiples of 8 (2274 answers, 630 edges): with the gradient's row stride
    padded to a multiple of 8 the backward products take the tuned bf16 path (chunks straddling the edge are
    masked / only feed unstored rows), whatever the padding holds -- here NaN -- and must equal the generic
    kernel's result on the unpadded gradient."""
    dt = torch.bfloat16
    x, xr = rnd((M, K), dt, 1)
    w, wr = rnd((N, K), dt, 2, 0.05)
    dy, dyr = rnd((M, N), dt, 4)
    pad = torch.full((M, (N + 7) // 8 * 8), float("nan"), device=DEV, dtype=dt)
    pad[:, :N] = dy
    dyp = pad[:, :N]
    assert dyp.stride(0) % 8 == 0 and N % 8 != 0
    dx = ops.linear_dgrad(dyp, w)
    assert rel_err(dx, dyr @ wr) < tol(dt)
    gw = torch.zeros((N, K), device=DEV)
    ops.linear_wgrad(dyp, x, gw, accumulate=False)
    assert bool(torch.isfinite(gw).all()) and rel_err(gw, dyr.t() @ xr) < tol(dt, 2e-5, 1e-4)
    # grouped launch of both (what the heads' backward issues)
    gw2 = torch.zeros((N, K), device=DEV)
    pd, dx2 = ops.p_dgrad(dyp, w)
    ops.gemm_group(dt, [ops.p_wgrad(dyp, x, gw2, False), pd])
    assert torch.equal(dx2, dx) and torch.equal(gw2, gw)
    # same numbers as the generic kernel on the contiguous gradient
    _l = ops._lib.lib if hasattr(ops, "_lib") else None
    from xggm_amd import _lib
    _lib.lib.xggm_gemm_set_generic(1)
    try:
        dxg = ops.linear_dgrad(dy, w)
        gwg = torch.zeros((N, K), device=DEV)
        ops.linear_wgrad(dy, x, gwg, accumulate=False)
    finally:
        _lib.lib.xggm_gemm_set_generic(0)
    assert rel_err(dx, dxg.double().cpu()) < 1e-2 and rel_err(gw, gwg.double().cpu()) < 1e-4


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("group_tile", [0, 1, 2, 3, 4, 7, 8, 9])
def test_grouped_gemm(ops, dt, group_tile):
    """forward (two modalities) + dgrad + wgrad in ONE launch == the four products done alone"""
    from xggm_amd import _lib
    _lib.lib.xggm_gemm_set_group_tile(group_tile)
    try:
        xl, xlr = rnd((640, 768), dt, 1)
        xv, xvr = rnd((1152, 768), dt, 2)
        w1, w1r = rnd((3072, 768), dt, 3, 0.05)
        w2, w2r = rnd((768, 768), dt, 4, 0.05)
        b1 = torch.randn(3072, generator=torch.Generator().manual_seed(5)).to(DEV)
        dy, dyr = rnd((1152, 768), dt, 6)
        res, resr = rnd((1152, 768), dt, 7)
        gw = torch.full((768, 768), 3.0, device=DEV)
        p1, y1, pre1 = ops.p_fwd(xl, w1, b1, act=ops.ACT_GELU, want_preact=True)
        p2, y2, _ = ops.p_fwd(xv, w2, None)
        p3, dx = ops.p_dgrad(dy, w2, residual=res)
        p4 = ops.p_wgrad(dy, xv, gw, accumulate=True)
        ops.gemm_group(dt, [p1, p2, p3, p4])
        pr = pre1.double().cpu()
        assert rel_err(pre1, xlr @ w1r.t() + b1.double().cpu()) < tol(dt)
        assert rel_err(y1, pr * 0.5 * (1 + torch.erf(pr / math.sqrt(2)))) < tol(dt)
        assert rel_err(y2, xvr @ w2r.t()) < tol(dt)
        assert rel_err(dx, dyr @ w2r + resr) < tol(dt)
        assert rel_err(gw, 3.0 + dyr.t() @ xvr) < tol(dt, 2e-5, 1e-4)
        # six problems in ONE launch (the cross-attention backward's five, the graph blocks' six read-out problems) give
        # the bits of the same problems launched alone
        def six():
            ps, outs = [], []
            for k in range(3):
                p, y, _ = ops.p_fwd(xv if k else xl, w2, None)
                pd, d = ops.p_dgrad(dy, w2)
                ps += [p, pd]
                outs += [y, d]
            return ps, outs
        ps, together = six()
        assert len(ps) == ops.GROUP_MAX
        ops.gemm_group(dt, ps)
        ps, alone = six()
        for p in ps:
            ops.gemm_group(dt, [p])
        for a_, b_ in zip(together, alone):
            assert torch.equal(a_, b_)
        # odd shapes fall back to single launches and still agree
        xo, xor_ = rnd((37, 100), dt, 8)
        wo, wor = rnd((50, 100), dt, 9, 0.1)
        p5, y5, _ = ops.p_fwd(xo, wo, None)
        p6, y6, _ = ops.p_fwd(xv, w2, None)
        ops.gemm_group(dt, [p5, p6])
        assert rel_err(y5, xor_ @ wor.t()) < tol(dt) and rel_err(y6, xvr @ w2r.t()) < tol(dt)
    finally:
        _lib.lib.xggm_gemm_set_group_tile(0)


@pytest.mark.parametrize("dt", DTS)
def test_linear_strided_rows(ops, dt):
    """pooler input lang[:, 0]: rows strided by T*H"""
    B, T, H = 5, 20, 128
    x, xr = rnd((B, T, H), dt, 1)
    w, wr = rnd((H, H), dt, 2, 0.1)
    y, _ = ops.linear_fwd(x[:, 0], w, None, act=ops.ACT_TANH)
    assert rel_err(y, torch.tanh(xr[:, 0] @ wr.t())) < tol(dt)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("N,H", [(36, 768), (64, 768), (36, 128), (5, 64)])
def test_bmm_nt(ops, dt, N, H):
    x, xr = rnd((3, N, H), dt, 1)
    S = ops.bmm_nt(x, x)
    assert rel_err(S, xr @ xr.transpose(1, 2)) < tol(dt, 2e-5, 1e-4)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("Sq,Sk", [(20, 20), (36, 36), (20, 36), (36, 20), (64, 64), (1, 7)])
@pytest.mark.parametrize("masked", [False, True])
def test_attention(ops, dt, Sq, Sk, masked):
    B, heads = 3, 2
    H = heads * 64
    # fused-QKV style buffers: q from one [B*Sq, 3H] buffer, k/v from another
    bufq, bufqr = rnd((B * Sq, 3 * H), dt, 1)
    bufk, bufkr = rnd((B * Sk, 3 * H), dt, 2)
    q, k, v = bufq[:, :H], bufk[:, H:2 * H], bufk[:, 2 * H:]
    qr, kr, vr = bufqr[:, :H], bufkr[:, H:2 * H], bufkr[:, 2 * H:]
    mask = None
    mr = 0.0
    if masked:
        lens = torch.tensor([Sk, max(1, Sk // 2), max(1, Sk - 3)])
        m01 = (torch.arange(Sk)[None, :] < lens[:, None]).float()
        mask = ((1 - m01) * -10000.0).to(DEV)
        mr = mask.double().cpu()[:, None, None, :]

    def ref(qr, kr, vr):
        qh = qr.reshape(B, Sq, heads, 64).permute(0, 2, 1, 3)
        kh = kr.reshape(B, Sk, heads, 64).permute(0, 2, 1, 3)
        vh = vr.reshape(B, Sk, heads, 64).permute(0, 2, 1, 3)
        p = torch.softmax(qh @ kh.transpose(-1, -2) / 8.0 + mr, dim=-1)
        return (p @ vh).permute(0, 2, 1, 3).reshape(B * Sq, H)

    out = ops.attn_fwd(q, k, v, mask, B, heads, Sq, Sk, 0.0, None, 0)
    assert rel_err(out, ref(qr, kr, vr)) < tol(dt)
    # backward vs autograd of the fp64 reference
    do, dor = rnd((B * Sq, H), dt, 3)
    qa, ka, va = (t.clone().requires_grad_(True) for t in (qr, kr, vr))
    (ref(qa, ka, va) * dor).sum().backward()
    dq = torch.empty((B * Sq, H), device=DEV, dtype=dt)
    dkv = torch.empty((B * Sk, 2 * H), device=DEV, dtype=dt)
    gb = torch.zeros(3 * H, device=DEV)
    ops.attn_bwd(q, k, v, mask, do, dq, dkv[:, :H], dkv[:, H:], B, heads, Sq, Sk, 0.0, None, 0, gb[:H], gb[H:2 * H],
                 gb[2 * H:])
    assert rel_err(gb[:H], qa.grad.sum(0)) < tol(dt, 1e-4, 6e-3)   # fused bias gradients (bf16: sums of bf16-rounded dS products)
    # the key bias cannot change a softmax: its gradient is zero up to rounding
    zb = 1e-3 if dt == torch.float32 else 1e-2  # bf16: dS is rounded before the product
    assert float((gb[H:2 * H].double().cpu() - ka.grad.sum(0)).abs().max()) < zb * float(qa.grad.abs().max()) * Sq
    assert rel_err(gb[2 * H:], va.grad.sum(0)) < tol(dt, 1e-4, 6e-3)
    assert rel_err(dq, qa.grad) < tol(dt, 5e-5)
    assert rel_err(dkv[:, :H], ka.grad) < tol(dt, 5e-5)
    assert rel_err(dkv[:, H:], va.grad) < tol(dt, 5e-5)


@pytest.mark.parametrize("dt", DTS)
def test_attention_dropout_consistency(ops, dt):
    """with dropout the forward equals the reference fed the exported Philox mask, and the
    backward uses the same mask."""
    B, heads, Sq, Sk, H = 2, 2, 20, 36, 128
    q, qr = rnd((B * Sq, H), dt, 1)
    k, kr = rnd((B * Sk, H), dt, 2)
    v, vr = rnd((B * Sk, H), dt, 3)
    rng = ops.make_rng(1234, DEV)
    p, sid = 0.1, 77
    out = ops.attn_fwd(q, k, v, None, B, heads, Sq, Sk, p, rng, sid)
    mask = ops.dropout_mask(B * heads * Sq * Sk, p, rng, sid, DEV).double().cpu().view(B, heads, Sq, Sk)
    keep = (mask > 0).double().mean().item()
    assert abs(keep - 0.9) < 0.03 and abs(mask.max().item() - 1 / 0.9) < 1e-6

    def ref(qr, kr, vr):
        qh = qr.reshape(B, Sq, heads, 64).permute(0, 2, 1, 3)
        kh = kr.reshape(B, Sk, heads, 64).permute(0, 2, 1, 3)
        vh = vr.reshape(B, Sk, heads, 64).permute(0, 2, 1, 3)
        pm = torch.softmax(qh @ kh.transpose(-1, -2) / 8.0, dim=-1) * mask
        return (pm @ vh).permute(0, 2, 1, 3).reshape(B * Sq, H)

    assert rel_err(out, ref(qr, kr, vr)) < tol(dt)
    do, dor = rnd((B * Sq, H), dt, 4)
    qa, ka, va = (t.clone().requires_grad_(True) for t in (qr, kr, vr))
    (ref(qa, ka, va) * dor).sum().backward()
    dq, dk, dv = (torch.empty_like(t) for t in (q, k, v))
    ops.attn_bwd(q, k, v, None, do, dq, dk, dv, B, heads, Sq, Sk, p, rng, sid)
    assert rel_err(dq, qa.grad) < tol(dt, 5e-5) and rel_err(dk, ka.grad) < tol(dt, 5e-5)
    assert rel_err(dv, va.grad) < tol(dt, 5e-5)
    # a different offset gives a different mask
    ops.rng_advance(rng)
    m2 = ops.dropout_mask(B * heads * Sq * Sk, p, rng, sid, DEV).double().cpu()
    assert (m2 != mask.view(-1)).any()


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("M,H", [(1152, 768), (7, 128), (32, 1536), (100, 256), (9, 2048)])
def test_layernorm_fwd_bwd(ops, dt, M, H):
    x, xr = rnd((M, H), dt, 1)
    res, resr = rnd((M, H), dt, 2)
    gen = torch.Generator().manual_seed(3)
    bias = torch.randn(H, generator=gen)
    gamma = 1 + 0.1 * torch.randn(H, generator=gen)
    beta = 0.1 * torch.randn(H, generator=gen)
    eps = 1e-12
    xin = x.clone()
    out, z, stats = ops.ln_fwd(xin, bias.to(DEV), res, gamma.to(DEV), beta.to(DEV), eps)
    zr = xr + bias.double() + resr
    za = zr.clone().requires_grad_(True)
    ga, ba = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(za, (H,), ga, ba, eps)
    assert rel_err(z, zr) < tol(dt, 1e-6, 5e-3)
    assert rel_err(out, yr) < tol(dt)
    dy, dyr = rnd((M, H), dt, 4)
    # the kernel normalises the stored (rounded) z; mirror that in the checker
    zs = z.double().cpu().requires_grad_(True)
    ys = torch.nn.functional.layer_norm(zs, (H,), ga, ba, eps)
    (ys * dyr).sum().backward()
    dg, db, dbias = (torch.zeros(H, device=DEV) for _ in range(3))
    d_in, d_res = ops.ln_bwd(dy, z, stats, gamma.to(DEV), dg, db, dbias, want_dres=True)
    assert rel_err(d_in, zs.grad) < tol(dt, 1e-4)
    assert rel_err(d_res, zs.grad) < tol(dt, 1e-4)
    assert rel_err(dg, ga.grad) < tol(dt, 1e-4, 2e-3)
    assert rel_err(db, ba.grad) < tol(dt, 1e-4, 2e-3)
    assert rel_err(dbias, zs.grad.sum(0)) < tol(dt, 1e-4, 5e-3)
    # accumulate into an existing residual gradient
    acc = d_res.clone()
    ops.ln_bwd(dy, z, stats, gamma.to(DEV), None, None, None, want_din=False, d_res=acc)
    assert rel_err(acc, 2 * zs.grad) < tol(dt, 1e-4, 1.5e-2)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("M,H,n", [(1152, 768, 3), (4096, 768, 2), (7, 128, 1), (100, 256, 4)])
def test_sum_of_layernorms_in_one_launch(ops, dt, M, H, n):
    """xggm_ln_sum_fwd: sum_k dropout(LayerNorm_k(x_k)), the read-out of src/module/gcn.py:70-77 / gin.py:80-87, against
    float64 without dropout and against the chain of accumulating ln_fwd launches it replaces with dropout (same masks,
    same statistics; the chain rounds the running sum to the storage type between the terms, this one once); and the
    grouped backward of the terms against separate ln_bwd calls, bit for bit."""
    gen = torch.Generator().manual_seed(11)
    xs, xrs = zip(*[rnd((M, H), dt, 40 + k) for k in range(n)])
    gam = [(1 + 0.1 * torch.randn(H, generator=gen)) for _ in range(n)]
    bet = [(0.1 * torch.randn(H, generator=gen)) for _ in range(n)]
    gd, bd = [g.to(DEV) for g in gam], [b.to(DEV) for b in bet]
    out, stats = ops.ln_sum_fwd(list(xs), gd, bd, 1e-5)
    ref = sum(torch.nn.functional.layer_norm(xr, (H,), g.double(), b.double(), 1e-5) for xr, g, b in zip(xrs, gam, bet))
    assert rel_err(out, ref) < tol(dt)
    rng = ops.make_rng(99, DEV)
    out_d, stats_d = ops.ln_sum_fwd(list(xs), gd, bd, 1e-5, p_post=0.5, rng=rng, sids=[300 + k for k in range(n)])
    chain = torch.empty_like(xs[0])
    for k in range(n):
        _, z, st = ops.ln_fwd(xs[k].clone(), None, None, gd[k], bd[k], 1e-5, p_post=0.5, rng=rng, sid_post=300 + k, out=chain,
                              accumulate=k > 0)
        assert torch.equal(z, xs[k]) and torch.equal(st, stats_d[k]) and torch.equal(st, stats[k])
    assert rel_err(out_d, chain.double().cpu()) < tol(dt, 1e-6, 8e-3)
    if dt == torch.float32 or n == 1:
        assert float((out_d.double() - chain.double()).abs().max()) < 1e-5
    # backward of the terms: one grouped launch == separate launches
    dy = rnd((M, H), dt, 60)[0]
    us = [rnd((M, H), dt, 70 + k)[0] for k in range(n)]
    sep_g = [[torch.zeros(H, device=DEV) for _ in range(3)] for _ in range(n)]
    sep = [ops.ln_bwd(dy, xs[k], stats_d[k], gd[k], *sep_g[k], p_post=0.5, rng=rng, sid_post=300 + k, gelu_aux=us[k])[0]
           for k in range(n)]
    grp_g = [[torch.zeros(H, device=DEV) for _ in range(3)] for _ in range(n)]
    grp = ops.ln_bwd_group([dict(dy=dy, z=xs[k], stats=stats_d[k], gamma=gd[k], dgamma=grp_g[k][0], dbeta=grp_g[k][1],
                                 dbias=grp_g[k][2], sid_post=300 + k, gelu_aux=us[k]) for k in range(n)], p_post=0.5, rng=rng)
    for k in range(n):
        assert torch.equal(grp[k][0], sep[k]) and grp[k][1] is None
        for a, b in zip(grp_g[k], sep_g[k]):
            assert torch.equal(a, b)
    with pytest.raises(RuntimeError, match="may not be one of the terms"):
        a = ops.LnSumArgs()
        a.inp[0], a.gamma[0], a.beta[0], a.out = xs[0].data_ptr(), gd[0].data_ptr(), bd[0].data_ptr(), xs[0].data_ptr()
        a.n, a.M, a.H, a.eps = 1, M, H, 1e-5
        ops.call("xggm_ln_sum_fwd_" + ops.sfx(dt), ops._ct.byref(a), ops.stream())


@pytest.mark.parametrize("B,N,H", [(32, 36, 768), (64, 64, 768), (5, 36, 128), (3, 17, 64), (9, 64, 256), (2, 1, 64)])
def test_gcnconv_tail_in_one_launch(ops, B, N, H):
    """xggm_agg_residual_ln_bf16: LayerNorm(res + M y) per sample -- GCNConv's tail (src/module/gcn.py:22-29) once the
    product y = x W^T has been taken -- against float64 on the same bf16 inputs, the saved rows / statistics against what
    ln_fwd saves for the same sum, and LayerNorm's own backward running on them."""
    gen = torch.Generator().manual_seed(B * 100 + N)
    M = torch.rand(B, N, N, generator=gen).to(DEV)
    y, yr = rnd((B, N, H), torch.bfloat16, 1)
    res, rr = rnd((B, N, H), torch.bfloat16, 2)
    gamma = (1 + 0.1 * torch.randn(H, generator=gen)).to(DEV)
    beta = (0.1 * torch.randn(H, generator=gen)).to(DEV)
    out, z, stats = ops.agg_residual_ln(M, y, res, gamma, beta, 1e-5)
    zr = rr + M.double().cpu() @ yr
    assert rel_err(z, zr) < 5e-3
    ref = torch.nn.functional.layer_norm(zr, (H,), gamma.double().cpu(), beta.double().cpu(), 1e-5)
    assert rel_err(out, ref) < 1.2e-2
    # the statistics are those of the ROUNDED rows (what the backward recomputes from), as ln_fwd's
    zs = z.double().cpu().view(B * N, H)
    mean = zs.mean(1)
    rstd = 1.0 / torch.sqrt(zs.var(1, unbiased=False) + 1e-5)
    assert float((stats[:, 0].double().cpu() - mean).abs().max()) < 1e-5 * (1 + float(mean.abs().max()))
    assert float((stats[:, 1].double().cpu() / rstd - 1).abs().max()) < 1e-5
    o2, z2, st2 = ops.ln_fwd(z.view(B * N, H).clone(), None, None, gamma, beta, 1e-5)
    assert torch.equal(z2, z.view(B * N, H)) and float((st2 - stats).abs().max()) < 1e-5 * (1 + float(st2.abs().max()))
    assert float((o2.float() - out.view(B * N, H).float()).abs().max()) <= 2 ** -6 * float(o2.float().abs().max())
    dy = rnd((B * N, H), torch.bfloat16, 3)[0]
    d1, _ = ops.ln_bwd(dy, z.view(B * N, H), stats, gamma, None, None, None)
    d2, _ = ops.ln_bwd(dy, z2, st2, gamma, None, None, None)
    assert rel_err(d1, d2.double().cpu()) < 1e-2
    with pytest.raises(RuntimeError, match="may not alias"):
        ops.call("xggm_agg_residual_ln_bf16", M.data_ptr(), y.data_ptr(), res.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                 y.data_ptr(), None, None, B, N, H, 1e-5, ops.stream())


@pytest.mark.parametrize("dt", DTS)
def test_grouped_row_requests_equal_separate_launches(ops, dt):
    """the language (640 rows, 20 tokens) and vision (1152 rows, 36 objects) LayerNorms / attention cores
    launched as ONE group give bit-identical activations and input gradients to separate launches; the deferred second stage of
    the LN backward adds exactly what the immediate one adds."""
    H, heads, B = 128, 2, 8
    rng = ops.make_rng(77, DEV)
    gen = torch.Generator().manual_seed(5)
    par = [[t.to(DEV) for t in (torch.randn(H, generator=gen), 1 + 0.1 * torch.randn(H, generator=gen),
                                0.1 * torch.randn(H, generator=gen))] for _ in range(2)]
    rows = (B * 20, B * 36)
    xs = [rnd((m, H), dt, 10 + i)[0] for i, m in enumerate(rows)]
    rs = [rnd((m, H), dt, 20 + i)[0] for i, m in enumerate(rows)]
    dys = [rnd((m, H), dt, 30 + i)[0] for i, m in enumerate(rows)]
    # separate
    sep = [ops.ln_fwd(x.clone(), par[i][0], rs[i], par[i][1], par[i][2], 1e-12, p_pre=0.1, rng=rng, sid_pre=5 + i)
           for i, x in enumerate(xs)]
    gsep = [[torch.zeros(H, device=DEV) for _ in range(3)] for _ in range(2)]
    bsep = [ops.ln_bwd(dys[i], sep[i][1], sep[i][2], par[i][1], *gsep[i], want_dres=True, p_pre=0.1, rng=rng, sid_pre=5 + i)
            for i in range(2)]
    # grouped + deferred
    reqs = [ops.LnFwdReq(x.clone(), par[i][0], rs[i], par[i][1], par[i][2], 1e-12, p_pre=0.1, rng=rng, sid_pre=5 + i)
            for i, x in enumerate(xs)]
    ops.launch_row_requests(reqs)
    for i in range(2):
        assert torch.equal(reqs[i].out, sep[i][0]) and torch.equal(reqs[i].z, sep[i][1])
        assert torch.equal(reqs[i].stats, sep[i][2])
    ggrp = [[torch.zeros(H, device=DEV) for _ in range(3)] for _ in range(2)]
    jobs = []
    breqs = [ops.LnBwdReq(dys[i], reqs[i].z, reqs[i].stats, par[i][1], *ggrp[i], p_pre=0.1, rng=rng, sid_pre=5 + i, defer=jobs)
             for i in range(2)]
    ops.launch_row_requests(breqs)
    assert len(jobs) == 2 and all(float(g.abs().sum()) == 0 for gs in ggrp for g in gs)  # nothing added yet
    ops.reduce_batch(jobs)
    for i in range(2):
        assert torch.equal(breqs[i].d_in, bsep[i][0]) and torch.equal(breqs[i].d_res, bsep[i][1])
        for a, b in zip(ggrp[i], gsep[i]):
            # same partial rows, summed in another (fixed) order by the batched second stage (8 row slices of
            # 16-byte loads against 4 slices of scalar loads): equal to fp32 summation-order noise
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-5 * float(b.abs().max()))
    # attention: self-attention of both streams in one launch
    Hh = heads * 64
    qkv = [rnd((m, 3 * Hh), dt, 40 + i)[0] for i, m in enumerate(rows)]
    S = (20, 36)
    outs = [ops.attn_fwd(t[:, :Hh], t[:, Hh:2 * Hh], t[:, 2 * Hh:], None, B, heads, S[i], S[i], 0.1, rng, 9 + i)
            for i, t in enumerate(qkv)]
    areqs = [ops.AttnFwdReq(t[:, :Hh], t[:, Hh:2 * Hh], t[:, 2 * Hh:], None, B, heads, S[i], S[i], 0.1, rng, 9 + i)
             for i, t in enumerate(qkv)]
    ops.launch_row_requests(areqs)
    for i in range(2):
        assert torch.equal(areqs[i].out, outs[i])
    dos = [rnd((m, Hh), dt, 50 + i)[0] for i, m in enumerate(rows)]
    d1 = [torch.empty_like(t) for t in qkv]
    d2 = [torch.empty_like(t) for t in qkv]
    for i, t in enumerate(qkv):
        ops.attn_bwd(t[:, :Hh], t[:, Hh:2 * Hh], t[:, 2 * Hh:], None, dos[i], d1[i][:, :Hh], d1[i][:, Hh:2 * Hh],
                     d1[i][:, 2 * Hh:], B, heads, S[i], S[i], 0.1, rng, 9 + i)
    ops.launch_row_requests([ops.AttnBwdReq(t[:, :Hh], t[:, Hh:2 * Hh], t[:, 2 * Hh:], None, dos[i], d2[i][:, :Hh],
                                            d2[i][:, Hh:2 * Hh], d2[i][:, 2 * Hh:], B, heads, S[i], S[i], 0.1, rng, 9 + i)
                             for i, t in enumerate(qkv)])
    for i in range(2):
        assert torch.equal(d1[i], d2[i])


@pytest.mark.parametrize("dt", DTS)
def test_layernorm_dropout_and_accumulate(ops, dt):
    """GNN read-out form: out += drop_.5(LN(x)); pre-dropout form: LN(drop_.1(x+b)+res)."""
    M, H = 108, 768
    x, xr = rnd((M, H), dt, 1)
    res, resr = rnd((M, H), dt, 2)
    gamma = torch.ones(H)
    beta = torch.zeros(H)
    rng = ops.make_rng(99, DEV)
    m_pre = ops.dropout_mask(M * H, 0.1, rng, 5, DEV).double().cpu().view(M, H)
    m_post = ops.dropout_mask(M * H, 0.5, rng, 6, DEV).double().cpu().view(M, H)
    assert abs((m_post > 0).double().mean().item() - 0.5) < 0.02
    base, baser = rnd((M, H), dt, 3)
    out = base.clone()
    _, z, stats = ops.ln_fwd(x.clone(), None, res, gamma.to(DEV), beta.to(DEV), 1e-5, p_pre=0.1, p_post=0.5,
                             rng=rng, sid_pre=5, sid_post=6, out=out, accumulate=True)
    zr = xr * m_pre + resr
    yr = torch.nn.functional.layer_norm(zr, (H,), gamma.double(), beta.double(), 1e-5) * m_post + baser
    assert rel_err(out, yr) < tol(dt)
    dy, dyr = rnd((M, H), dt, 4)
    zs = z.double().cpu().requires_grad_(True)
    xs = xr.clone().requires_grad_(True)
    (torch.nn.functional.layer_norm(zs, (H,), gamma.double(), beta.double(), 1e-5) * m_post * dyr).sum().backward()
    d_in, d_res = ops.ln_bwd(dy, z, stats, gamma.to(DEV), None, None, None, want_dres=True, p_pre=0.1, p_post=0.5,
                             rng=rng, sid_pre=5, sid_post=6)
    assert rel_err(d_res, zs.grad) < tol(dt, 1e-4)
    assert rel_err(d_in, zs.grad * m_pre) < tol(dt, 1e-4)


@pytest.mark.parametrize("dt", DTS)
def test_embeddings(ops, dt):
    from oracle import xggm_oracle as O
    B, T, H, V = 4, 20, 128, 64
    gen = torch.Generator().manual_seed(1)
    P = {"e.word_embeddings.weight": torch.randn(V, H, generator=gen).to(dt).double(),
         "e.position_embeddings.weight": torch.randn(32, H, generator=gen).to(dt).double(),
         "e.token_type_embeddings.weight": torch.randn(2, H, generator=gen).to(dt).double(),
         "e.LayerNorm.weight": (1 + 0.1 * torch.randn(H, generator=gen)).double(),
         "e.LayerNorm.bias": (0.1 * torch.randn(H, generator=gen)).double()}
    P = {k: v.requires_grad_(True) for k, v in P.items()}
    ids = torch.randint(0, V, (B, T), generator=gen)
    ids[:, -3:] = 0  # padding rows
    seg = torch.zeros_like(ids)
    seg[1] = 1
    ref = O.bert_embeddings(P, "e.", ids, seg)
    dev = {k: v.detach().to(DEV, dt if "embeddings" in k else torch.float32) for k, v in P.items()}
    out, z, stats = ops.embed_fwd(ids.to(DEV), seg.to(DEV), dev["e.word_embeddings.weight"],
                                  dev["e.position_embeddings.weight"], dev["e.token_type_embeddings.weight"],
                                  dev["e.LayerNorm.weight"], dev["e.LayerNorm.bias"], 1e-12, 0.0, None, 0)
    assert rel_err(out.view(B, T, H), ref) < tol(dt)
    # the pass's input glue riding on this launch (xggm_embed_fwd_side_*): additive mask (src/lxrt/modeling.py:919-928) and
    # fp32 -> bf16 casts by appended workgroups; the embedding's own results are untouched
    mask = (ids != 0).long().to(DEV)
    feats = torch.randn(B * 36 * 200 + 3, generator=gen).to(DEV)
    boxes = torch.rand(B * 36 * 4, generator=gen).to(DEV)
    m_out = torch.empty(mask.shape, device=DEV)
    f_out, b_out = (torch.empty(t.shape, device=DEV, dtype=torch.bfloat16) for t in (feats, boxes))
    o2, z2, s2 = ops.embed_fwd(ids.to(DEV), seg.to(DEV), dev["e.word_embeddings.weight"], dev["e.position_embeddings.weight"],
                               dev["e.token_type_embeddings.weight"], dev["e.LayerNorm.weight"], dev["e.LayerNorm.bias"], 1e-12,
                               0.0, None, 0, side=[(ops.SIDE_ADDITIVE_MASK, mask, m_out), (ops.SIDE_CAST_BF16, feats, f_out),
                                                   (ops.SIDE_CAST_BF16, boxes, b_out)])
    assert torch.equal(o2, out) and torch.equal(z2, z) and torch.equal(s2, stats)
    assert torch.equal(m_out, ops.additive_mask(mask)) and torch.equal(m_out, (1.0 - mask.float()) * -10000.0)
    assert torch.equal(f_out, feats.to(torch.bfloat16)) and torch.equal(b_out, boxes.to(torch.bfloat16))
    dy, dyr = rnd((B * T, H), dt, 2)
    (ref * dyr.view(B, T, H)).sum().backward()
    g = {k: torch.zeros(v.shape, device=DEV) for k, v in P.items()}
    ops.embed_bwd(ids.to(DEV), seg.to(DEV), dy, z, stats, dev["e.LayerNorm.weight"], g["e.word_embeddings.weight"],
                  g["e.position_embeddings.weight"], g["e.token_type_embeddings.weight"], g["e.LayerNorm.weight"],
                  g["e.LayerNorm.bias"], 0.0, None, 0)
    for k in P:
        assert rel_err(g[k], P[k].grad) < tol(dt, 1e-4, 2e-2), k
    assert float(g["e.word_embeddings.weight"][0].abs().max()) == 0.0  # padding_idx row


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("H,F", [(768, 2048), (128, 64)])
def test_visual_embedding(ops, dt, H, F):
    from oracle import xggm_oracle as O
    M = 72
    gen = torch.Generator().manual_seed(1)
    P = {"v.visn_fc.weight": (0.02 * torch.randn(H, F, generator=gen)).to(dt).double(),
         "v.visn_fc.bias": 0.1 * torch.randn(H, generator=gen).double(),
         "v.visn_layer_norm.weight": (1 + 0.1 * torch.randn(H, generator=gen)).double(),
         "v.visn_layer_norm.bias": 0.1 * torch.randn(H, generator=gen).double(),
         "v.box_fc.weight": torch.randn(H, 4, generator=gen).double(),
         "v.box_fc.bias": 0.1 * torch.randn(H, generator=gen).double(),
         "v.box_layer_norm.weight": (1 + 0.1 * torch.randn(H, generator=gen)).double(),
         "v.box_layer_norm.bias": 0.1 * torch.randn(H, generator=gen).double()}
    P = {k: v.requires_grad_(True) for k, v in P.items()}
    feats, fr = rnd((M, F), dt, 2)
    fr = (fr.abs()).clone()
    feats = feats.abs()
    boxes, br = rnd((M, 4), dt, 3)
    ref = O.visual_feat_encoder(P, "v.", fr, br)
    d32 = {k: v.detach().float().to(DEV) for k, v in P.items()}
    u, _ = ops.linear_fwd(feats, d32["v.visn_fc.weight"].to(dt), None)
    out, z1, z2, stats = ops.visn_embed_fwd(u, d32["v.visn_fc.bias"], boxes, d32["v.box_fc.weight"],
                                            d32["v.box_fc.bias"], d32["v.visn_layer_norm.weight"],
                                            d32["v.visn_layer_norm.bias"], d32["v.box_layer_norm.weight"],
                                            d32["v.box_layer_norm.bias"], 1e-12, 0.0, None, 0)
    assert rel_err(out, ref) < tol(dt, 5e-5)
    dy, dyr = rnd((M, H), dt, 4)
    (ref * dyr).sum().backward()
    grads = {k: torch.zeros(s, device=DEV) for k, s in
             dict(dbf=H, dg1=H, db1=H, dWb=(H, 4), dbb=H, dg2=H, db2=H).items()}
    du = ops.visn_embed_bwd(dy, z1, z2, stats, boxes, d32["v.visn_layer_norm.weight"],
                            d32["v.box_layer_norm.weight"], grads, 0.0, None, 0)
    names = dict(dbf="v.visn_fc.bias", dg1="v.visn_layer_norm.weight", db1="v.visn_layer_norm.bias",
                 dWb="v.box_fc.weight", dbb="v.box_fc.bias", dg2="v.box_layer_norm.weight",
                 db2="v.box_layer_norm.bias")
    for k, n in names.items():
        assert rel_err(grads[k], P[n].grad) < tol(dt, 2e-4, 2e-2), n
    gw = torch.zeros(H, F, device=DEV)
    ops.linear_wgrad(du, feats, gw, False)
    assert rel_err(gw, P["v.visn_fc.weight"].grad) < tol(dt, 2e-4, 2e-2)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("N", [36, 64, 5])
def test_aggregate(ops, dt, N):
    B, H = 3, 768
    x, xr = rnd((B, N, H), dt, 1)
    adj = torch.randn(B, N, N, generator=torch.Generator().manual_seed(2))
    a = adj.double()
    assert rel_err(ops.aggregate(adj.to(DEV), x), a @ xr) < tol(dt)
    assert rel_err(ops.aggregate(adj.to(DEV), x, mode=ops.AGG_TRANSPOSE), a.transpose(1, 2) @ xr) < tol(dt)
    assert rel_err(ops.aggregate(adj.to(DEV), x, mode=ops.AGG_SYMMETRIZE), (a + a.transpose(1, 2)) @ xr) < tol(dt)
    eps = torch.tensor([0.3], device=DEV)
    out = x.clone()
    ops.aggregate(adj.to(DEV), x, scale_ptr=eps, self_w=1.0, out=out)
    assert rel_err(out, xr + xr + 1.3 * (a @ xr)) < tol(dt)
    dh, dhr = rnd((B, N, H), dt, 3)
    acc = torch.zeros(1, device=DEV)
    ops.agg_dot(adj.to(DEV), x, dh, acc)
    assert abs(float(acc) - float(((a @ xr) * dhr).sum())) < 1e-3 * float(((a @ xr) * dhr).abs().sum())
    # a ragged last column block, a batch that fills whole XCD groups (the bf16 kernel deals samples to XCDs in eights),
    # accumulation into an existing output
    for B2, H2 in ((16, 96), (9, 200)):
        x2, x2r = rnd((B2, N, H2), dt, 5)
        adj2 = torch.randn(B2, N, N, generator=torch.Generator().manual_seed(6))
        got = ops.aggregate(adj2.to(DEV), x2)
        assert rel_err(got, adj2.double() @ x2r) < tol(dt)
        base, baser = rnd((B2, N, H2), dt, 7)
        ops.aggregate(adj2.to(DEV), x2, mode=ops.AGG_TRANSPOSE, out=base)
        assert rel_err(base, baser + adj2.double().transpose(1, 2) @ x2r) < tol(dt)
    if dt == torch.bfloat16:
        # the adjacency is NOT rounded to bf16 (hi + lo split): a matrix whose entries need > 8 mantissa bits
        adj3 = (1.0 + torch.arange(N * N, dtype=torch.float32).reshape(1, N, N) / 4096.0).repeat(2, 1, 1)
        x3, x3r = rnd((2, N, H), dt, 8)
        got = ops.aggregate(adj3.to(DEV), x3).double().cpu()
        want = adj3.double() @ x3r
        rounded = adj3.to(torch.bfloat16).double() @ x3r
        assert rel_err(got, want) < 4e-3 and rel_err(got, want) < 0.5 * rel_err(rounded, want) + 3e-3


@pytest.mark.parametrize("N", [36, 64, 7])
def test_adj_regen(ops, N):
    from oracle import xggm_oracle as O
    B, H = 4, 96
    x = torch.randn(B, N, H, generator=torch.Generator().manual_seed(1)).double()
    S = (x @ x.transpose(1, 2))
    S[0, 2, 3] = S[0, :, 3].max() + 1.0  # an off-diagonal column maximum
    S[1, 4, 5] = S[1, 5, 5]              # an exact tie: torch picks the first index (row 4)
    Sd = S.float().to(DEV)
    adj, colmax, argmax = ops.adj_regen_fwd(Sd)
    Sr = S.float().double().requires_grad_(True)
    s = Sr / Sr.max(dim=1)[0].unsqueeze(-1)
    s = torch.sigmoid(s)
    ref = s.triu(1) + s.tril(-1)
    assert rel_err(adj, ref) < 1e-6
    assert torch.equal(argmax.cpu().long(), Sr.max(dim=1)[1])  # bit-exact indices
    assert int(argmax[0, 3]) == 2 and int(argmax[1, 5]) == 4
    g = torch.randn(B, N, N, generator=torch.Generator().manual_seed(2))
    (ref * g.double()).sum().backward()
    dS = ops.adj_regen_bwd(g.to(DEV), Sd, adj, colmax, argmax)
    assert rel_err(dS, Sr.grad) < 1e-5
    # the whole regeneration incl. S = x x^T against the oracle
    xs = x.float()
    Sx = ops.bmm_nt(xs.to(DEV), xs.to(DEV))
    a2, _, _ = ops.adj_regen_fwd(Sx)
    assert rel_err(a2, O.regen_adj(xs.double())) < 1e-5


@pytest.mark.parametrize("N", [36, 64, 4])
def test_adj_init_index_map_bit_exact(ops, N):
    from oracle import xggm_oracle as O
    B = 3
    NE = N * (N - 1) // 2
    ii, jj = O.triu_index_table(N)
    for k in (0, 1, NE // 2, NE - 1):
        assert ops.triu_index(k, N) == (int(ii[k]), int(jj[k]))
    e = torch.arange(B * NE, dtype=torch.float32).view(B, NE) + 1.0
    randn = torch.randn(B, N, N, generator=torch.Generator().manual_seed(1))
    adj, g = ops.adj_init_fwd(e.to(DEV), N, 0.7, randn=randn.to(DEV))
    a0 = O.adj_init(e, N)
    an, ag = O.add_edge_noise_v2(a0, randn, 0.7)
    # noiseless scatter is exact (integers): compare positions bit for bit
    adj0, _ = ops.adj_init_fwd(e.to(DEV), N, 0.7, randn=torch.zeros(B, N, N, device=DEV))
    assert torch.equal(adj0.cpu(), a0)
    assert rel_err(adj, an) < 1e-6 and rel_err(g, ag) < 1e-6
    d = torch.randn(B, N, N, generator=torch.Generator().manual_seed(2))
    de = ops.adj_init_bwd(d.to(DEV))
    assert rel_err(de, d[:, ii, jj] + d[:, jj, ii]) < 1e-6
    # Philox noise: symmetric, zero diagonal, unit variance
    rng = ops.make_rng(5, DEV)
    an2, g2 = ops.adj_init_fwd(None, N, 1.0, rng=rng, sid=9, B=64)
    assert torch.equal(an2, an2.transpose(1, 2)) and float(an2.diagonal(dim1=1, dim2=2).abs().max()) == 0
    if N >= 36:
        off = an2[:, ii, jj]
        assert abs(float(off.mean())) < 0.05 and abs(float(off.std()) - 1.0) < 0.05
        assert torch.equal(g2, -an2)


@pytest.mark.parametrize("dt", DTS)
def test_feature_noise_pool_bcast(ops, dt):
    B, N, H = 3, 36, 128
    x, xr = rnd((B, N, H), dt, 1)
    z = torch.randn(B, N, H, generator=torch.Generator().manual_seed(2))
    out, g = ops.feature_noise(x, 0.7, randn=z.to(DEV))
    assert rel_err(out, xr + 0.7 * z.double()) < tol(dt) and rel_err(g, -z.double() / 0.7) < 1e-6
    rng = ops.make_rng(3, DEV)
    o2, g2 = ops.feature_noise(x, 1.0, rng=rng, sid=4)
    assert abs(float(g2.std()) - 1.0) < 0.05 and abs(float(g2.mean())) < 0.05
    p, pr = rnd((B, H), dt, 3)
    cat = ops.pool_concat_fwd(p, x)
    ref = torch.cat([pr, torch.tanh(xr.mean(1))], dim=-1)
    assert rel_err(cat, ref) < tol(dt)
    d, dr = rnd((B, 2 * H), dt, 4)
    dx, dn = ops.pool_concat_bwd(d, cat, N)
    t = cat.double().cpu()[:, H:]
    assert rel_err(dx, dr[:, :H]) < tol(dt)
    assert rel_err(dn, (dr[:, H:] * (1 - t * t) / N).unsqueeze(1).expand(B, N, H)) < tol(dt)
    bc = ops.bcast_rows(p, N)
    assert torch.equal(bc, p.unsqueeze(1).expand(B, N, H))
    assert rel_err(ops.sum_rows(x), xr.sum(1)) < tol(dt)


@pytest.mark.parametrize("dt", DTS)
def test_losses(ops, dt):
    from oracle import xggm_oracle as O
    B, N, W = 4, 36, 768
    s, sr = rnd((B, N, W), dt, 1)
    g = torch.randn(B, N, W, generator=torch.Generator().manual_seed(2))
    sigma = 0.7
    coef = 0.5 * sigma ** 2 / (B * N * W)
    loss = ops.dsm_fwd(s, g.to(DEV), coef)
    sa = sr.clone().requires_grad_(True)
    ref = O.loss_func(sa, g.double(), sigma)
    ref.backward()
    assert abs(float(loss) - float(ref)) < 1e-5 * abs(float(ref))
    gout = torch.tensor(2.5, device=DEV)
    ds = ops.dsm_bwd(s, g.to(DEV), gout, coef)
    assert rel_err(ds, 2.5 * sa.grad) < tol(dt)
    # symmetric KL over 768-wide and 36-wide rows
    for W2, dt2 in ((768, dt), (36, torch.float32)):
        x, xr = rnd((B, N, W2), dt2, 3)
        y, yr = rnd((B, N, W2), dt2, 4)
        coef = 1.0 / (B * N * W2)
        l = ops.symkl_fwd(x, y, coef)
        xa, ya = xr.clone().requires_grad_(True), yr.clone().requires_grad_(True)
        r = O.compute_kl_loss(xa, ya)
        r.backward()
        assert abs(float(l) - float(r)) < 2e-5 * abs(float(r))
        dx, dy = ops.symkl_bwd(x, y, gout, coef, True, True)
        assert rel_err(dx, 2.5 * xa.grad) < tol(dt2, 1e-4) and rel_err(dy, 2.5 * ya.grad) < tol(dt2, 1e-4)
    A = 2274
    lg = (3 * torch.randn(B, A, generator=torch.Generator().manual_seed(5)))
    tg = (torch.rand(B, A, generator=torch.Generator().manual_seed(6)) > 0.99).float()
    l = ops.bce_fwd(lg.to(DEV), tg.to(DEV), 1.0 / B)
    la = lg.double().requires_grad_(True)
    r = O.bce_with_logits_mean(la, tg.double()) * A
    r.backward()
    assert abs(float(l) - float(r)) < 1e-5 * abs(float(r))
    dl = ops.bce_bwd(lg.to(DEV), tg.to(DEV), gout, 1.0 / B, dt)
    assert rel_err(dl, 2.5 * la.grad) < tol(dt)


def test_bertadam_against_golden(ops):
    """the reference BertAdam trajectory stored in tests/golden/pieces.npz"""
    from helpers import load_golden
    from oracle import xggm_oracle as O
    g = load_golden("pieces")
    for key, lr in (("1", 4e-3), ("2", 1e-3)):
        p = torch.from_numpy(g["adam_p%s_0" % key]).clone().to(DEV).view(-1)
        n = p.numel()
        # pad to the arena's 16-byte granularity
        P = torch.zeros(((n + 3) // 4) * 4 + 4, device=DEV)
        P[:n] = p
        M, V = torch.zeros_like(P), torch.zeros_like(P)
        shadow = torch.zeros(P.numel(), device=DEV, dtype=torch.bfloat16)
        step = torch.zeros(1, dtype=torch.int64, device=DEV)
        lr_scale = torch.zeros(1, device=DEV)
        for s in range(6):
            G = torch.zeros_like(P)
            G[:n] = torch.from_numpy(g["adam_g%s" % key][s]).view(-1)
            ops.sched_step(step, lr_scale, 10, 0.1)
            assert abs(float(lr_scale) - O.warmup_linear(s / 10, 0.1)) < 1e-6
            ops.bertadam(P[:n], G[:n], M[:n], V[:n], shadow[:n], None, 5.0, lr, lr_scale, 0.9, 0.999, 1e-6, 0.01)
            assert rel_err(P[:n], torch.from_numpy(g["adam_p%s" % key][s]).view(-1)) < 1e-6
            assert rel_err(shadow[:n].float(), P[:n]) < 4e-3
        assert int(step) == 6


def test_sqnorm_and_clip(ops):
    n = 1_000_003
    g = torch.randn(n + 1, generator=torch.Generator().manual_seed(1)).to(DEV)[:n]
    gp = torch.zeros(((n + 3) // 4) * 4, device=DEV)
    gp[:n] = g
    out = torch.zeros(1, device=DEV)
    ops.sqnorm(gp[:n], out)
    ref = float((g.double() ** 2).sum())
    assert abs(float(out) - ref) < 1e-5 * ref
    # fixed summation order: repeated calls (and hence data-parallel replicas) agree bit for bit
    big = torch.randn(20_000_001, generator=torch.Generator().manual_seed(2)).to(DEV)[:20_000_000]
    vals = []
    for _ in range(6):
        o = torch.zeros(1, device=DEV)
        ops.sqnorm(big, o)
        vals.append(float(o))
    assert len(set(vals)) == 1 and abs(vals[0] - float((big.double() ** 2).sum())) < 1e-5 * vals[0]
    # clip: grads 100x too large are scaled to norm 5 before the update
    p = torch.ones(8, device=DEV)
    G = torch.full((8,), 100.0, device=DEV)
    sq = torch.zeros(1, device=DEV)
    ops.sqnorm(G, sq)
    m, v = torch.zeros(8, device=DEV), torch.zeros(8, device=DEV)
    ops.bertadam(p, G, m, v, None, sq, 5.0, 0.1, None, 0.9, 0.999, 1e-6, 0.0)
    gc = 100.0 * 5.0 / (math.sqrt(8 * 100.0 ** 2) + 1e-6)
    mm, vv = 0.1 * gc, 0.001 * gc * gc
    assert abs(float(p[0]) - (1 - 0.1 * mm / (math.sqrt(vv) + 1e-6))) < 1e-5


def test_philox_normal_moments(ops):
    rng = ops.make_rng(7, DEV)
    z = ops.normal(1 << 20, rng, 3, DEV).double().cpu()
    assert abs(float(z.mean())) < 5e-3 and abs(float(z.std()) - 1) < 5e-3
    assert abs(float((z ** 4).mean()) - 3.0) < 0.05
    z2 = ops.normal(1 << 20, rng, 4, DEV).double().cpu()
    assert abs(float((z * z2).mean())) < 5e-3  # streams are independent


def test_bad_arguments_fail_loudly(ops):
    x = torch.zeros(4, 6, device=DEV)
    with pytest.raises(RuntimeError):
        ops.ln_fwd(x, None, None, torch.ones(6, device=DEV), torch.zeros(6, device=DEV), 1e-5)  # H % 4
    with pytest.raises(RuntimeError):
        ops.attn_fwd(torch.zeros(130, 64, device=DEV), torch.zeros(130, 64, device=DEV),
                     torch.zeros(130, 64, device=DEV), None, 2, 1, 65, 65, 0.0, None, 0)  # S > 64
    with pytest.raises(RuntimeError):
        ops.linear_fwd(torch.zeros(4, 8), torch.zeros(3, 8), None)  # CPU tensors: no fallback


# ----------------------------------------------------------------------------- fp8 forward (BASELINE config C5)
@pytest.mark.parametrize("dt", DTS)
def test_quantize_fp8_is_torch_e4m3fn_rounding(ops, dt):
    """xggm_quantize_fp8e4m3: bit-exact against torch's float8_e4m3fn cast (OCP e4m3fn, round-to-nearest-even,
    subnormals, ties, signed zeros) of clamp(x * scale, +-448) computed in fp32; amax is max |x| exactly."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(4096 * 8, generator=g) * torch.logspace(-6, 3, 4096 * 8)  # subnormal range up to saturation
    x[:64] = torch.tensor([0.0, -0.0, 448.0, -448.0, 1e9, -1e9, 2 ** -9, 2 ** -10, 1.0625, 1.1875, 0.0009765625 * 1.5,
                           17.0, 18.0, 19.0, 20.0, 21.0] * 4)
    x = x.to(dt)
    for qs in (None, 0.37, 64.0):
        s = None if qs is None else torch.tensor([qs], device=DEV)
        amax = torch.zeros(1, device=DEV)
        y = ops.quantize_fp8(x.to(DEV), s, amax)
        ref = (x.float() * (torch.tensor(qs) if qs else 1.0)).clamp(-448, 448).to(torch.float8_e4m3fn)
        assert torch.equal(y.cpu().view(torch.uint8), ref.view(torch.uint8)), qs
        assert float(amax) == float(x.float().abs().max())
    with pytest.raises(RuntimeError, match="multiple of 8"):
        ops.quantize_fp8(torch.zeros(12, device=DEV))


@pytest.mark.parametrize("shape", [(640, 768, 768), (1152, 3072, 768), (1152, 768, 3072), (36, 2304, 768), (70, 200, 48)])
@pytest.mark.parametrize("tile", [0, 1, 3, 5])
def test_gemm_fp8_matches_dequantised_product(ops, shape, tile):
    """xggm_gemm_fp8e4m3 against the fp32 product of the SAME e4m3 operands (what the matrix cores must compute:
    exact products, fp32 accumulation) with scales, bias, GeLU and residual; every tile variant; ragged M / N.
    Tolerance: bf16 rounding of the result plus fp32 accumulation-order noise."""
    from xggm_amd import _lib
    BF16 = torch.bfloat16
    M, N, K = shape
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) * 0.05
    b = torch.randn(N, generator=g).to(DEV)
    res = torch.randn(M, N, generator=g).to(BF16).to(DEV)
    ax, aw = x.abs().max().reshape(1).to(DEV), w.abs().max().reshape(1).to(DEV)
    qx, sx = ops.fp8_scale_for(ax)
    qw, sw = ops.fp8_scale_for(aw)
    x8, w8 = ops.quantize_fp8(x.to(BF16).to(DEV), qx), ops.quantize_fp8(w.to(DEV), qw)
    _lib.lib.xggm_gemm_set_tile(tile)
    try:
        y, pre = ops.linear_fwd_fp8(x8, w8, sx, sw, bias=b, act=ops.ACT_GELU, want_preact=True, residual=res)
        y32, _ = ops.linear_fwd_fp8(x8, w8, sx, sw, out_f32=True)
    finally:
        _lib.lib.xggm_gemm_set_tile(0)
    prod = (x8.float().double() @ w8.float().double().t()) * float(sx) * float(sw)
    assert rel_err(y32, prod) < 1e-4  # the fp8 MFMA sums each 32-term group before the fp32 accumulate
    u = (prod + b.double()).float()
    assert rel_err(pre, u) < 4e-3
    ub = pre.float()  # the activation sees the stored pre-activation
    want = torch.nn.functional.gelu(ub) + res.float()
    assert rel_err(y, want) < 4e-3
    # against the unquantised product: the e4m3 error itself (3 mantissa bits per operand, averaged over K)
    full = x.to(BF16).double().to(DEV) @ w.double().to(DEV).t()
    assert rel_err(y32, full) < 6e-2
    for bad in (dict(K=K - 8),):
        with pytest.raises(RuntimeError, match="multiples of 16"):
            ops.call("xggm_gemm_fp8e4m3", ops.ptr(x8), ops.ptr(w8), ops.ptr(y), M, N, bad["K"], K, K, N, None, None, None,
                     None, None, 0, 0, ops.stream())


def test_layernorm_forward_sums_split_k_partials(ops):
    """the FFN output product as S = 3 split-K slices (fp32 partial sums through the batch dimension of the grouped
    GEMM) handed to the residual LayerNorm, which adds them on the way in: against fp64 torch of the same bf16
    operands, and against the unsplit path (which rounds the product to bf16 before the LayerNorm)."""
    BF = torch.bfloat16
    M, K, H, S = 1152, 3072, 768, 3
    x, x64 = rnd((M, K), BF, 1)
    w, w64 = rnd((H, K), BF, 2, 0.03)
    res, res64 = rnd((M, H), BF, 3)
    b = torch.randn(H, generator=torch.Generator().manual_seed(4)).to(DEV)
    gamma = (1 + 0.1 * torch.randn(H, generator=torch.Generator().manual_seed(5))).to(DEV)
    beta = (0.1 * torch.randn(H, generator=torch.Generator().manual_seed(6))).to(DEV)
    p, part = ops.p_fwd_splitk(x, w, S)
    ops.gemm_group(BF, [p])
    assert tuple(part.shape) == (S, M, H)
    prod = x64.to(DEV) @ w64.to(DEV).t()
    assert rel_err(part.sum(0), prod) < 1e-5
    ln = ops.LnFwdReq(part, b, res, gamma, beta, 1e-12, dtype=BF)
    ops.launch_row_requests([ln])
    z = prod + b.double() + res64.to(DEV)
    want = torch.nn.functional.layer_norm(z, (H,), gamma.double(), beta.double(), 1e-12)
    assert rel_err(ln.z, z) < 4e-3 and rel_err(ln.out, want) < 6e-3
    # the unsplit path
    p1, h, _ = ops.p_fwd(x, w, None)
    ops.gemm_group(BF, [p1])
    ln1 = ops.LnFwdReq(h, b, res, gamma, beta, 1e-12)
    ops.launch_row_requests([ln1])
    assert rel_err(ln.out, ln1.out.double()) < 1e-2
    with pytest.raises(RuntimeError, match="multiple of 64"):
        ops.p_fwd_splitk(x, w, 5)


@pytest.mark.parametrize("group_tile", [0, 1, 2, 3, 4, 7, 8, 9])
def test_wgrad_norm_slots(ops, group_tile):
    """xggm_gemm_problem.sqsum: every 64 x 64 block of the stored fp32 weight gradient leaves its sum of squares in
    its slot -- for every tile size of the grouped kernels (a 128-wide tile writes 2 or 4 slots), with ``accumulate``
    (the slot holds the sum of the ACCUMULATED values), for ragged edges (2274 rows: the last block is partial), in a
    group with a problem that has no slots; the fixed-order total is the squared norm of the gradients."""
    from xggm_amd import _lib
    BF = torch.bfloat16
    shapes_ = [(1152, 768, 3072), (640, 2274, 1536), (1152, 768, 768)]  # (M tokens, N out, K in)
    _lib.lib.xggm_gemm_set_group_tile(group_tile)
    try:
        for accumulate in (False, True):
            probs, checks = [], []
            for i, (M, N, K) in enumerate(shapes_):
                Np = (N + 7) // 8 * 8  # odd widths: rows padded to a multiple of 8 elements, as the heads do
                dy = rnd((M, Np), BF, 10 + i, 0.5)[0][:, :N]
                x, _ = rnd((M, K), BF, 20 + i)
                gw = torch.randn(N, K, device=DEV) if accumulate else torch.empty(N, K, device=DEV)
                prev = gw.clone()
                nr, nc = (N + 63) // 64, (K + 63) // 64
                slots = torch.full((nr * nc,), float("nan"), device=DEV) if i != 2 else None
                probs.append(ops.p_wgrad(dy, x, gw, accumulate, slots))
                checks.append((dy, x, gw, prev, slots, nr, nc, N, K))
            ops.gemm_group(BF, probs)
            for dy, x, gw, prev, slots, nr, nc, N, K in checks:
                want = dy.double().t() @ x.double() + (prev.double() if accumulate else 0)
                assert rel_err(gw, want) < 1e-5
                if slots is None:
                    continue
                pad = torch.zeros(nr * 64, nc * 64, device=DEV, dtype=torch.float64)
                pad[:N, :K] = gw.double() ** 2
                blocks = pad.view(nr, 64, nc, 64).sum(dim=(1, 3)).reshape(-1)
                assert torch.isfinite(slots).all()
                assert float((slots.double() - blocks).abs().max()) < 1e-5 * float(blocks.max())
                assert abs(float(slots.double().sum()) - float((gw.double() ** 2).sum())) < 1e-6 * float(blocks.sum())
    finally:
        _lib.lib.xggm_gemm_set_group_tile(0)
    # fp32 storage runs on the generic kernel, which has no such epilogue: refused, not silently skipped
    dy, _ = rnd((64, 64), torch.float32, 1)
    p = ops.p_wgrad(dy, dy, torch.empty(64, 64, device=DEV), False, torch.zeros(1, device=DEV))
    with pytest.raises(RuntimeError, match="sqsum"):
        ops.gemm_group(torch.float32, [p])


def test_sched_step_multi_matches_reference_schedule(ops):
    """xggm_sched_step_multi: warmup_linear(step / t_total, warmup) (src/lxrt/optimization.py:42-48) and step += 1 for
    several counters of one table in one launch, untouched counters stay; against the host schedule."""
    from oracle import xggm_oracle as O
    steps = torch.tensor([0, 3, 7, 19, 40, 5], dtype=torch.int64, device=DEV)
    scale = torch.full((6,), -1.0, device=DEV)
    entries = [(0, 20, 0.1), (2, 20, 0.1), (3, 20, 0.1), (4, 20, 0.1), (5, -1, 0.1)]
    ops.sched_step_multi(steps, scale, entries)
    assert steps.tolist() == [1, 3, 8, 20, 41, 6]
    want = [O.warmup_linear(0 / 20, 0.1), -1.0, O.warmup_linear(7 / 20, 0.1), O.warmup_linear(19 / 20, 0.1),
            O.warmup_linear(40 / 20, 0.1), 1.0]
    assert np.allclose(scale.cpu().numpy(), want, rtol=1e-6, atol=1e-7)
    with pytest.raises(RuntimeError, match="listed twice"):
        ops.sched_step_multi(steps, scale, [(1, 20, 0.1), (1, 20, 0.1)])


def test_grid_sum_workspace_is_per_stream(ops):
    """ops.sum_ws: the ticket workspace of the loss kernels' grid-wide sum belongs to ONE stream (launches of one stream run
    in order; two overlapping launches on one counter would corrupt it for good) -- another stream gets its own, and both
    give the right sums when they run side by side."""
    home = ops.sum_ws(DEV)
    side = torch.cuda.Stream()
    gen = torch.Generator().manual_seed(2)
    s = torch.randn(1152 * 768, generator=gen).to(torch.bfloat16).to(DEV)
    g = torch.randn(1152 * 768, generator=gen).to(DEV)
    want = float(((s.double() - g.double()) ** 2).sum() * 0.5)
    torch.cuda.synchronize()
    outs = []
    with torch.cuda.stream(side):
        assert ops.sum_ws(DEV).data_ptr() != home.data_ptr()
        for _ in range(50):
            outs.append(ops.dsm_fwd(s, g, 0.5))
    for _ in range(50):
        outs.append(ops.dsm_fwd(s, g, 0.5))
    torch.cuda.synchronize()
    assert ops.sum_ws(DEV).data_ptr() == home.data_ptr()
    for o in outs:
        assert abs(float(o) - want) < 1e-5 * want
    assert float(home[0]) == 0.0


def test_clip_norm_pair_with_the_pass_tail(ops):
    """xggm_clip_norm_f32: the norm nn.utils.clip_grad_norm_ returns (src/vqa/vqacpv2.py:175) over ranges of the gradient
    buffer (squared) and of the slot table (summed as they are) in one pair of launches, against float64; the finishing
    launch takes BertAdam.step's schedule step (src/lxrt/optimization.py:42-48, 170-180) and the RNG advance along;
    repeated calls give the same bits (fixed summation order)."""
    from oracle import xggm_oracle as O
    gen = torch.Generator(device="cpu").manual_seed(5)
    g = torch.randn(3_000_011, generator=gen).to(DEV)
    slots = torch.rand(5000, generator=gen).to(DEV)
    spans = [(0, 1_000_000), (1_000_004, 1_000_004 + 37), (1_200_000, 3_000_011), (8, 8)]
    slot_spans = [(0, 1024), (2048, 5000)]
    want = sum(float((g[a:b].double() ** 2).sum()) for a, b in spans) + sum(float(slots[a:b].double().sum()) for a, b in slot_spans)
    out, norm = torch.full((1,), -3.0, device=DEV), torch.zeros(1, device=DEV)
    steps = torch.tensor([0, 3, 7], dtype=torch.int64, device=DEV)
    scale = torch.full((3,), -1.0, device=DEV)
    rng = torch.tensor([11, 40], dtype=torch.int64, device=DEV)
    ops.clip_norm(g, spans, slots, slot_spans, out, norm, mul=0.25, sched=(steps, scale, [(0, 20, 0.1), (2, 20, 0.1)]), rng=(rng, 3))
    assert abs(float(out) - 0.25 * want) < 2e-6 * want
    assert abs(float(norm) - (0.25 * want) ** 0.5) < 2e-6 * want ** 0.5
    assert steps.tolist() == [1, 3, 8] and rng.tolist() == [11, 43]
    assert np.allclose(scale.cpu().numpy(), [O.warmup_linear(0 / 20, 0.1), -1.0, O.warmup_linear(7 / 20, 0.1)], rtol=1e-6, atol=1e-7)
    first = float(out)
    for _ in range(3):  # no tail: only the norm; same bits every time
        o2 = torch.zeros(1, device=DEV)
        ops.clip_norm(g, spans, slots, slot_spans, o2, None, mul=0.25)
        assert float(o2) == first
    assert steps.tolist() == [1, 3, 8] and rng.tolist() == [11, 43]
    # ranges only / slots only / nothing at all (the finish then writes 0)
    o3 = torch.ones(1, device=DEV)
    ops.clip_norm(g, spans[:1], None, [], o3)
    assert abs(float(o3) - float((g[:1_000_000].double() ** 2).sum())) < 2e-6 * want
    ops.clip_norm(g, [], slots, slot_spans[:1], o3)
    assert abs(float(o3) - float(slots[:1024].double().sum())) < 1e-3
    ops.clip_norm(g, [], None, [], o3)
    assert float(o3) == 0.0
    with pytest.raises(RuntimeError, match="listed twice"):
        ops.clip_norm(g, spans[:1], None, [], o3, sched=(steps, scale, [(1, 20, 0.1), (1, 20, 0.1)]))


def test_clip_norm_over_the_bf16_wire(ops):
    """xggm_clip_norm_bf16: the data-parallel norm pass (sum of squares of the bf16 wire arena's active ranges, times
    1 / world^2) in one pair of launches, against float64; ``accumulate`` extends a sum another call (or an all-reduce)
    left in *out -- the sharded update's two halves give what one call over all ranges gives, to rounding."""
    gen = torch.Generator(device="cpu").manual_seed(6)
    g = (torch.randn(2_000_003, generator=gen) * 0.1).to(torch.bfloat16).to(DEV)
    spans = [(0, 700_000), (700_008, 700_008 + 13), (1_000_000, 2_000_003)]
    want = sum(float((g[a:b].double() ** 2).sum()) for a, b in spans)
    out, norm = torch.full((1,), 9.0, device=DEV), torch.zeros(1, device=DEV)
    ops.clip_norm_bf16(g, spans, out, norm, mul=0.25)
    assert abs(float(out) - 0.25 * want) < 2e-6 * want and abs(float(norm) - (0.25 * want) ** 0.5) < 2e-6 * want ** 0.5
    o2 = torch.zeros(1, device=DEV)
    ops.clip_norm_bf16(g, spans[:2], o2)                                    # first half seeds
    steps = torch.tensor([4], dtype=torch.int64, device=DEV)
    scale = torch.zeros(1, device=DEV)
    ops.clip_norm_bf16(g, spans[2:], o2, norm, accumulate=True, mul=0.25, sched=(steps, scale, [(0, -1, 0.1)]))
    assert abs(float(o2) - float(out)) < 2e-6 * want and steps.tolist() == [5] and float(scale) == 1.0
    first = float(out)
    for _ in range(2):
        o3 = torch.zeros(1, device=DEV)
        ops.clip_norm_bf16(g, spans, o3, None, mul=0.25)
        assert float(o3) == first
    with pytest.raises(RuntimeError, match="multiple of 8"):
        ops.clip_norm_bf16(g, [(4, 100)], o2)


# ------------------------------------------------------------------------------------------------ fp8 forward (C5)
def _e4m3(x, q):
    """reference quantisation: torch's float8_e4m3fn cast of clamp(x * q, +-448), as uint8 bits"""
    return (x.float() * q).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)


def _deq(x8):
    return x8.view(torch.float8_e4m3fn).float()


@pytest.mark.parametrize("tile", [0, 1, 2, 3])
def test_gemm_grouped_fp8_matches_dequantised_products(ops, tile):
    """xggm_gemm_grouped_fp8e4m3: the four forward products of one cross-attention round (two Q, two fused KV) and an
    FFN pair incl. GELU + pre-activation + e4m3 copy of the result and the split-K fp32 slabs, against the fp64
    product of the SAME e4m3 operands; every tile of the grouped kernel."""
    from xggm_amd import _lib
    BF = torch.bfloat16
    g = torch.Generator().manual_seed(11)

    def operand(M, K, scale):
        x = torch.randn(M, K, generator=g) * scale
        a = x.abs().max().reshape(1).to(DEV)
        q, s = ops.fp8_scale_for(a)
        return ops.quantize_fp8(x.to(BF).to(DEV), q).view(torch.uint8), s

    _lib.lib.xggm_gemm_set_group_tile(tile)
    try:
        # cross-attention round: lang queries, vision keys/values, vision queries, lang keys/values
        xl, sl = operand(640, 768, 1.0)
        xv, sv = operand(1152, 768, 1.5)
        wq, sq = operand(768, 768, 0.03)
        wkv, skv = operand(1536, 768, 0.03)
        bq = torch.randn(768, generator=g).to(DEV)
        bkv = torch.randn(1536, generator=g).to(DEV)
        probs, outs = [], []
        for x8, sx, w8, sw, b in ((xl, sl, wq, sq, bq), (xv, sv, wkv, skv, bkv), (xv, sv, wq, sq, bq), (xl, sl, wkv, skv, bkv)):
            p, y, _ = ops.p_fwd8(x8, w8, sx, sw, b)
            probs.append(p)
            outs.append((y, x8, sx, w8, sw, b))
        ops.gemm_group8(probs)
        for y, x8, sx, w8, sw, b in outs:
            ref = _deq(x8).double() @ _deq(w8).double().t() * float(sx) * float(sw) + b.double()
            assert rel_err(y, ref) < 4e-3
        # FFN pair: intermediate product with GELU, pre-activation and the e4m3 copy of the activation
        w1, s1 = operand(3072, 768, 0.03)
        b1 = torch.randn(3072, generator=g).to(DEV)
        res = []
        probs = []
        for x8, sx in ((xl, sl), (xv, sv)):
            act8 = torch.empty((x8.shape[0], 3072), device=DEV, dtype=torch.uint8)
            qa = torch.tensor([37.0], device=DEV)
            amax = torch.zeros(1, device=DEV)
            p, act, u = ops.p_fwd8(x8, w1, sx, s1, b1, act=ops.ACT_GELU, want_preact=True, emit8=(act8, qa, amax))
            probs.append(p)
            res.append((x8, sx, act, u, act8, qa, amax))
        ops.gemm_group8(probs)
        for x8, sx, act, u, act8, qa, amax in res:
            pre = _deq(x8).double() @ _deq(w1).double().t() * float(sx) * float(s1) + b1.double()
            assert rel_err(u, pre) < 4e-3
            want = torch.nn.functional.gelu(u.float())
            assert rel_err(act, want) < 4e-3
            # the e4m3 copy is the quantised fp32 activation (before its bf16 rounding): compare through the values
            assert rel_err(_deq(act8) / 37.0, want) < 5e-2  # 3 mantissa bits: <= 6.25 % per element, ~3 % rms
            a = float(want.abs().max())
            assert float(amax) == 0.0 or abs(float(amax) - a) < 1e-2 * a  # recorded only beyond half the range 448/37
            assert (a > 0.5 * 448 / 37) == (float(amax) > 0)
        # an uncalibrated entry (qscale <= 0): quantised with 1, maximum always recorded
        act8 = torch.empty((640, 3072), device=DEV, dtype=torch.uint8)
        qa, amax = torch.tensor([-1.0], device=DEV), torch.zeros(1, device=DEV)
        p, act, u = ops.p_fwd8(xl, w1, sl, s1, b1, act=ops.ACT_GELU, want_preact=True, emit8=(act8, qa, amax))
        ops.gemm_group8([p])
        want = torch.nn.functional.gelu(u.float())
        assert abs(float(amax) - float(want.abs().max())) < 1e-2 * float(want.abs().max())
        assert rel_err(_deq(act8), want) < 5e-2
        # FFN output product as 3 split-K slabs
        a8, sa = operand(1152, 3072, 0.5)
        w2, s2 = operand(768, 3072, 0.03)
        p, part, _ = ops.p_fwd8(a8, w2, sa, s2, None, split=3)
        ops.gemm_group8([p])
        ref = _deq(a8).double() @ _deq(w2).double().t() * float(sa) * float(s2)
        assert part.shape == (3, 1152, 768) and rel_err(part.sum(0), ref) < 1e-4
    finally:
        _lib.lib.xggm_gemm_set_group_tile(0)
    with pytest.raises(RuntimeError, match="multiples of 16"):
        p, _, _ = ops.p_fwd8(xl[:, :760], wq[:, :760], sl, sq, None)
        ops.gemm_group8([p])


def test_producers_emit_e4m3_copies(ops):
    """the residual LayerNorm and the attention core write their output a second time as e4m3 (the operand of the
    next fp8 product) with the site's scale, and record the maximum by the scale-table protocol"""
    BF = torch.bfloat16
    # LayerNorm: two problems in one launch, each with its own scale entry
    reqs, refs = [], []
    for i, M in enumerate((640, 1152)):
        h, _ = rnd((M, 768), BF, 20 + i)
        r, _ = rnd((M, 768), BF, 30 + i)
        gam = (1 + 0.1 * torch.randn(768, generator=torch.Generator().manual_seed(40 + i))).to(DEV)
        bet = (0.1 * torch.randn(768, generator=torch.Generator().manual_seed(50 + i))).to(DEV)
        q = torch.tensor([50.0 if i == 0 else -1.0], device=DEV)  # calibrated / uncalibrated entry
        amax = torch.zeros(1, device=DEV)
        req = ops.LnFwdReq(h.clone(), None, r, gam, bet, 1e-12, emit8=(q, amax))
        reqs.append(req)
        refs.append((q, amax))
    ops.launch_row_requests(reqs)
    for req, (q, amax) in zip(reqs, refs):
        out = req.out.float()
        qq = float(q) if float(q) > 0 else 1.0
        # quantised from the fp32 value the bf16 output was rounded from: equal up to one bf16 rounding before the cast
        d = (_deq(req.out8) / qq - out).abs()
        # (element-wise: <= 1/16 of the value in the normal range; with scale 1 values below 2^-6 sit on the 2^-9 subnormal grid)
        assert float((d / (out.abs() + 2e-2)).max()) < 0.08 and rel_err(_deq(req.out8) / qq, out) < 5e-2
        m = float(out.abs().max())
        if float(q) > 0:
            assert (float(amax) > 0) == (m > 0.5 * 448 / qq)
        if float(amax) > 0:
            assert abs(float(amax) - m) < 1e-2 * m
    assert float(refs[1][1]) > 0  # the uncalibrated entry always records
    # attention core: context as bf16 and as e4m3
    B, heads, Sq, Sk = 3, 12, 20, 36
    qh, _ = rnd((B * Sq, 768), BF, 60)
    kh, _ = rnd((B * Sk, 768), BF, 61)
    vh, _ = rnd((B * Sk, 768), BF, 62)
    qs, amax = torch.tensor([100.0], device=DEV), torch.zeros(1, device=DEV)
    rng = ops.make_rng(1, DEV)
    a = ops.AttnFwdReq(qh, kh, vh, None, B, heads, Sq, Sk, 0.0, rng, 7, emit8=(qs, amax))
    b = ops.AttnFwdReq(qh, kh, vh, None, B, heads, Sq, Sk, 0.0, rng, 7)
    ops.launch_row_requests([a])
    ops.launch_row_requests([b])
    assert torch.equal(a.out, b.out)
    assert torch.equal(a.out8.cpu(), _e4m3(a.out, 100.0).cpu())  # quantised from the bf16 value: bit-exact
    m = float(a.out.float().abs().max())
    assert (float(amax) > 0) == (m > 0.5 * 448 / 100) and (float(amax) == 0 or float(amax) == m)


def test_bertadam_ex_writes_e4m3_shadow_and_reads_bf16_gradients(ops):
    """xggm_bertadam_ex against xggm_bertadam_f32 on the same state: identical p / m / v / bf16 shadow; the e4m3 copy
    equals the quantised new weights chunk by chunk with the table's scales, chunks with id 0 stay untouched, maxima
    beyond 3/4 of an entry's range are recorded; bf16 gradients give the update of their fp32 values; the learning
    rate can come from a device scalar."""
    n = 256 * 40
    g_ = torch.Generator().manual_seed(3)
    p0 = (torch.randn(n, generator=g_) * 0.02).to(DEV)
    gr = (torch.randn(n, generator=g_) * 1e-3).to(DEV)
    m0 = (torch.randn(n, generator=g_) * 1e-4).to(DEV)
    v0 = (torch.rand(n, generator=g_) * 1e-6).to(DEV)
    sq = (gr.double() ** 2).sum().float().reshape(1)
    lr_scale = torch.tensor([0.5], device=DEV)

    def state():
        return p0.clone(), m0.clone(), v0.clone(), torch.zeros(n, device=DEV, dtype=torch.bfloat16)

    pa, ma, va, sa = state()
    ops.bertadam(pa, gr, ma, va, sa, sq, 5.0, 1e-2, lr_scale, 0.9, 0.999, 1e-6, 0.01)
    # e4m3 copy: chunks 0-9 entry 1, 10-19 none, 20-39 entry 2; the range starts at arena element 256 * 7
    ids = torch.zeros(64, dtype=torch.int16, device=DEV)
    ids[7:17] = 1
    ids[27:47] = 2
    big = float(p0.abs().max())
    qtab = torch.tensor([1.0, 448.0 / (2.0 * big), 448.0 / (1.1 * big)], device=DEV)  # entry 1: max at half the range
    amax = torch.zeros(3, device=DEV)
    s8 = torch.full((n,), 0x55, dtype=torch.uint8, device=DEV)
    pb, mb, vb, sb = state()
    ops.bertadam_ex(pb, gr, mb, vb, sb, sq, 5.0, 123.0, lr_scale, 0.9, 0.999, 1e-6, 0.01,
                    lr_dev=torch.tensor([1e-2], device=DEV), w8=(s8, ids, qtab, amax), elem0=256 * 7)
    assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb) and torch.equal(sa, sb)
    assert torch.equal(s8[:2560].cpu(), _e4m3(pb[:2560], float(qtab[1])).cpu())
    assert bool((s8[2560:5120] == 0x55).all())
    assert torch.equal(s8[5120:].cpu(), _e4m3(pb[5120:], float(qtab[2])).cpu())
    assert float(amax[1]) == 0.0  # nothing beyond 3/4 of entry 1's range
    assert float(amax[2]) == float(pb[5120:].abs().max()) and float(amax[0]) == 0.0
    # bf16 gradients
    gb = gr.to(torch.bfloat16)
    pc, mc, vc, sc = state()
    ops.bertadam(pc, gb.float(), mc, vc, sc, sq, 5.0, 1e-2, lr_scale, 0.9, 0.999, 1e-6, 0.01)
    pd, md, vd, sd = state()
    ops.bertadam_ex(pd, gb, md, vd, sd, sq, 5.0, 1e-2, lr_scale, 0.9, 0.999, 1e-6, 0.01)
    assert torch.equal(pc, pd) and torch.equal(mc, md) and torch.equal(vc, vd) and torch.equal(sc, sd)
    # squared norm of a bf16 buffer
    out = torch.zeros(1, device=DEV)
    ops.sqnorm_bf16(gb, out)
    assert abs(float(out) - float((gb.double() ** 2).sum())) < 1e-5 * float((gb.double() ** 2).sum())


def test_fp8_scale_update_protocol(ops):
    """xggm_fp8_scale_update on a 6-entry table over several calls: recorded maxima set range = margin x max of the
    history, an uncalibrated entry stays uncalibrated until something is recorded, a silent entry keeps its scale
    (weights) or halves its range once per wrapped history (activations), amax is cleared, pos advances only when
    asked, and a sub-range call leaves the other entries alone."""
    H = 4
    amax = torch.zeros(8, device=DEV)
    hist = torch.zeros(8 * H, device=DEV)
    q = torch.tensor([1.0, -1.0, -1.0, 10.0, 10.0, 10.0, 7.0, 7.0], device=DEV)
    d = torch.ones(8, device=DEV)
    pos = torch.zeros(1, dtype=torch.int64, device=DEV)
    amax[1], amax[3] = 2.0, 5.0
    ops.fp8_scale_update(amax, hist, q, d, pos, 1, 5, H, 1.25, 1, 1)  # entries 1..5
    assert int(pos) == 1 and float(amax.abs().max()) == 0.0
    assert abs(float(q[1]) - 448 / (1.25 * 2.0)) < 1e-3 and float(q[2]) == -1.0          # calibrated / still not
    assert abs(float(q[3]) - 448 / (1.25 * 5.0)) < 1e-3 and float(q[4]) == 10.0         # recorded / silent: kept
    assert float(q[0]) == 1.0 and float(q[6]) == 7.0                                      # outside the range
    assert abs(float(d[3]) * float(q[3]) - 1.0) < 1e-6
    for step in range(2, 5):  # three more silent calls: the history of entry 3 still holds 5.0 until it wraps
        ops.fp8_scale_update(amax, hist, q, d, pos, 1, 5, H, 1.25, 1, 1)
        assert int(pos) == step
    assert abs(float(q[3]) - 448 / (1.25 * 5.0)) < 1e-3
    assert float(q[4]) == 20.0 and float(q[5]) == 20.0  # a whole history without a near-range value: range halves
    ops.fp8_scale_update(amax, hist, q, d, pos, 1, 5, H, 1.25, 1, 1)  # slot 0 again: the 5.0 is overwritten
    assert abs(float(q[3]) - 448 / (1.25 * 5.0)) < 1e-3  # kept until the history wraps with nothing recorded
    # weights: no shrink, no bump
    amax[6] = 3.0
    ops.fp8_scale_update(amax, hist, q, d, pos, 6, 2, H, 4.0 / 3.0, 0, 0)
    assert int(pos) == 5 and abs(float(q[6]) - 448 / (4.0 / 3.0 * 3.0)) < 1e-3 and float(q[7]) == 7.0


def test_fp8_amax_entries_spread_over_slots(ops):
    """an amax entry may be spread over 2^k floats (xggm_*_problem.amax_slots): workgroup w of a producer raises slot
    w % slots -- hundreds of workgroups raising one address were 0.25 ms of the fp8 iteration -- and the scale update
    takes the maximum over the slots and clears them all.  Producers: LayerNorm forward, attention forward, the GELU
    epilogue; a 1-float entry keeps the old behaviour (the other fp8 tests)."""
    BF = torch.bfloat16
    S = 64
    gen = torch.Generator().manual_seed(11)
    # LayerNorm forward
    M, H = 640, 768
    h, r = torch.randn(M, H, generator=gen).to(BF).to(DEV), torch.randn(M, H, generator=gen).to(BF).to(DEV)
    gam, bet = torch.ones(H, device=DEV), torch.zeros(H, device=DEV)
    outs = []
    for slots in (1, S):
        q, amax = torch.tensor([200.0], device=DEV), torch.zeros(slots, device=DEV)
        req = ops.LnFwdReq(h.clone(), None, r, gam, bet, 1e-12, emit8=(q, amax))
        ops.launch_row_requests([req])
        outs.append((req.out8.clone(), amax.clone()))
    assert torch.equal(outs[0][0], outs[1][0])
    assert float(outs[0][1].max()) > 0 and float(outs[1][1].max()) == float(outs[0][1][0])
    assert int((outs[1][1] > 0).sum()) > 8  # really spread: 80 workgroups over 64 slots
    # attention forward
    B, heads, Sq = 8, 12, 36
    qkv = [torch.randn(B * Sq, 768, generator=gen).to(BF).to(DEV) for _ in range(3)]
    rng = ops.make_rng(3, DEV)
    am = []
    for slots in (1, S):
        qs, amax = torch.tensor([1000.0], device=DEV), torch.zeros(slots, device=DEV)
        a = ops.AttnFwdReq(qkv[0], qkv[1], qkv[2], None, B, heads, Sq, Sq, 0.0, rng, 7, emit8=(qs, amax))
        ops.launch_row_requests([a])
        am.append(amax.clone())
    assert float(am[0][0]) > 0 and float(am[1].max()) == float(am[0][0]) and int((am[1] > 0).sum()) > 8
    # the scale update reduces and clears the slots
    Hh = 4
    amax = torch.zeros(3 * S, device=DEV)
    amax[1 * S + 17], amax[1 * S + 40], amax[2 * S + 63] = 2.0, 5.0, 3.0
    hist, q, d = torch.zeros(3 * Hh, device=DEV), torch.full((3,), -1.0, device=DEV), torch.ones(3, device=DEV)
    pos = torch.zeros(1, dtype=torch.int64, device=DEV)
    ops.fp8_scale_update(amax, hist, q, d, pos, 1, 2, Hh, 1.25, 1, 1, slots=S)
    assert float(amax.abs().max()) == 0.0 and float(q[0]) == -1.0
    assert abs(float(q[1]) - 448 / (1.25 * 5.0)) < 1e-3 and abs(float(q[2]) - 448 / (1.25 * 3.0)) < 1e-3
    with pytest.raises(RuntimeError, match="power of two"):
        ops.fp8_scale_update(amax, hist, q, d, pos, 0, 1, Hh, 1.25, 1, 1, slots=48)


# ------------------------------------------------------------------------------------------------ preprocessing
def test_cosine_adjacency_matches_reference_golden_and_oracle(ops):
    """xggm_cosine_adjacency_f32 (data/preprocess/vqa/compute_adjacency.py:38-45, :90) against the fixture the
    reference's own function produced (ties at the maximum, a zero embedding under the eps clamp) and against the
    oracle's double loop at other sizes (N = 5, 36, 64; D = 768 and a width that is not a multiple of the 64-wide
    chunk); the batched builder gathers label embeddings by id."""
    from helpers import load_golden
    from oracle import xggm_oracle as O
    from xggm_amd import synth
    from xggm_amd.preprocess.compute_adjacency import compute_cosin_sim_v2, build_adjacency
    g = load_golden("adjacency")
    seed, D = int(g["seed"]), int(g["D"])
    cls = torch.from_numpy(synth._rng(seed, "adj_class_table").standard_normal((60, D), dtype=np.float32))
    att = torch.from_numpy(synth._rng(seed, "adj_attr_table").standard_normal((45, D), dtype=np.float32))
    att[7] = 0.0
    att[8] = cls[8]
    oid, aid = torch.from_numpy(g["objects_id"]), torch.from_numpy(g["attrs_id"])
    adj = build_adjacency(cls.to(DEV), att.to(DEV), oid, aid, chunk=3)
    assert adj.shape == (4, 36, 36)
    # fp32 sums over D = 768 in another order than torch's, then a division by the maximum: a few ulp of 1
    assert float((adj.cpu() - torch.from_numpy(g["adj"])).abs().max()) < 5e-6
    assert float(adj.max()) == 1.0 and torch.equal(adj, adj.transpose(1, 2))
    gen = torch.Generator().manual_seed(2)
    for N, Dm in ((5, 768), (36, 100), (64, 768), (1, 64)):
        a = torch.randn(3, N, Dm, generator=gen)
        b = torch.randn(3, N, Dm, generator=gen)
        got = compute_cosin_sim_v2(a.to(DEV), b.to(DEV), normalize=True).cpu()
        for i in range(3):
            assert float((got[i] - O.adjacency_of(a[i], b[i])).abs().max()) < 5e-6, (N, Dm)
        raw = compute_cosin_sim_v2(a[0].to(DEV), b[0].to(DEV)).cpu()  # the reference signature: one image, no / max
        assert float((raw - O.compute_cosin_sim_v2(a[0], b[0])).abs().max()) < 1e-5
    with pytest.raises(RuntimeError, match="multiple of 4"):
        compute_cosin_sim_v2(torch.zeros(1, 4, 6, device=DEV), torch.zeros(1, 4, 6, device=DEV))
    with pytest.raises(RuntimeError, match="GPU"):
        compute_cosin_sim_v2(torch.zeros(4, 8), torch.zeros(4, 8))


def test_add_n_pad_rows_and_fan_out(ops):
    """the kernels that replaced the framework's glue inside a pass: xggm_add_n (sum of 2..4 tensors, fp32 arithmetic,
    one rounding; may write over an input), xggm_pad_rows_bf16 (cast + zero-padded row stride in one launch) and
    functional.fan_out (gradients of several consumers meet in ONE launch; values and gradients of plain reuse)."""
    from xggm_amd import functional as XF
    F32, BF = torch.float32, torch.bfloat16
    gen = torch.Generator().manual_seed(5)
    for dt in (F32, BF):
        ts = [torch.randn(7, 33, 5, generator=gen).to(dt) for _ in range(4)]
        for n in (2, 3, 4):
            got = ops.add_n([t.to(DEV) for t in ts[:n]]).cpu()
            ref = sum(t.double() for t in ts[:n]).to(dt)  # fp64 sum rounded once == fp32 sum rounded once here
            assert float((got.double() - ref.double()).abs().max()) <= (0 if dt == F32 else 1e-2) + 1e-6
        a, b = ts[0].to(DEV), ts[1].to(DEV)
        want = ops.add_n([a, b]).clone()
        assert torch.equal(ops.add_n([a, b], out=a), want) and torch.equal(a, want)
    x = torch.randn(5, 630, generator=gen)
    for src in (x, x.to(BF)):
        p = ops.pad_rows(src.to(DEV), 632)
        assert p.shape == (5, 630) and p.stride() == (632, 1) and p.dtype == BF
        assert torch.equal(p.cpu(), src.to(BF))
        full = torch.as_strided(p, (5, 632), (632, 1))
        assert float(full[:, 630:].abs().max()) == 0.0
    with pytest.raises(RuntimeError, match="pad_rows"):
        _lib_call_bad_pad()
    # fan_out: three consumers with different weights
    w = [0.5, -2.0, 3.0]
    for dt in (F32, BF):
        v = torch.randn(4, 36, 16, generator=gen).to(dt).to(DEV).requires_grad_(True)
        parts = XF.fan_out(v, 3)
        assert all(torch.equal(p, v) for p in parts)
        sum((p.float() * c).sum() for p, c in zip(parts, w)).backward()
        ref = torch.full_like(v, sum(w))
        assert float((v.grad.float() - ref.float()).abs().max()) < (1e-6 if dt == F32 else 2e-2)
    v = torch.randn(3, 8, device=DEV, requires_grad=True)
    a, b = XF.fan_out(v, 2)
    (a.sum() * 2).backward()  # one consumer never used: its gradient is None, not a zero fill
    assert torch.equal(v.grad, torch.full_like(v, 2.0))


def _lib_call_bad_pad():
    from xggm_amd import _lib
    x = torch.zeros(4, 8, device=DEV)
    out = torch.zeros(4, 4, device=DEV, dtype=torch.bfloat16)
    _lib.call("xggm_pad_rows_bf16", x.data_ptr(), 1, out.data_ptr(), 4, 8, 4, None)


def test_gather_and_scatter_rows(ops):
    """xggm_gather_rows_bf16 / xggm_scatter_rows_bf16: the rows of the word-embedding gradient a data-parallel step
    exchanges (dist.GradSync.set_sparse_table); duplicates, out-of-range indices, a width that is not a multiple of 256."""
    gen = torch.Generator().manual_seed(4)
    for V, H in ((100, 768), (37, 36)):
        t = torch.randn(V, H, generator=gen)
        idx = torch.tensor([0, 5, 5, V - 1, 17, -1, V, 3], dtype=torch.int64)
        got = ops.gather_rows(t.to(DEV), idx.to(DEV)).cpu()
        ok = (idx >= 0) & (idx < V)
        assert torch.equal(got[ok], t[idx[ok]].to(torch.bfloat16))
        dst = torch.zeros(V, H, dtype=torch.bfloat16, device=DEV)
        rows = torch.randn(8, H, generator=gen).to(torch.bfloat16)
        rows[2] = rows[1]  # duplicates carry identical rows
        ops.scatter_rows(rows.to(DEV), idx.to(DEV), dst)
        want = torch.zeros(V, H, dtype=torch.bfloat16)
        for i in range(8):
            if ok[i]:
                want[idx[i]] = rows[i]
        assert torch.equal(dst.cpu(), want)


def test_embedding_gradient_with_heavy_duplicates(ops):
    """the owner-gather of the embedding gradients: one token in EVERY row (600 occurrences: more than one round of the
    owner's list), a token that occurs once, segment id 1 on most rows -- against index_add in fp64, and the padding rows
    (index 0 of every table) untouched"""
    dt = torch.bfloat16
    B, T, H, V = 30, 20, 128, 16
    gen = torch.Generator().manual_seed(3)
    ids = torch.full((B, T), 7, dtype=torch.long)
    ids[3, 5] = 11
    ids[:, -1] = 0
    seg = torch.ones_like(ids)
    seg[0] = 0
    M = B * T
    dy, _ = rnd((M, H), dt, 2)
    z, _ = rnd((M, H), dt, 4)
    # statistics of z rows, so that the LayerNorm backward in front of the scatter is the real one
    zf = z.float()
    stats = torch.stack([zf.mean(1), torch.rsqrt(zf.var(1, unbiased=False) + 1e-12)], 1).contiguous()
    gamma = torch.ones(H, device=DEV)
    g = {k: torch.zeros(s, device=DEV) for k, s in (("w", (V, H)), ("p", (32, H)), ("t", (2, H)), ("g", (H,)), ("b", (H,)))}
    ops.embed_bwd(ids.to(DEV), seg.to(DEV), dy, z, stats, gamma, g["w"], g["p"], g["t"], g["g"], g["b"], 0.0, None, 0)
    # the dz rows the kernel scattered: recompute the LN backward in fp64 from the same bf16 inputs
    xh = (zf.double().cpu() - stats[:, :1].double().cpu()) * stats[:, 1:].double().cpu()
    dyd = dy.double().cpu()
    dz = stats[:, 1:].double().cpu() * (dyd - dyd.mean(1, keepdim=True) - xh * (dyd * xh).mean(1, keepdim=True))
    dz = dz.to(dt).double()  # the kernel stores dz in the activation type before gathering it
    flat, segf = ids.view(-1), seg.view(-1)
    pos = torch.arange(M) % T
    for name, key, n in (("w", flat, V), ("p", pos, 32), ("t", segf, 2)):
        want = torch.zeros(n, H, dtype=torch.float64).index_add_(0, key, dz)
        want[0] = 0  # padding_idx
        # fp32 on the device against fp64 here: a handful of dz elements round to the neighbouring bf16 value
        assert rel_err(g[name], want) < 2e-4, name
        assert float(g[name][0].abs().max()) == 0.0


@pytest.mark.parametrize("dt", DTS)
def test_reductions_are_reproducible_bit_for_bit(ops, dt):
    """No floating-point atomics on the path: the sums many workgroups contribute to (bias gradients out of the GEMM
    epilogue and the attention backward, embedding-table gradients, losses, GIN's eps gradient) have a fixed order, so
    repeating a launch gives the same BITS -- also while another stream keeps the GPU busy with something else and the
    workgroups therefore finish in another order.  (src/param.py:129-132: the reference seeds everything; its runs
    are reproducible too.)"""
    B, S, heads = 8, 36, 2
    H = heads * 64
    M = B * S
    side = torch.cuda.Stream()
    noise = torch.randn(1 << 22, device=DEV)

    def disturb(k):  # different amounts of foreign work in flight for every repetition
        with torch.cuda.stream(side):
            for _ in range(k):
                noise.mul_(1.0001)

    dy, _ = rnd((M, H), dt, 1)
    w, _ = rnd((H, 4 * H), dt, 2, 0.05)
    u, _ = rnd((M, 4 * H), dt, 3)
    qkv, _ = rnd((M, 3 * H), dt, 4)
    do, _ = rnd((M, H), dt, 5)
    ids = torch.randint(1, 20, (B, 20), generator=torch.Generator().manual_seed(6)).to(DEV)
    z, _ = rnd((B * 20, H), dt, 7)
    stats = torch.stack([z.float().mean(1), torch.rsqrt(z.float().var(1, unbiased=False) + 1e-12)], 1).contiguous()
    dye, _ = rnd((B * 20, H), dt, 8)
    logit = torch.randn(B, 2274, device=DEV)
    target = (torch.rand(B, 2274, device=DEV) > 0.99).float()
    adj = torch.rand(B, S, S, device=DEV)
    x3, _ = rnd((B, S, H), dt, 9)
    dh3, _ = rnd((B, S, H), dt, 10)
    rng = ops.make_rng(5, DEV)

    def once(k):
        disturb(k)
        cs = torch.zeros(4 * H, device=DEV)
        p, dx = ops.p_dgrad(dy, w, gelu_aux=u, colsum=cs)
        ops.gemm_group(dt, [p])
        gb = torch.zeros(3 * H, device=DEV)
        dqkv = torch.empty_like(qkv)
        ops.attn_bwd(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], None, do, dqkv[:, :H], dqkv[:, H:2 * H], dqkv[:, 2 * H:],
                     B, heads, S, S, 0.1, rng, 3, gb[:H], gb[H:2 * H], gb[2 * H:])
        ge = [torch.zeros(s, device=DEV) for s in ((20, H), (32, H), (2, H), (H,), (H,))]
        ops.embed_bwd(ids, None, dye, z, stats, torch.ones(H, device=DEV), *ge, 0.0, None, 0)
        losses = torch.stack([ops.bce_fwd(logit, target, 1.0 / logit.numel()),
                              ops.dsm_fwd(x3, dh3.float(), 0.5 / x3.numel()),
                              ops.symkl_fwd(x3, dh3, 1.0 / x3.numel())])
        eps = torch.zeros(1, device=DEV)
        ops.agg_dot(adj, x3, dh3, eps)
        torch.cuda.synchronize()
        return [cs, dx, gb, dqkv, ge[0], ge[1], losses, eps]

    first = once(0)
    for k in (3, 11, 1):
        again = once(k)
        for i, (a, b) in enumerate(zip(first, again)):
            assert torch.equal(a, b), (i, k, float((a.float() - b.float()).abs().max()))



def test_prefetch_ranges_ride_on_the_next_row_launch_and_change_nothing(ops):
    """xggm_prefetch_next queues byte ranges that the next LayerNorm / attention launch reads beside its own rows and
    discards (weights for the products behind it): same outputs bit for bit with and without a
    queued range, odd sizes and a fifth range included, and the queue is empty afterwards (a second launch is plain)."""
    from xggm_amd import _lib
    dt = torch.bfloat16
    x, _ = rnd((1152, 768), dt, 1)
    res, _ = rnd((1152, 768), dt, 2)
    g = torch.randn(768, generator=torch.Generator().manual_seed(3)).to(DEV)
    b = torch.randn(768, generator=torch.Generator().manual_seed(4)).to(DEV)
    w = torch.randn(5 * 768 * 768 + 24, device=DEV).to(dt)

    def ln():
        out, z, st = ops.ln_fwd(x.clone(), None, res, g, b, 1e-12)
        torch.cuda.synchronize()
        return out, st

    ref = ln()
    for t in (w, w[8:8 + 3 * 768 * 768 + 8], w[:64], w[16:100000], w[:1000]):  # five: the last one is dropped
        ops.prefetch_next(t)
    got = ln()
    again = ln()
    assert torch.equal(ref[0], got[0]) and torch.equal(ref[1], got[1])
    assert torch.equal(ref[0], again[0])
    assert _lib.lib.xggm_prefetch_next(w.data_ptr() + 2, 4096) != 0  # misaligned: refused on the host
    assert b"16-byte" in _lib.lib.xggm_last_error()
    # the attention core as carrier
    B, heads, S, H = 4, 12, 36, 768
    q, _ = rnd((B * S, H), dt, 5)
    k, _ = rnd((B * S, H), dt, 6)
    v, _ = rnd((B * S, H), dt, 7)

    def att():
        r = ops.AttnFwdReq(q, k, v, None, B, heads, S, S, 0.0, None, 0)
        ops.launch_row_requests([r])
        torch.cuda.synchronize()
        return r.out

    ref_a = att()
    ops.prefetch_next(w)
    assert torch.equal(ref_a, att())

def test_tile_choice_changes_speed_only(ops):
    """ops.gemm_group pins the tile of a launch from the measured table (gemm_tiles_gfx950.json).  Every tile walks k
    in the same order and the column sums are taken per 32 output rows whatever the tile, so the heaviest launch of the
    step -- FFN2 backward: dgrad * gelu'(u) + column sums, weight gradient -- gives the SAME BITS under every tile; the
    64 x 64 norm slots are the same numbers summed in a tile-dependent (but fixed) order: equal to fp32 rounding.  The
    table is part of the configuration: one table, one set of bits.  And the table really steers the launch."""
    from xggm_amd import _lib
    dt = torch.bfloat16
    M, H, I = 640, 768, 3072
    d_h, _ = rnd((M, H), dt, 1)
    w2, _ = rnd((H, I), dt, 2, 0.05)
    u, _ = rnd((M, I), dt, 3)
    act, _ = rnd((M, I), dt, 4)

    def run(pin):
        _lib.lib.xggm_gemm_set_group_tile(pin)
        try:
            cs = torch.zeros(I, device=DEV)
            gw = torch.empty(H, I, device=DEV)
            sq = torch.zeros((H // 64) * (I // 64), device=DEV)
            pd, dx = ops.p_dgrad(d_h, w2, gelu_aux=u, colsum=cs)
            pw = ops.p_wgrad(d_h, act, gw, False, sqsum=sq)
            saved, ops.TILE_TABLE = ops.TILE_TABLE, {}  # the explicit pin, not the table, chooses here
            try:
                ops.gemm_group(dt, [pw, pd])
            finally:
                ops.TILE_TABLE = saved
            torch.cuda.synchronize()
            return dx, cs, gw, sq
        finally:
            _lib.lib.xggm_gemm_set_group_tile(0)

    ref = run(1)
    for pin in (0, 2, 3, 4, 7, 8, 9):
        got = run(pin)
        for a, b in zip(ref[:3], got[:3]):  # dx, column sums, weight gradient
            assert torch.equal(a, b), pin
        assert torch.allclose(ref[3], got[3], rtol=2e-6, atol=0), pin  # 4096 fp32 squares per slot, another order
        assert torch.equal(got[3], run(pin)[3]), pin                   # ... and the same order every time
    # the table: a signature that is in it reaches the library as a pin (seen through the hook), others do not
    sig_seen = []
    saved_t, saved_h = ops.TILE_TABLE, ops.TILE_HOOK
    try:
        cs = torch.zeros(I, device=DEV)
        pd, _ = ops.p_dgrad(d_h, w2, gelu_aux=u, colsum=cs)
        pw = ops.p_wgrad(d_h, act, torch.empty(H, I, device=DEV), False)
        sig = ops.gemm_signature(dt, [pw, pd])
        assert sig == "bf16|768x3072x640:00f+640x3072x768:10gc"
        ops.TILE_TABLE = {sig: 1}
        ops.TILE_HOOK = lambda dt_, chunk, s, arr: sig_seen.append((s, ops.TILE_TABLE.get(s, 0))) or 0
        ops.gemm_group(dt, [pw, pd])
        assert sig_seen == [(sig, 1)]
    finally:
        ops.TILE_TABLE, ops.TILE_HOOK = saved_t, saved_h
