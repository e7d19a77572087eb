"""Row-kernel timings at the training shapes (hipGraph replays, one launch per graph node; us per launch):
residual LayerNorm forward (attention block: bf16 input; FFN block: 3 fp32 split-K slabs) and backward, paired
(language + vision rows in one launch) as the step issues them; the attention core forward / backward."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xggm_amd import ops  # noqa: E402

BF = torch.bfloat16
dev = "cuda"


def timeit(fn, n=200):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * n)


def ln_fwd(slabs, p=0.1, emit8=False):
    rng = ops.make_rng(1, dev)
    qs, am = torch.full((1,), 8.0, device=dev), torch.zeros(1, device=dev)
    H = 768
    gam, bet, bias = torch.ones(H, device=dev), torch.zeros(H, device=dev), torch.zeros(H, device=dev)
    reqs_in = []
    for M in (640, 1152):
        x = torch.randn((slabs, M, H) if slabs else (M, H), device=dev).to(torch.float32 if slabs else BF)
        r = torch.randn(M, H, device=dev).to(BF)
        reqs_in.append((x, r))

    def run():
        reqs = [ops.LnFwdReq(x, bias, r, gam, bet, 1e-12, p_pre=p, rng=rng, sid_pre=3, dtype=BF,
                             emit8=(qs, am) if emit8 else None) for x, r in reqs_in]
        ops.launch_row_requests(reqs)
        return reqs
    return run


def ln_bwd(p=0.1):
    rng = ops.make_rng(1, dev)
    H = 768
    gam = torch.ones(H, device=dev)
    data = []
    for M in (640, 1152):
        dy = torch.randn(M, H, device=dev).to(BF)
        z = torch.randn(M, H, device=dev).to(BF)
        st = torch.rand(M, 2, device=dev) + 0.5
        data.append((dy, z, st))
    keep = []

    def run():
        reqs = [ops.LnBwdReq(dy, z, st, gam, None, None, None, p_pre=p, rng=rng, sid_pre=3, defer=keep) for dy, z, st in data]
        ops.launch_row_requests(reqs)
        del keep[:]
    return run


def attn(bwd, emit8=False):
    rng = ops.make_rng(1, dev)
    qs, am = torch.full((1,), 8.0, device=dev), torch.zeros(1, device=dev)
    B, heads = 32, 12
    shapes = ((20, 20), (36, 36))
    data = []
    for Sq, Sk in shapes:
        qkv = torch.randn(B * Sq, 2304, device=dev).to(BF)
        d_out = torch.randn(B * Sq, 768, device=dev).to(BF)
        dqkv = torch.empty_like(qkv)
        mask = torch.zeros(B, Sk, device=dev)
        data.append((qkv, d_out, dqkv, mask, Sq, Sk))

    def run():
        reqs = []
        for qkv, d_out, dqkv, mask, Sq, Sk in data:
            q, k, v = qkv[:, :768], qkv[:, 768:1536], qkv[:, 1536:]
            if bwd:
                reqs.append(ops.AttnBwdReq(q, k, v, mask, d_out, dqkv[:, :768], dqkv[:, 768:1536], dqkv[:, 1536:], B, heads, Sq, Sk,
                                           0.1, rng, 5))
            else:
                reqs.append(ops.AttnFwdReq(q, k, v, mask, B, heads, Sq, Sk, 0.1, rng, 5, emit8=(qs, am) if emit8 else None))
        ops.launch_row_requests(reqs)
    return run


if __name__ == "__main__":
    print("LN fwd pair, bf16 input (attention block)   %6.2f us" % timeit(ln_fwd(0)))
    print("LN fwd pair, bf16 input, no dropout         %6.2f us" % timeit(ln_fwd(0, p=0.0)))
    print("LN fwd pair, bf16 input + e4m3 copy         %6.2f us" % timeit(ln_fwd(0, emit8=True)))
    print("LN fwd pair, 3 fp32 split-K slabs (FFN)     %6.2f us" % timeit(ln_fwd(3)))
    print("LN fwd pair, 2 fp32 split-K slabs           %6.2f us" % timeit(ln_fwd(2)))
    print("LN bwd pair                                 %6.2f us" % timeit(ln_bwd()))
    print("LN bwd pair, no dropout                     %6.2f us" % timeit(ln_bwd(0.0)))
    print("attention fwd pair (20x20 + 36x36)          %6.2f us" % timeit(attn(False)))
    print("attention fwd pair + e4m3 copy              %6.2f us" % timeit(attn(False, emit8=True)))
    print("attention bwd pair                          %6.2f us" % timeit(attn(True)))
