"""The grouped launches of one layer (both streams) under every pinned tile (xggm_gemm_set_group_tile: 0 heuristic,
1: 64x64, 2: 128x64, 3: 128x128 on 4 waves, 4: 128x128 on 8 waves); 40 launches per graph replay, hot."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xggm_amd import ops, _lib  # noqa: E402

dev, BF = "cuda", torch.bfloat16
H, I = 768, 3072
MS = (1152, 640)


def timeit(fn, reps=40, iters=4):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return 1000.0 * e0.elapsed_time(e1) / (iters * reps)


def fwd(N, K, gelu=False):
    w = (torch.randn(N, K, device=dev) * 0.05).to(BF)
    b = torch.randn(N, device=dev)
    xs = [torch.randn(M, K, device=dev).to(BF) for M in MS]
    ps = [ops.p_fwd(x, w, b, **(dict(act=ops.ACT_GELU, want_preact=True) if gelu else {})) for x in xs]
    return [p[0] for p in ps], ps


def splitk():
    w = (torch.randn(H, I, device=dev) * 0.05).to(BF)
    xs = [torch.randn(M, I, device=dev).to(BF) for M in MS]
    ps = [ops.p_fwd_splitk(x, w, 3) for x in xs]
    return [p[0] for p in ps], ps


def bwd(N, K, gelu=False):
    """dy [M, N] -> dx = dy W ([M, K]) and dW [N, K] = dy^T x"""
    w = (torch.randn(N, K, device=dev) * 0.05).to(BF)
    probs, keep = [], []
    for M in MS:
        dy = torch.randn(M, N, device=dev).to(BF)
        x = torch.randn(M, K, device=dev).to(BF)
        u = torch.randn(M, K, device=dev).to(BF)
        gw = torch.zeros(N, K, device=dev)
        cs = torch.zeros(K, device=dev)
        pd, dx = ops.p_dgrad(dy, w, gelu_aux=u if gelu else None, colsum=cs if gelu else None, defer=[])
        probs += [ops.p_wgrad(dy, x, gw, False), pd]
        keep += [dy, x, u, gw, cs, dx]
    return probs, keep


def main():
    cases = [("QKV fwd", fwd(2304, H)), ("attn-out fwd", fwd(H, H)), ("FFN1 fwd + GELU", fwd(I, H, True)),
             ("FFN2 fwd split-K 3", splitk()), ("FFN2 bwd (gelu', colsum)", bwd(H, I, True)), ("FFN1 bwd", bwd(I, H)),
             ("QKV bwd", bwd(2304, H)), ("attn-out bwd", bwd(H, H))]
    pins = (1, 2, 3, 4)
    print("%-28s" % "launch" + "".join("%9s" % ("tile %d" % p) for p in pins))
    for name, (probs, keep) in cases:
        row = []
        for p in pins:
            _lib.lib.xggm_gemm_set_group_tile(p)
            row.append(timeit(lambda: ops.gemm_group(BF, probs)))
        _lib.lib.xggm_gemm_set_group_tile(0)
        print("%-28s" % name + "".join("%9.1f" % t for t in row), flush=True)


if __name__ == "__main__":
    main()
