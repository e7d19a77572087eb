"""Cost of the epilogue work of the N = 3072 launches (both streams, graph-replayed, hot operands):
FFN1 forward pair with / without the GELU epilogue (+ pre-activation store), FFN2 dgrad pair plain / * gelu'(u) /
+ column sums."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xggm_amd import ops  # noqa: E402

dev, BF = "cuda", torch.bfloat16
H, I = 768, 3072


def timeit(fn, iters=30):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return 1000.0 * e0.elapsed_time(e1) / iters


def main():
    w1 = (torch.randn(I, H, device=dev) * 0.05).to(BF)
    w2 = (torch.randn(H, I, device=dev) * 0.05).to(BF)
    b1 = torch.randn(I, device=dev)
    xs = [torch.randn(M, H, device=dev).to(BF) for M in (1152, 640)]
    us = [torch.randn(M, I, device=dev).to(BF) for M in (1152, 640)]
    cs = [torch.zeros(I, device=dev) for _ in range(2)]
    for label, kw in (("FFN1 fwd pair, bias only", dict()),
                      ("FFN1 fwd pair, GELU + preact", dict(act=ops.ACT_GELU, want_preact=True))):
        ps = [ops.p_fwd(x, w1, b1, **kw) for x in xs]
        print("%-40s %.1f us" % (label, timeit(lambda: ops.gemm_group(BF, [p[0] for p in ps]))), flush=True)
    for label, aux, col in (("FFN2 dgrad pair, plain", False, False), ("FFN2 dgrad pair, * gelu'(u)", True, False),
                            ("FFN2 dgrad pair, * gelu'(u) + colsum", True, True)):
        ps = [ops.p_dgrad(x, w2, gelu_aux=u if aux else None, colsum=c if col else None, defer=[]) for x, u, c in zip(xs, us, cs)]
        print("%-40s %.1f us" % (label, timeit(lambda: ops.gemm_group(BF, [p[0] for p in ps]))), flush=True)


if __name__ == "__main__":
    main()
