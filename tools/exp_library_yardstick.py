"""A yardstick, not a product path: what the vendor library (torch.matmul -> hipBLASLt / rocBLAS) needs for the dense
products of one layer at the step's shapes (both streams as ONE problem of 1792 rows -- the best case for a library
call; the package launches the two streams as two problems of one grouped grid), next to the package's own launches.
Timed as 40 back-to-back launches inside one hipGraph (no launch gaps), hot operands."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xggm_amd import ops  # noqa: E402

dev, BF = "cuda", torch.bfloat16


def timeit(fn, reps=40, iters=5):
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


def main():
    H, I = 768, 3072
    Ml, Mv = 640, 1152
    M = Ml + Mv
    for name, N, K in (("QKV fwd", 2304, H), ("attn-out fwd", H, H), ("FFN1 fwd", I, H), ("FFN2 fwd", H, I)):
        x = torch.randn(M, K, device=dev).to(BF)
        w = (torch.randn(N, K, device=dev) * 0.05).to(BF)
        y = torch.empty(M, N, device=dev, dtype=BF)
        t_lib = timeit(lambda: torch.matmul(x, w.t(), out=y))
        xs = [x[:Mv], x[Mv:]]
        ps = [ops.p_fwd(xx, w, None) for xx in xs]
        t_own = timeit(lambda: ops.gemm_group(BF, [p[0] for p in ps]))
        fl = 2.0 * M * N * K
        print("%-14s M=%d N=%d K=%d: library %.1f us (%.0f TF)   package pair %.1f us (%.0f TF)"
              % (name, M, N, K, t_lib, fl / t_lib / 1e6, t_own, fl / t_own / 1e6), flush=True)
    # backward of FFN2: dgrad d_act = d_h W2 ([M,768] x [768,3072]) and wgrad dW2 = d_h^T act
    d_h = torch.randn(M, H, device=dev).to(BF)
    act = torch.randn(M, I, device=dev).to(BF)
    w2 = (torch.randn(H, I, device=dev) * 0.05).to(BF)
    dx = torch.empty(M, I, device=dev, dtype=BF)
    gw = torch.empty(H, I, device=dev, dtype=BF)
    t1 = timeit(lambda: torch.matmul(d_h, w2, out=dx))
    t2 = timeit(lambda: torch.matmul(d_h.t(), act, out=gw))
    gws = [torch.zeros(H, I, device=dev) for _ in range(2)]
    probs = []
    for sl, g in ((slice(0, Mv), gws[0]), (slice(Mv, M), gws[1])):
        pd, _ = ops.p_dgrad(d_h[sl], w2)
        probs += [ops.p_wgrad(d_h[sl], act[sl], g, False), pd]
    t_own = timeit(lambda: ops.gemm_group(BF, probs))
    fl = 4.0 * M * H * I
    print("FFN2 bwd (plain dgrad + wgrad): library %.1f + %.1f = %.1f us (%.0f TF)   package 4-group %.1f us (%.0f TF)"
          % (t1, t2, t1 + t2, fl / (t1 + t2) / 1e6, t_own, fl / t_own / 1e6), flush=True)


if __name__ == "__main__":
    main()
