"""What bounds the heaviest launch of the step (FFN2 backward group: two weight-gradient + two dgrad products)?
Variants on HOT operands (the same buffers every launch: they live in L2 / the Infinity Cache) and on COLD ones (24
sets of buffers cycled: 1.7 GB, nothing survives between launches -- the training step's situation): fp32 vs bf16
weight-gradient output, dgrad pair alone, weight-gradient pair alone."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xggm_amd import ops  # noqa: E402

dev, BF = "cuda", torch.bfloat16
H, I = 768, 3072


def make_set(gw_dtype):
    w2 = (torch.randn(H, I, device=dev) * 0.05).to(BF)
    out = []
    for M in (1152, 640):
        d_h = torch.randn(M, H, device=dev).to(BF)
        act = torch.randn(M, I, device=dev).to(BF)
        u = torch.randn(M, I, device=dev).to(BF)
        gw = torch.zeros(H, I, device=dev, dtype=gw_dtype)
        cs = torch.zeros(I, device=dev)
        out.append((d_h, act, u, gw, cs, w2))
    return out


def problems(s, what):
    probs, keep = [], []
    for d_h, act, u, gw, cs, w2 in s:
        if what in ("all", "dgrad"):
            pd, dx = ops.p_dgrad(d_h, w2, gelu_aux=u, colsum=cs, defer=[])
            probs.append(pd)
            keep.append(dx)
        if what in ("all", "wgrad"):
            probs.append(ops.p_wgrad(d_h, act, gw, False))
    return probs, keep


def time_sets(sets, what, iters=20):
    pk = [problems(s, what) for s in sets]
    for p, _ in pk:
        ops.gemm_group(BF, p)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for p, _ in pk:
            ops.gemm_group(BF, p)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return 1000.0 * e0.elapsed_time(e1) / (iters * len(sets))


def main():
    for gdt in (torch.float32, BF):
        hot = [make_set(gdt)]
        cold = [make_set(gdt) for _ in range(24)]
        for what in ("all", "dgrad", "wgrad"):
            th, tc = time_sets(hot, what), time_sets(cold, what)
            print("gw %-8s %-6s hot %.1f us   cold %.1f us" % (str(gdt).split(".")[-1], what, th, tc), flush=True)
        del hot, cold
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
