"""Do the grouped GEMMs of the step wait for HBM?  The same forward pair (FFN1 shape) launched 96 times inside one
graph, (hot) always with the same weight matrix -- what every other micro-benchmark here does -- and (cold) with 96
different ones (450 MB: beyond L2 and the 256 MB Infinity Cache, as in the step, where a pass streams 442 MB of
weights); (touched) = cold, but a streaming read of the NEXT launch's weights is queued in front of each launch.
Register-staged against LDS-DMA k-loops.      python tools/gemm_cold.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xggm_amd import ops, _lib  # noqa: E402

BF = torch.bfloat16
N_SETS = 96


def graph_time(fns):
    for f in fns[:3]:
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for f in fns:
            f()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * len(fns)) * 1e3


def main():
    dev = "cuda"
    shapes = {"FFN1 fwd pair": (3072, 768), "QKV fwd pair": (2304, 768), "attn-out fwd pair": (768, 768)}
    for name, (N, K) in shapes.items():
        xs = [torch.randn(M, K, device=dev).to(BF) for M in (1152, 640)]
        ws = [(torch.randn(N, K, device=dev) * 0.05).to(BF) for _ in range(N_SETS)]
        keep = []

        def launch(w):
            made = [ops.p_fwd(x, w) for x in xs]
            keep.append(made)
            ps = [m[0] for m in made]
            return lambda: ops.gemm_group(BF, ps)

        hot = [launch(ws[0]) for _ in range(N_SETS)]
        cold = [launch(w) for w in ws]
        sink = torch.zeros((), device=dev)
        touched = []
        for i, w in enumerate(ws):
            nxt = ws[(i + 1) % N_SETS]
            f = cold[i]
            touched.append((lambda f=f, nxt=nxt: (f(), ops.sqnorm_bf16(nxt.view(-1), sink) if hasattr(ops, "sqnorm_bf16") else None)))
        t_touch_only = graph_time([(lambda w=w: ops.sqnorm_bf16(w.view(-1), sink)) for w in ws])
        for flag, what in ((0x400, "registers"), (0, "lds-dma")):
            _lib.lib.xggm_gemm_set_tile(flag)
            print("%-18s %-9s hot %6.1f us   cold %6.1f us   cold + touch of the next weights %6.1f us (the touch alone %4.1f us)" % (
                name, what, graph_time(hot), graph_time(cold), graph_time(touched), t_touch_only), flush=True)
    _lib.lib.xggm_gemm_set_tile(0)


if __name__ == "__main__":
    main()
