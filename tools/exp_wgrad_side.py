"""Weight gradients on a second stream?  The backward's critical path is the chain of input-gradient products (dgrad);
the weight-gradient products (wgrad) only feed the optimiser.  Today one grouped launch carries both.  This times a synthetic
backward of 14 two-stream FFN blocks + 14 attention-projection blocks (dependent launches, a LayerNorm-sized launch between
the products) in the two forms, each captured into one graph: (a) grouped as the step launches them, (b) dgrad on the main
stream, wgrad as a parallel branch joined at the end.   python tools/exp_wgrad_side.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xggm_amd import ops  # noqa: E402
from tools.gemm_ktime import problem  # noqa: E402

BF = torch.bfloat16
ROWS = (1152, 640)


def blocks():
    """per layer: the products of the FFN backward and of the attention projections' backward (both modality streams)"""
    out = []
    for _ in range(14):
        layer = []
        for (n_out, n_in) in ((768, 3072), (3072, 768), (768, 768), (2304, 768)):  # FFN2, FFN1, attention output, QKV
            d = [problem("dgrad", M, n_in, n_out) for M in ROWS]   # dx[M, n_in] = dy[M, n_out] w[n_out, n_in]
            w = [problem("wgrad", n_out, n_in, M) for M in ROWS]   # gw[n_out, n_in] = dy[M, n_out]^T x[M, n_in]
            layer.append((d, w))
        out.append(layer)
    return out


def main():
    torch.cuda.set_device(0)
    B = blocks()
    small = torch.zeros(1792 * 768, device="cuda")
    side = torch.cuda.Stream()

    def filler():  # a latency-bound row kernel between the products (LayerNorm / attention core stand-in)
        ops.zero_ranges(small, [(0, small.numel())])

    def grouped():
        for layer in B:
            for d, w in layer:
                ops.gemm_group(BF, [w[0][0], d[0][0], w[1][0], d[1][0]])
                filler()

    def split():
        main_s = torch.cuda.current_stream()
        for layer in B:
            for d, w in layer:
                ev = torch.cuda.Event()
                ev.record(main_s)          # the incoming gradient exists
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    ops.gemm_group(BF, [w[0][0], w[1][0]])
                ops.gemm_group(BF, [d[0][0], d[1][0]])
                filler()
        main_s.wait_stream(side)

    def dgrad_only():
        for layer in B:
            for d, w in layer:
                ops.gemm_group(BF, [d[0][0], d[1][0]])
                filler()

    def time_graph(fn, n=20):
        cap = torch.cuda.Stream()
        cap.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cap):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=cap):
                fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            g.replay()
        e0.record()
        for _ in range(n):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    print("grouped (wgrad + dgrad in one launch)      %.3f ms" % time_graph(grouped), flush=True)
    print("dgrad chain on main, wgrad on a branch     %.3f ms" % time_graph(split), flush=True)
    print("dgrad chain alone (lower bound)            %.3f ms" % time_graph(dgrad_only), flush=True)


if __name__ == "__main__":
    main()
