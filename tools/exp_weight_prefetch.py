"""Would weights that are already in the Infinity Cache make the step's GEMMs faster?  tools/gemm_cold.py says hot operands
(the same matrix re-read: L2 and Infinity Cache hits) save 0.9 ... 2.6 us per launch against 96 different matrices from
HBM, and that a streaming pre-read in front of the launch buys nothing -- but that pre-read used NON-TEMPORAL loads, which
may not allocate anywhere.  Here the next launch's weights are read with plain loads (a torch reduction) in front of every
launch; the GEMM's own time is the pair's time minus the touch's.     python tools/exp_weight_prefetch.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xggm_amd import ops  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tools"))
from gemm_cold import graph_time, N_SETS, BF  # noqa: E402


def main():
    dev = "cuda"
    shapes = {"FFN1 fwd pair": (3072, 768), "QKV fwd pair": (2304, 768), "attn-out fwd pair": (768, 768)}
    for name, (N, K) in shapes.items():
        xs = [torch.randn(M, K, device=dev).to(BF) for M in (1152, 640)]
        ws = [(torch.randn(N, K, device=dev) * 0.05).to(BF) for _ in range(N_SETS)]
        keep = []
        outs = [torch.zeros((), device=dev, dtype=torch.int64) for _ in range(N_SETS)]

        def launch(w):
            made = [ops.p_fwd(x, w) for x in xs]
            keep.append(made)
            ps = [m[0] for m in made]
            return lambda: ops.gemm_group(BF, ps)

        def touch(w, o):  # a plain-load read of the whole matrix
            return lambda: torch.sum(w.view(torch.int16).view(-1), (0,), dtype=torch.int64, out=o)

        hot = [launch(ws[0]) for _ in range(N_SETS)]
        cold = [launch(w) for w in ws]
        junk = [(torch.randn(N, K, device=dev) * 0.05).to(BF) for _ in range(N_SETS)]
        useful, useless = [], []
        for i in range(N_SETS):
            f = cold[i]
            t = touch(ws[(i + 1) % N_SETS], outs[i])
            useful.append(lambda t=t, f=f: (f(), t()))   # GEMM i, then a plain-load read of launch i + 1's weights
            t2 = touch(junk[i], outs[i])
            useless.append(lambda t=t2, f=f: (f(), t()))  # GEMM i, then the same read of a matrix nobody uses
        t_hot, t_cold = graph_time(hot), graph_time(cold)
        t_a, t_b = graph_time(useful), graph_time(useless)
        print("%-18s hot %6.1f us   cold %6.1f us   GEMM + read of the NEXT launch's weights %6.1f us   GEMM + read of an unrelated "
              "matrix %6.1f us   -> weights already in the Infinity Cache are worth %5.1f us per launch" % (name, t_hot, t_cold, t_a, t_b, t_b - t_a),
              flush=True)


if __name__ == "__main__":
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    main()
