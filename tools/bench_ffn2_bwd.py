"""Cost of the fused epilogues of the heaviest launch of the step (FFN2 backward, both streams): plain dgrad vs
dgrad * gelu'(u) vs that plus the column sums (bias gradient)."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xggm_amd import ops
from tools.bench_gemm import timeit

dev, BF = "cuda", torch.bfloat16


def main():
    H, I = 768, 3072
    w2 = (torch.randn(H, I, device=dev) * 0.05).to(BF)
    streams = []
    for M in (1152, 640):
        d_h = torch.randn(M, H, device=dev).to(BF)
        act = torch.randn(M, I, device=dev).to(BF)
        u = torch.randn(M, I, device=dev).to(BF)
        gw = torch.zeros(H, I, device=dev)
        cs = torch.zeros(I, device=dev)
        streams.append((d_h, act, u, gw, cs))
    for label, aux, col in (("plain dgrad", False, False), ("* gelu'(u)", True, False), ("* gelu'(u) + colsum", True, True)):
        probs, keep = [], []
        for d_h, act, u, gw, cs in streams:
            pd, dx = ops.p_dgrad(d_h, w2, gelu_aux=u if aux else None, colsum=cs if col else None)
            probs += [ops.p_wgrad(d_h, act, gw, False), pd]
            keep.append(dx)
        t = timeit(lambda: ops.gemm_group(BF, probs)) * 1e6
        print("%-24s %.1f us" % (label, t), flush=True)


def split_experiment():
    """dgrad pair alone + weight-gradient products batched 8 per launch (4 layers' worth) vs the 4-group.
    Needs MAX_GROUP = 8 in gemm.hip / GEMM groups of 8 in ops.gemm_group (measured once: 4-group 42.3 / 38.7 /
    31.2 / 16.0 us vs split 41.0 / 49.0 / 39.0 / 18.4 us for FFN2 / FFN1 / QKV / attn-out: no gain, not adopted)."""
    H, I = 768, 3072
    w2 = (torch.randn(H, I, device=dev) * 0.05).to(BF)
    for name, (no, ni) in (("FFN2 (768 <- 3072)", (H, I)), ("FFN1 (3072 <- 768)", (I, H)), ("QKV (2304 <- 768)", (2304, H)),
                          ("attn-out (768 <- 768)", (H, H))):
        w = (torch.randn(no, ni, device=dev) * 0.05).to(BF)
        dg, wg, keep = [], [], []
        for rep in range(4):
            for M in (1152, 640):
                dy = torch.randn(M, no, device=dev).to(BF)
                x = torch.randn(M, ni, device=dev).to(BF)
                gw = torch.zeros(no, ni, device=dev)
                pd, dx = ops.p_dgrad(dy, w)
                keep += [dy, x, gw, dx]
                if rep == 0:
                    dg.append(pd)
                wg.append(ops.p_wgrad(dy, x, gw, False))
        t4 = timeit(lambda: ops.gemm_group(BF, [wg[0], dg[0], wg[1], dg[1]])) * 1e6
        td = timeit(lambda: ops.gemm_group(BF, dg)) * 1e6
        tw = timeit(lambda: ops.gemm_group(BF, wg)) * 1e6
        print("%-24s 4-group %.1f us | dgrad pair %.1f + 8 wgrads %.1f / 4 = %.1f us" % (name, t4, td, tw, td + tw / 4), flush=True)


if __name__ == "__main__":
    split_experiment()
    main()
