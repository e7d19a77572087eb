"""fp8 (e4m3) against bf16 forward products at the shapes of the training step (B = 32) and of the validation
sweep (B = 512), same tile, same pipeline, single-problem launches, n calls inside one hipGraph
(BASELINE config C5: is halving the operand bytes worth a quantisation pass at these sizes?).  Also times the
quantiser itself (the pass an activation needs when its producer does not emit e4m3 directly)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xggm_amd import ops, _lib  # noqa: E402
from tools.bench_gemm import timeit  # noqa: E402

SHAPES = [(1152, 2304, 768), (1152, 768, 768), (1152, 3072, 768), (1152, 768, 3072), (640, 2304, 768),
          (640, 3072, 768), (640, 768, 3072), (18432, 2304, 768), (18432, 3072, 768), (18432, 768, 3072)]
TILES = {"64x64": (2, 0), "128x64": (3, 3), "128x128": (5, 5)}  # name -> (bf16 pin, fp8 pin)


def main():
    dev = "cuda"
    print("%-22s %-8s %10s %10s %8s %12s" % ("M,N,K", "tile", "bf16 us", "fp8 us", "ratio", "quant(x) us"))
    for M, N, K in SHAPES:
        x = torch.randn(M, K, device=dev).to(torch.bfloat16)
        w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
        b = torch.zeros(N, device=dev)
        one = torch.ones(1, device=dev)
        x8, w8 = ops.quantize_fp8(x, one), ops.quantize_fp8(w, one)
        tq = timeit(lambda: ops.quantize_fp8(x, one, None))
        for name, (pb, pf) in TILES.items():
            _lib.lib.xggm_gemm_set_tile(pb)
            t16 = timeit(lambda: ops.linear_fwd(x, w, b, act=ops.ACT_GELU))
            _lib.lib.xggm_gemm_set_tile(pf)
            t8 = timeit(lambda: ops.linear_fwd_fp8(x8, w8, one, one, bias=b, act=ops.ACT_GELU))
            _lib.lib.xggm_gemm_set_tile(0)
            print("%-22s %-8s %10.1f %10.1f %8.2f %12.1f   (%.0f / %.0f TFLOP/s)" % (
                "%d,%d,%d" % (M, N, K), name, t16 * 1e6, t8 * 1e6, t16 / t8, tq * 1e6, 2e-12 * M * N * K / t16,
                2e-12 * M * N * K / t8), flush=True)


if __name__ == "__main__":
    main()
