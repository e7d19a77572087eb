"""k-tile time of the single-launch tuned GEMM variants (prefetch depth / tile) at 1-2 tiles per CU."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xggm_amd import ops, _lib
from tools.bench_gemm import timeit

dev = "cuda"
VARS = {"64x64d1": (6, 64, 64), "64x64d2": (1, 64, 64), "64x64d4": (2, 64, 64), "128x64d1": (7, 128, 64),
        "128x64d2": (3, 128, 64), "128x128d1": (8, 128, 128), "128x128d2": (5, 128, 128)}


def main():
    for name, (code, bm, bn) in VARS.items():
        for swz in (0, 0x100):
            _lib.lib.xggm_gemm_set_tile(code | swz)
            for tiles_per_cu in (1, 2):
                M, N = 16 * bm, 16 * bn * tiles_per_cu
                row = []
                for K in (512, 1024, 2048, 4096):
                    x = torch.randn(M, K, device=dev).bfloat16()
                    w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
                    row.append(timeit(lambda: ops.linear_fwd(x, w, None)) * 1e6)
                slope = (row[-1] - row[-2]) / 32
                print("%-10s xcd=%d tiles/CU=%d " % (name, 0 if swz else 1, tiles_per_cu) + " ".join("%8.1f" % r for r in row) +
                      "  us; per k-tile %.3f us; TF %.0f" % (slope, 2.0 * M * N * 4096 / row[-1] / 1e6), flush=True)
    _lib.lib.xggm_gemm_set_tile(0)


if __name__ == "__main__":
    main()
