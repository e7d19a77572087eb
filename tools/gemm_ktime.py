"""Tile-time model of the grouped bf16 GEMM: launch time vs K at 1, 2 and 4 tiles per CU, per form.
Two equal problems per launch (a single problem takes the one-GEMM path)."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xggm_amd import ops, _lib
from tools.bench_gemm import timeit

dev = "cuda"
BF = torch.bfloat16


def problem(form, M, N, K):
    """(problem, tensors to keep alive: the problem holds raw pointers)"""
    if form == "fwd":
        x = torch.randn(M, K, device=dev).bfloat16()
        w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        p, y, _ = ops.p_fwd(x, w)
        return p, (x, w, y)
    if form == "dgrad":  # dx[M, N] = dy[M, K] w[K, N]
        dy = torch.randn(M, K, device=dev).bfloat16()
        wt = (torch.randn(K, N, device=dev) * 0.05).bfloat16()
        p, dx = ops.p_dgrad(dy, wt)
        return p, (dy, wt, dx)
    dy = torch.randn(K, M, device=dev).bfloat16()  # gw[M, N] = dy[K, M]^T x[K, N]
    xx = torch.randn(K, N, device=dev).bfloat16()
    gw = torch.zeros(M, N, device=dev)
    return ops.p_wgrad(dy, xx, gw, False), (dy, xx, gw)


def main():
    for tile, code, tm in (("128x128", 3, 128), ("128x64", 2, 128), ("64x64", 1, 64)):
        _lib.lib.xggm_gemm_set_group_tile(code)
        tn = 64 if tile == "128x64" else tm
        for form in ("fwd", "dgrad", "wgrad"):
            for tiles_per_cu in (1, 2, 4):
                M, N = 16 * tm, 8 * tn * tiles_per_cu
                row = []
                for K in (64, 512, 1024, 2048, 4096):
                    made = [problem(form, M, N, K) for _ in range(2)]
                    ps = [m[0] for m in made]
                    row.append(timeit(lambda: ops.gemm_group(BF, ps)) * 1e6)
                slope = (row[-1] - row[-2]) / 32
                print("%-8s %-6s tiles/CU=%d  " % (tile, form, tiles_per_cu) + " ".join("%8.1f" % r for r in row) +
                      "   us;  per 64-k-tile %.3f us, TF at K=4096: %.0f" % (slope, 4.0 * M * N * 4096 / row[-1] / 1e6),
                      flush=True)
    _lib.lib.xggm_gemm_set_group_tile(0)


if __name__ == "__main__":
    main()
