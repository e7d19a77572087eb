"""GNN aggregate (adj @ x) against the HBM roofline: the training shape (B = 32, N = 36) and the stress
configuration C4 (B = 64, N = 64), H = 768, bf16 features, fp32 adjacency."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xggm_amd import ops
from tools.bench_gemm import timeit


def main():
    H = 768
    for B, N in ((32, 36), (64, 64), (256, 64)):
        x = torch.randn(B, N, H, device="cuda").bfloat16()
        adj = torch.rand(B, N, N, device="cuda")
        out = torch.empty_like(x)
        for mode, name in ((ops.AGG_PLAIN, "A x"), (ops.AGG_TRANSPOSE, "A^T x")):
            t = timeit(lambda: ops.aggregate(adj, x, mode=mode, out=out), n=20)
            byt = 2 * B * N * H * 2 + B * N * N * 4
            print("B=%3d N=%2d %-6s %6.1f us  %6.0f GB/s algorithmic (%.1f MB), %.1f GFLOP/s" %
                  (B, N, name, t * 1e6, byt / t / 1e9, byt / 1e6, 2.0 * B * N * N * H / t / 1e9), flush=True)


if __name__ == "__main__":
    main()
