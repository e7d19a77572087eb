"""ablation: is the tuned GEMM bound by cache bandwidth?  Same launch with row strides = 0, so every
workgroup streams the SAME two panels (all L1/L2 hits) -- results are garbage, only time matters."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xggm_amd import ops, _lib
from bench_gemm import timeit
dev = "cuda"
for (M, N, K) in [(1152, 768, 3072), (1152, 3072, 768), (1152, 768, 768), (640, 768, 768)]:
    x = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for tile in (2, 4, 5):
        _lib.lib.xggm_gemm_set_tile(tile)
        t_norm = timeit(lambda: ops.gemm_raw(torch.bfloat16, x, w, y, M, N, K, K, 1, K, 1, N))
        t_same = timeit(lambda: ops.gemm_raw(torch.bfloat16, x, w, y, M, N, K, 0, 1, 0, 1, N))
        fl = 2.0 * M * N * K
        print("(%d,%d,%d) tile %d: normal %.1f us (%.0f TF)   same-panel %.1f us (%.0f TF)"
              % (M, N, K, tile, t_norm * 1e6, fl / t_norm / 1e12, t_same * 1e6, fl / t_same / 1e12))
_lib.lib.xggm_gemm_set_tile(0)
