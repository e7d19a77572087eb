import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xggm_amd import ops, _lib
dev = "cuda"
M, N, K = 1152, 3072, 768
x = torch.randn(M, K, device=dev).bfloat16()
w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
for tile in (6, 7, 8):
    _lib.lib.xggm_gemm_set_tile(tile)
    for _ in range(5):
        ops.linear_fwd(x, w, None)
torch.cuda.synchronize()
