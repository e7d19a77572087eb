"""run a few representative bf16 GEMMs repeatedly (for rocprofv3 --pmc / --kernel-trace)."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xggm_amd import ops
dev = "cuda"
for (M, N, K) in [(1152, 768, 3072), (1152, 3072, 768), (1152, 768, 768)]:
    x = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    dy = torch.randn(M, N, device=dev).bfloat16()
    gw = torch.zeros(N, K, device=dev)
    for _ in range(10):
        ops.linear_fwd(x, w, None)
        ops.linear_dgrad(dy, w)
        ops.linear_wgrad(dy, x, gw, False)
torch.cuda.synchronize()
