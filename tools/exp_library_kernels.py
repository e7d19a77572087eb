"""which kernels (macro tile, MFMA shape, ...) the vendor library picks for the step's forward shapes: run under
rocprofv3 --kernel-trace --stats and read the kernel names"""
import torch
dev, BF = "cuda", torch.bfloat16
M = 1792
for N, K in ((2304, 768), (768, 768), (3072, 768), (768, 3072)):
    x = torch.randn(M, K, device=dev).to(BF)
    w = (torch.randn(N, K, device=dev) * 0.05).to(BF)
    y = torch.empty(M, N, device=dev, dtype=BF)
    for _ in range(20):
        torch.matmul(x, w.t(), out=y)
torch.cuda.synchronize()
