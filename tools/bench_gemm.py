"""Micro-benchmark of xggm_gemm_bf16 on the shapes of the training step (B = 32):
forward / dgrad / wgrad forms, tuned vs generic kernel, interleaved in one process."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xggm_amd import ops, _lib

dev = "cuda"
shapes = [  # (M tokens, N out, K in)
    (1152, 2304, 768), (1152, 768, 768), (1152, 3072, 768), (1152, 768, 3072), (1152, 768, 2048),
    (640, 2304, 768), (640, 768, 768), (640, 3072, 768), (640, 768, 3072), (32, 2274, 1536)]


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


tot = {0: 0.0, 1: 0.0}
flops_tot = 0.0
print("%-28s %10s %10s %10s" % ("shape (M,N,K) form", "tuned TF", "generic TF", "us tuned"))
for M, N, K in shapes:
    x = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    dy = torch.randn(M, N, device=dev).bfloat16()
    gw = torch.zeros(N, K, device=dev)
    forms = {"fwd": lambda: ops.linear_fwd(x, w, None), "dgrad": lambda: ops.linear_dgrad(dy, w),
             "wgrad": lambda: ops.linear_wgrad(dy, x, gw, False)}
    for name, fn in forms.items():
        res = {}
        for generic in (0, 1):
            _lib.lib.xggm_gemm_set_generic(generic)
            res[generic] = timeit(fn)
        _lib.lib.xggm_gemm_set_generic(0)
        fl = 2.0 * M * N * K
        weight = 1
        tot[0] += res[0] * weight
        tot[1] += res[1] * weight
        flops_tot += fl
        print("%-28s %10.1f %10.1f %10.1f" % ("(%d,%d,%d) %s" % (M, N, K, name), fl / res[0] / 1e12, fl / res[1] / 1e12,
                                              res[0] * 1e6))
print("aggregate: tuned %.1f TF, generic %.1f TF" % (flops_tot / tot[0] / 1e12, flops_tot / tot[1] / 1e12))
