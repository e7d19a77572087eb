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


def timeit(fn, n=20):
    """GPU time per call: n calls captured into one hipGraph (no host launch latency inside)."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * n) * 1e-3


def main():
    VARS = ["64x64d2", "64x64d4", "64x64d1", "128x64d1", "128x128d1", "auto"]
    CODE = {"64x64d2": 1, "64x64d4": 2, "64x64d1": 6, "128x64d1": 7, "128x128d1": 8, "auto": 0}
    tot = {v: 0.0 for v in VARS}
    flops_tot = 0.0
    print("%-26s " % "shape (M,N,K) form" + " ".join("%9s" % v for v in VARS) + "   (TFLOP/s)")
    for M, N, K in shapes:
        x = torch.randn(M, K, device=dev).bfloat16()
        w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        dy = torch.randn(M, N, device=dev).bfloat16()
        gw = torch.zeros(N, K, device=dev)
        forms = {"fwd": lambda: ops.linear_fwd(x, w, None), "dgrad": lambda: ops.linear_dgrad(dy, w),
                 "wgrad": lambda: ops.linear_wgrad(dy, x, gw, False)}
        for name, fn in forms.items():
            fl = 2.0 * M * N * K
            row = []
            for vi, v in enumerate(VARS):
                _lib.lib.xggm_gemm_set_tile(CODE[v])
                t = timeit(fn)
                tot[v] += t
                row.append(fl / t / 1e12)
            flops_tot += fl
            print("%-26s " % ("(%d,%d,%d) %s" % (M, N, K, name)) + " ".join("%9.1f" % r for r in row))
    _lib.lib.xggm_gemm_set_generic(0)
    _lib.lib.xggm_gemm_set_tile(0)
    print("%-26s " % "aggregate" + " ".join("%9.1f" % (flops_tot / tot[v] / 1e12) for v in VARS))

    # ---- grouped launches vs separate launches ------------------------------------------------------
    print("\ngroup                                   separate(us)  g64x64  g128x64  g128x128   auto")


    def mk(M, N, K):
        x = torch.randn(M, K, device=dev).bfloat16()
        w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        dy = torch.randn(M, N, device=dev).bfloat16()
        gw = torch.zeros(N, K, device=dev)
        return x, w, dy, gw


    # the problems hold raw pointers: the outputs p_fwd / p_dgrad allocate must outlive the launches
    KEEP = []
    _fwd, _dgrad = ops.p_fwd, ops.p_dgrad

    def keep_fwd(*a, **k):
        r = _fwd(*a, **k)
        KEEP.append(r[1:])
        return r

    def keep_dgrad(*a, **k):
        r = _dgrad(*a, **k)
        KEEP.append(r[1:])
        return r

    ops.p_fwd, ops.p_dgrad = keep_fwd, keep_dgrad
    groups = {
        "FFN2 bwd visn: wgrad+dgrad (768x3072)": lambda t: [ops.p_wgrad(t[0][2], t[0][0], t[0][3], False), ops.p_dgrad(t[0][2], t[0][1])[0]],
        "FFN1 bwd visn: wgrad+dgrad (3072x768)": lambda t: [ops.p_wgrad(t[1][2], t[1][0], t[1][3], False), ops.p_dgrad(t[1][2], t[1][1])[0]],
        "attn-out bwd visn (768x768)": lambda t: [ops.p_wgrad(t[2][2], t[2][0], t[2][3], False), ops.p_dgrad(t[2][2], t[2][1])[0]],
        "QKV fwd lang+visn": lambda t: [ops.p_fwd(t[3][0], t[3][1])[0], ops.p_fwd(t[4][0], t[4][1])[0]],
        "FFN1 fwd lang+visn": lambda t: [ops.p_fwd(t[5][0], t[5][1])[0], ops.p_fwd(t[1][0], t[1][1])[0]],
        "FFN2 bwd lang+visn: 2x(wgrad+dgrad)": lambda t: [ops.p_wgrad(t[0][2], t[0][0], t[0][3], False), ops.p_dgrad(t[0][2], t[0][1])[0],
                                                          ops.p_wgrad(t[6][2], t[6][0], t[6][3], False), ops.p_dgrad(t[6][2], t[6][1])[0]],
        "attn-out bwd lang+visn (4 gemms)": lambda t: [ops.p_wgrad(t[2][2], t[2][0], t[2][3], False), ops.p_dgrad(t[2][2], t[2][1])[0],
                                                       ops.p_wgrad(t[7][2], t[7][0], t[7][3], False), ops.p_dgrad(t[7][2], t[7][1])[0]],
    }
    T = [mk(1152, 768, 3072), mk(1152, 3072, 768), mk(1152, 768, 768), mk(640, 2304, 768), mk(1152, 2304, 768),
         mk(640, 3072, 768), mk(640, 768, 3072), mk(640, 768, 768)]
    BF = torch.bfloat16
    for name, build in groups.items():
        probs = build(T)
        def sep():
            for p in probs:
                ops.gemm_group(BF, [p])
        row = [timeit(sep) * 1e6]
        for v in (1, 2, 3, 0):
            _lib.lib.xggm_gemm_set_group_tile(v)
            row.append(timeit(lambda: ops.gemm_group(BF, probs)) * 1e6)
        _lib.lib.xggm_gemm_set_group_tile(0)
        print("%-40s %10.1f %8.1f %8.1f %8.1f %8.1f" % (name, *row))



if __name__ == "__main__":
    main()
