"""BertAdam on an arena-sized span, fp32 against bf16 gradients (the data-parallel wire arena's form): does the update
gain more from the narrower gradient stream than its bytes (30 -> 28 B per parameter)?  rocprofv3 of the one-rank
data-parallel step showed 908 us against 1153 us per launch on the same box (profiles/r04_experiments/dp_one_rank.txt).
    python tools/bench_adam_streams.py [n_parameters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xggm_amd import ops  # noqa: E402


def timeit(fn, n=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))
    return ts[len(ts) // 2] * 1e-3


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 190_000_000
    n -= n % 256
    p, m, v = (torch.randn(n, device="cuda") * 0.01 for _ in range(3))
    v.abs_()
    g32 = torch.randn(n, device="cuda") * 0.01
    g16 = g32.to(torch.bfloat16)
    sh = torch.empty(n, device="cuda", dtype=torch.bfloat16)
    sqn = torch.ones(1, device="cuda")
    sc = torch.ones(1, device="cuda")
    lr = torch.full((1,), 1e-5, device="cuda")
    for name, g, bytes_ in (("fp32 gradients", g32, 30.0), ("bf16 gradients", g16, 28.0)):
        t = timeit(lambda: ops.bertadam_multi([((p, g, m, v, sh, sqn, 5.0, 1e-5, sc, 0.9, 0.999, 1e-6, 0.01), dict(lr_dev=lr))]))
        print("%s: %.1f us, %.2f TB/s (%d B per parameter)" % (name, t * 1e6, bytes_ * n / t / 1e12, bytes_), flush=True)
    # the same with the gradient stream alone / without it: what each stream costs
    t = timeit(lambda: ops.sqnorm_multi(g32, [(0, n)], sqn))
    print("read of the fp32 gradient alone (sqnorm): %.1f us, %.2f TB/s" % (t * 1e6, 4.0 * n / t / 1e12))


if __name__ == "__main__":
    main()
