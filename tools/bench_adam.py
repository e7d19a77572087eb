"""BertAdam step bandwidth on an arena-sized vector (30 B per parameter: p, g, m, v read; p, m, v, bf16 shadow
written).  Variant / grid cap are read from XGGM_ADAM_VARIANT / XGGM_ADAM_GRID at first call."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xggm_amd import ops
from tools.bench_gemm import timeit


def main():
    n = 110_000_000
    p, g, m, v = (torch.randn(n, device="cuda") * 0.01 for _ in range(4))
    v.abs_()
    sh = torch.empty(n, device="cuda", dtype=torch.bfloat16)
    sqn = torch.ones(1, device="cuda")
    t = timeit(lambda: ops.bertadam(p, g, m, v, sh, sqn, 5.0, 1e-5, None, 0.9, 0.999, 1e-6, 0.01), n=5)
    print("variant %s grid %s: %.1f us, %.0f GB/s" % (os.environ.get("XGGM_ADAM_VARIANT", "0"), os.environ.get("XGGM_ADAM_GRID", "4096"),
                                                  t * 1e6, 30.0 * n / t / 1e9), flush=True)
    t = timeit(lambda: ops.sqnorm(g, sqn) if hasattr(ops, "sqnorm") else None, n=5)
    print("sqnorm: %.1f us, %.0f GB/s" % (t * 1e6, 4.0 * n / t / 1e9))


if __name__ == "__main__":
    main()
