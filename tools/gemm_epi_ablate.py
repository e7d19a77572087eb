"""Where the register epilogue's cycles go (instrumented build): stamps and launch time with the bf16 stores of the
epilogue removed (ablate bit 4).   make -C x-ggm_amd/csrc stamp && python tools/gemm_epi_ablate.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.gemm_stamps import run, _lib  # noqa: E402
from tools.gemm_phase_report import GROUPS  # noqa: E402

if __name__ == "__main__":
    import torch
    from tools.bench_gemm import timeit
    from xggm_amd import ops
    one = torch.ones((), device="cuda")
    t = timeit(lambda: ops.add_scalars([one, one]), n=50) * 1e6
    print("floor of a dependent launch inside a graph (one-thread kernel): %.2f us" % t, flush=True)
    big = torch.zeros(1 << 22, device="cuda")
    t = timeit(lambda: ops.zero_ranges(big, [(0, 1 << 22)]), n=50) * 1e6
    print("16 MB fill: %.2f us" % t, flush=True)
    for name in ("QKV fwd pair", "FFN1 fwd pair", "attn-out fwd pair"):
        for bits in (0, 4, 2):
            _lib.lib.xggm_gemm_set_ablate(bits)
            run(2, GROUPS[name], "%-18s 128x64 ablate=%d" % (name, bits))
    _lib.lib.xggm_gemm_set_ablate(0)
