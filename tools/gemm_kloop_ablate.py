"""What a k-iteration of the grouped GEMM waits for (instrumented build): k-loop cycles with parts removed --
8 no global loads, 16 no LDS stores, 32 no MFMAs, 64 fragment reads all from one address (still issued), 128 no
barrier; combinations add.   make -C x-ggm_amd/csrc stamp && python tools/gemm_kloop_ablate.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.gemm_stamps import run, _lib  # noqa: E402
from tools.gemm_phase_report import GROUPS  # noqa: E402

if __name__ == "__main__":
    names = sys.argv[1:] or ["QKV fwd pair", "FFN2 bwd group"]
    for name in names:
        for code, tn in ((2, "128x64"), (3, "128x128")):
            for bits in (0, 8, 16, 32, 8 | 16, 8 | 16 | 32, 16 | 128, 8 | 16 | 128, 8 | 16 | 32 | 128):
                _lib.lib.xggm_gemm_set_ablate(bits)
                run(code, GROUPS[name], "%-16s %-7s ablate=%3d" % (name, tn, bits))
    _lib.lib.xggm_gemm_set_ablate(0)
