"""What a k-iteration of the LDS-DMA k-loop waits for (instrumented build): k-loop cycles with parts removed --
8 no LDS-DMA loads inside the loop, 32 no MFMAs, 64 fragment reads all from one address (still issued), 128 no barrier,
256 no fragment reads (the MFMAs run on whatever the registers hold); combinations add.
   make -C x-ggm_amd/csrc stamp EXTRA=-DXGGM_KABLATE && python tools/gemm_glds_ablate.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.gemm_stamps import run, _lib  # noqa: E402
from tools.gemm_phase_report import GROUPS  # noqa: E402

if __name__ == "__main__":
    names = sys.argv[1:] or ["FFN1 fwd pair", "QKV fwd pair", "FFN2 bwd group"]
    for name in names:
        for code, tn in ((1, "64x64"), (2, "128x64"), (3, "128x128")):
            for bits in (0, 8, 32, 256, 8 | 32, 8 | 256, 32 | 256, 8 | 32 | 128, 8 | 256 | 128, 8 | 32 | 64, 8 | 32 | 256 | 128):
                _lib.lib.xggm_gemm_set_ablate(bits)
                run(code, GROUPS[name], "%-16s %-7s ablate=%3d" % (name, tn, bits))
    _lib.lib.xggm_gemm_set_ablate(0)
