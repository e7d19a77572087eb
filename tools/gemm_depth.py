"""LDS stages of the LDS-DMA k-loop against co-residency (instrumented build): the forward pairs on the 128 x 128 and
128 x 64 tiles with 2, 3 and 4 stages.   make -C x-ggm_amd/csrc stamp && python tools/gemm_depth.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.gemm_stamps import run, _lib  # noqa: E402
from tools.gemm_phase_report import GROUPS  # noqa: E402

if __name__ == "__main__":
    for name in sys.argv[1:] or ["QKV fwd pair", "FFN1 fwd pair", "FFN2 bwd group"]:
        for code, tn in ((2, "128x64"), (3, "128x128")):
            for bits, ns in ((0x800, 2), (0x1000, 3), (0x2000, 4)):
                _lib.lib.xggm_gemm_set_tile(bits)
                run(code, GROUPS[name], "%-16s %-7s stages %d" % (name, tn, ns))
    _lib.lib.xggm_gemm_set_tile(0)
    _lib.lib.xggm_gemm_set_group_tile(0)
