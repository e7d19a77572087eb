"""Tiles alone on their CU against tiles that share it (instrumented build), for the launches of the step and the tiles
the grouped kernel has.   make -C x-ggm_amd/csrc stamp && python tools/gemm_share.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.gemm_stamps import run, _lib  # noqa: E402
from tools.gemm_phase_report import GROUPS  # noqa: E402

if __name__ == "__main__":
    for name in sys.argv[1:] or ["FFN1 fwd pair", "QKV fwd pair", "attn-out fwd pair", "FFN2 bwd group", "FFN1 bwd group"]:
        for code, tn in ((1, "64x64"), (2, "128x64"), (3, "128x128")):
            run(code, GROUPS[name], "%-18s %-7s" % (name, tn))
    _lib.lib.xggm_gemm_set_group_tile(0)
