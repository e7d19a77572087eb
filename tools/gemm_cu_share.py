"""Co-resident workgroups on neighbouring tiles (cu_remap in gemm.hip) against the plain XCD order: launch time and
k-loop cycles of the step's grouped launches.   make -C x-ggm_amd/csrc stamp && python tools/gemm_cu_share.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.gemm_stamps import run, _lib  # noqa: E402
from tools.gemm_phase_report import GROUPS  # noqa: E402

if __name__ == "__main__":
    for name, shapes in GROUPS.items():
        for code, tn in ((0, "auto"), (2, "128x64"), (1, "64x64")):
            for flag, what in ((0x200, "xcd order"), (0, "cu order")):
                _lib.lib.xggm_gemm_set_tile(flag)
                run(code, shapes, "%-18s %-7s %-9s" % (name, tn, what))
    _lib.lib.xggm_gemm_set_tile(0)
