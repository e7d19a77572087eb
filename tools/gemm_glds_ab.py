"""LDS-DMA k-loop (gemm_kloop_glds) against the register-staged one, per operand-layout form, at the step's shapes
(instrumented build: launch time and k-loop cycles).   make -C x-ggm_amd/csrc stamp && python tools/gemm_glds_ab.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.gemm_stamps import run, _lib  # noqa: E402

FORMS = {
    "fwd   x2 (1152|640 x 3072 x 768)": [("fwd", 1152, 3072, 768), ("fwd", 640, 3072, 768)],
    "dgrad x2 (1152|640 x 3072 x 768)": [("dgrad", 1152, 3072, 768), ("dgrad", 640, 3072, 768)],
    "dgrad x2 (1152|640 x 768 x 3072)": [("dgrad", 1152, 768, 3072), ("dgrad", 640, 768, 3072)],
    "wgrad x2 (3072 x 768 x 1152|640)": [("wgrad", 3072, 768, 1152), ("wgrad", 3072, 768, 640)],
    "wgrad x2 (768 x 3072 x 1152|640)": [("wgrad", 768, 3072, 1152), ("wgrad", 768, 3072, 640)],
}

if __name__ == "__main__":
    for name, shapes in FORMS.items():
        for code, tn in ((2, "128x64"), (1, "64x64"), (3, "128x128")):
            for flag, what in ((0x400, "registers"), (0, "lds-dma")):
                _lib.lib.xggm_gemm_set_tile(flag)
                run(code, shapes, "%-34s %-7s %-9s" % (name, tn, what))
    _lib.lib.xggm_gemm_set_tile(0)
