"""Phase report (in-kernel cycle stamps, instrumented build) of the grouped GEMM launches the training step issues
most: where a launch spends its time -- prologue, k-loop, epilogue -- and how long the launch is against the phases of
one tile (launch overhead, waves of tiles, tail).   make -C x-ggm_amd/csrc stamp && python tools/gemm_phase_report.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.gemm_stamps import run  # noqa: E402  (selects the stamped library)

GROUPS = {
    "QKV fwd pair": [("fwd", 1152, 2304, 768), ("fwd", 640, 2304, 768)],
    "attn-out fwd pair": [("fwd", 1152, 768, 768), ("fwd", 640, 768, 768)],
    "FFN1 fwd pair": [("fwd", 1152, 3072, 768), ("fwd", 640, 3072, 768)],
    "FFN2 bwd group": [("wgrad", 3072, 768, 1152), ("dgrad", 1152, 3072, 768), ("wgrad", 3072, 768, 640), ("dgrad", 640, 3072, 768)],
    "FFN1 bwd group": [("wgrad", 768, 3072, 1152), ("dgrad", 1152, 768, 3072), ("wgrad", 768, 3072, 640), ("dgrad", 640, 768, 3072)],
    "QKV bwd group": [("wgrad", 768, 2304, 1152), ("dgrad", 1152, 768, 2304), ("wgrad", 768, 2304, 640), ("dgrad", 640, 768, 2304)],
    "attn-out bwd group": [("wgrad", 768, 768, 1152), ("dgrad", 1152, 768, 768), ("wgrad", 768, 768, 640), ("dgrad", 640, 768, 768)],
}

if __name__ == "__main__":
    for name, shapes in GROUPS.items():
        for code, tn in ((0, "auto"), (1, "64x64"), (2, "128x64"), (3, "128x128"), (4, "128x128/8w")):
            run(code, shapes, "%-18s %-7s" % (name, tn))
