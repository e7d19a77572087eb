"""Start-up against steady state of the LDS-DMA k-loop (instrumented build): k-loop cycles of one launch shape at
K = 64 ... 3072 -- the intercept is what the first tiles' loads cost while every CU starts at once, the slope what one
more k-tile costs.   make -C x-ggm_amd/csrc stamp && python tools/gemm_kslope.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.gemm_stamps import run, _lib  # noqa: E402

if __name__ == "__main__":
    for code, tn in ((1, "64x64"), (2, "128x64"), (3, "128x128")):
        for K in (64, 128, 256, 512, 768, 1536, 3072):
            run(code, [("fwd", 1152, 3072, K), ("fwd", 640, 3072, K)], "%-7s FFN1-like fwd pair K=%d" % (tn, K))
    _lib.lib.xggm_gemm_set_group_tile(0)
