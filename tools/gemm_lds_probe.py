"""the three operand-layout forms of the tuned GEMM at the step's shapes, a few launches each, for counter runs:
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE ... -- python tools/gemm_lds_probe.py [tile pin]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xggm_amd import ops, _lib  # noqa: E402

BF = torch.bfloat16
pin = int(sys.argv[1]) if len(sys.argv) > 1 else 0
_lib.lib.xggm_gemm_set_tile(pin)
for M, N, K in [(1152, 768, 3072), (1152, 3072, 768), (1152, 2304, 768)]:
    x = torch.randn(M, K, device="cuda").to(BF)
    w = (torch.randn(N, K, device="cuda") * .05).to(BF)
    dy = torch.randn(M, N, device="cuda").to(BF)
    gw = torch.zeros(N, K, device="cuda")
    for _ in range(5):
        ops.linear_fwd(x, w, None)       # A k-major, B k-major
        ops.linear_dgrad(dy, w)          # A k-major, B row-major (transposing LDS reads)
        ops.linear_wgrad(dy, x, gw, False)  # both row-major
torch.cuda.synchronize()
