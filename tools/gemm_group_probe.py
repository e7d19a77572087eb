"""A few launches of every grouped GEMM of the step (tools/gemm_phase_report.GROUPS), for counter runs:
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d OUT -- python tools/gemm_group_probe.py [flags] [tile]
flags: value for xggm_gemm_set_tile (0x400 = register-staged k-loop everywhere); tile: grouped tile code (0 = auto)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xggm_amd import ops, _lib  # noqa: E402
from tools.gemm_ktime import problem  # noqa: E402
from tools.gemm_phase_report import GROUPS  # noqa: E402

flags = int(sys.argv[1], 0) if len(sys.argv) > 1 else 0
tile = int(sys.argv[2]) if len(sys.argv) > 2 else 0
only = sys.argv[3] if len(sys.argv) > 3 else None
_lib.lib.xggm_gemm_set_tile(flags)
_lib.lib.xggm_gemm_set_group_tile(tile)
for name, shapes in GROUPS.items():
    if only and only not in name:
        continue
    made = [problem(f, M, N, K) for f, M, N, K in shapes]
    ps = [m[0] for m in made]
    for _ in range(4):
        ops.gemm_group(torch.bfloat16, ps)
torch.cuda.synchronize()
