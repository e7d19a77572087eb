"""Phase breakdown of the attention core (instrumented build: make -C x-ggm_amd/csrc stamp): mean cycles per phase
of a workgroup.  fwd slots: 0 entry, 1 tiles in LDS, 2 scores, 3 softmax, 4 P (dropout, bf16), 5 P V + staging, 6 stored.
bwd slots: 0 entry, 1 tiles in LDS, 2 scores + dP, 3 softmax, 4 dS / Pd rows, 5 gradient tiles stored."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["XGGM_LIB"] = os.path.join(ROOT, "x-ggm_amd", "csrc", "build_stamp", "libxggm_hip.so")
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from xggm_amd import _lib  # noqa: E402
from tools.bench_rowops import attn, timeit  # noqa: E402

fn = _lib.lib.xggm_attn_set_stamp
fn.argtypes = [ctypes.c_void_p]
for bwd in (False, True):
    run = attn(bwd)
    t = timeit(run)
    buf = torch.zeros(16 * 4096, dtype=torch.int64, device="cuda")
    fn(buf.data_ptr())
    run()
    torch.cuda.synchronize()
    fn(None)
    s = buf.view(-1, 16).cpu().double()
    s = s[s[:, 0] != 0]
    n = 7 if not bwd else 6
    d = [(s[:, i + 1] - s[:, i]).mean().item() for i in range(n - 1)]
    span = (s[:, n - 1].max() - s[:, 0].min()).item()
    print("%s: %.2f us/launch, %d workgroups, first entry -> last exit %.0f cycles; phases (mean cycles): %s; whole workgroup %.0f"
          % ("bwd" if bwd else "fwd", t, s.shape[0], span, " ".join("%.0f" % x for x in d), (s[:, n - 1] - s[:, 0]).mean().item()))
