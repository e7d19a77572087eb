"""Which workgroups share a CU?  (instrumented build)  Prints, for one grouped launch, the block indices that ran on
each CU of XCD 0 and their entry times: the basis of the tile order that lets co-resident workgroups share an operand
panel through the CU's L1.   make -C x-ggm_amd/csrc stamp && python tools/gemm_wg_map.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.gemm_stamps import _lib, ops, problem, BF, torch  # noqa: E402
from tools.gemm_phase_report import GROUPS  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "QKV fwd pair"
    code = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    _lib.lib.xggm_gemm_set_group_tile(code)
    made = [problem(f, M, N, K) for f, M, N, K in GROUPS[name]]
    ps = [m[0] for m in made]
    for _ in range(3):
        ops.gemm_group(BF, ps)
    buf = torch.zeros(8 * 8192, dtype=torch.int64, device="cuda")
    fn = _lib.lib.xggm_gemm_set_stamp
    fn.argtypes = [ctypes.c_void_p]
    fn(buf.data_ptr())
    ops.gemm_group(BF, ps)
    torch.cuda.synchronize()
    fn(None)
    s = buf.view(-1, 8).cpu()
    n = int((s[:, 0] != 0).sum())
    s = s[:n]
    hw, xcc = s[:, 5], s[:, 6] & 0xf
    cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 0x1) << 4) | (((hw >> 13) & 0x7) << 5)
    print("%s, tile code %d: %d workgroups; XCD of block b for b < 16: %s" % (name, code, n, xcc[:16].tolist()))
    for x in (0, 1):
        sel = (xcc == x).nonzero().flatten()
        t0 = int(s[sel, 0].min())
        print("XCD %d: %d workgroups, %d CUs" % (x, len(sel), len(set(cu[sel].tolist()))))
        by = {}
        for b in sel.tolist():
            by.setdefault(int(cu[b]), []).append((b, int(s[b, 0]) - t0, int(s[b, 4]) - t0))
        for c in sorted(by):
            print("   cu %3d: " % c + "  ".join("b%-4d [%6d,%6d]" % v for v in by[c]))


if __name__ == "__main__":
    main()
