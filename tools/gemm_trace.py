"""Timeline of a tile of the grouped GEMM (instrumented build): the first wave of every workgroup stamps the cycle
counter around each step of the LDS-DMA k-loop (wait for the tile's loads, barrier, issue of the next loads, fragment
reads + MFMAs) and of the register epilogue; averages over the workgroups of one launch.
   make -C x-ggm_amd/csrc stamp && python tools/gemm_trace.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from tools.gemm_stamps import _lib, ops, problem, BF  # noqa: E402
from tools.gemm_phase_report import GROUPS  # noqa: E402

TRACE_BASE = 8 * 8192


def trace(tile_code, forms_shapes, label, bits=0, ablate=0):
    _lib.lib.xggm_gemm_set_ablate(ablate | 0x10000)  # 0x10000: the timeline points are live
    _lib.lib.xggm_gemm_set_group_tile(tile_code)
    _lib.lib.xggm_gemm_set_tile(bits)
    made = [problem(f, M, N, K) for f, M, N, K in forms_shapes]
    ps = [m[0] for m in made]
    for _ in range(3):
        ops.gemm_group(BF, ps)
    buf = torch.zeros(TRACE_BASE + 128 * 8192, dtype=torch.int64, device="cuda")
    fn = _lib.lib.xggm_gemm_set_stamp
    fn.argtypes = [ctypes.c_void_p]
    fn(buf.data_ptr())
    ops.gemm_group(BF, ps)
    torch.cuda.synchronize()
    fn(None)
    _lib.lib.xggm_gemm_set_ablate(0)
    s = buf[:TRACE_BASE].view(-1, 8).cpu()
    n = int((s[:, 0] != 0).sum())
    tr = buf[TRACE_BASE:].view(-1, 128)[:n].cpu().double()
    K = forms_shapes[0][3] if forms_shapes[0][0] != "wgrad" else None
    nk = min(K // 64, 27) if K else 10
    e = tr[:, 16:16 + 4 * nk].view(n, nk, 4)
    nxt = torch.cat([e[:, 1:, 0], tr[:, 14:15]], 1)  # top of the next iteration / end of the loop
    wait = (e[:, :, 1] - e[:, :, 0]).mean(0)
    bar = (e[:, :, 2] - e[:, :, 1]).mean(0)
    iss = (e[:, :, 3] - e[:, :, 2]).mean(0)
    mma = (nxt - e[:, :, 3]).mean(0)
    print("%s: %d workgroups, k-tiles traced %d" % (label, n, nk))
    print("   prologue issue -> loop top %6.0f" % (e[:, 0, 0] - tr[:, 15]).mean().item())
    fmt = lambda v: " ".join("%5.0f" % x for x in v.tolist())
    print("   wait for loads  " + fmt(wait))
    print("   barrier         " + fmt(bar))
    print("   issue next      " + fmt(iss))
    print("   reads + MFMAs   " + fmt(mma))
    print("   per k-tile      " + fmt(wait + bar + iss + mma) + "   mean %.0f" % (wait + bar + iss + mma).mean().item())
    ep = tr[:, 0:6]
    print("   last barrier %5.0f | epilogue: bias arrives %5.0f, block rows %s, stores acknowledged %5.0f" % (
        (tr[:, 0] - tr[:, 14]).mean().item(), (tr[:, 1] - tr[:, 0]).mean().item(),
        fmt(torch.cat([(ep[:, 3:6] - ep[:, 2:5]), (tr[:, 12:13] - ep[:, 5:6])], 1).mean(0)), (tr[:, 13] - tr[:, 12]).mean().item()),
        flush=True)


if __name__ == "__main__":
    if sys.argv[1:2] == ["ablate"]:
        # needs  make stamp EXTRA=-DXGGM_KABLATE : 8 no LDS-DMA in the loop, 32 no MFMAs, 256 no fragment reads, 128 no barrier
        for code, tn in ((3, "128x128"), (2, "128x64")):
            for ab in (0, 8, 32, 256, 8 | 32, 8 | 256, 32 | 256):
                trace(code, GROUPS["QKV fwd pair"], "QKV fwd pair %-8s ablate %3d" % (tn, ab), 0, ab)
        _lib.lib.xggm_gemm_set_ablate(0)
        sys.exit(0)
    for name in sys.argv[1:] or ["QKV fwd pair", "FFN1 fwd pair"]:
        for code, tn in ((2, "128x64"), (3, "128x128"), (4, "128x128/8w")):
            for bits, ns in ((0, "default stages"), (0x1000, "3 stages")):
                trace(code, GROUPS[name], "%-16s %-10s %s" % (name, tn, ns), bits)
    _lib.lib.xggm_gemm_set_tile(0)
    _lib.lib.xggm_gemm_set_group_tile(0)
