"""Phase breakdown of the grouped GEMM tile routine from in-kernel cycle stamps (instrumented build:
`make -C x-ggm_amd/csrc stamp`).  Slots: 0 entry, 1 prologue done, 2 k-loop done, 3 first epilogue
half done, 4 exit."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["XGGM_LIB"] = os.path.join(ROOT, "x-ggm_amd", "csrc", "build_stamp", "libxggm_hip.so")
sys.path.insert(0, ROOT)
import torch
from xggm_amd import ops, _lib
from tools.gemm_ktime import problem
from tools.bench_gemm import timeit

BF = torch.bfloat16


def run(tile_code, forms_shapes, label):
    _lib.lib.xggm_gemm_set_group_tile(tile_code)
    made = [problem(f, M, N, K) for f, M, N, K in forms_shapes]
    ps = [m[0] for m in made]
    t_us = timeit(lambda: ops.gemm_group(BF, ps)) * 1e6
    buf = torch.zeros(8 * 8192, dtype=torch.int64, device="cuda")
    fn = _lib.lib.xggm_gemm_set_stamp
    fn.argtypes = [ctypes.c_void_p]
    fn(buf.data_ptr())
    ops.gemm_group(BF, ps)
    torch.cuda.synchronize()
    fn(None)
    s = buf.view(-1, 8).cpu()
    s = s[s[:, 0] != 0].double()
    t0 = s[:, 0].min()
    span = (s[:, 4].max() - t0).item()
    d = [(s[:, i + 1] - s[:, i]).mean().item() for i in range(4)]
    stg = (s[:, 7] - s[:, 2]).mean().item()
    start = (s[:, 0] - t0)
    # co-residency: workgroups whose [entry, exit] intervals overlap on one CU (counters are per XCD)
    hw, xcc = s[:, 5].long(), s[:, 6].long() & 0xf
    cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 0x1) << 4) | (((hw >> 13) & 0x7) << 5) | (xcc << 8)
    maxc, ncu = 0, len(set(cu.tolist()))
    for c in set(cu.tolist()):
        iv = s[cu == c]
        ev = sorted([(a.item(), 1) for a in iv[:, 0]] + [(b.item(), -1) for b in iv[:, 4]])
        live = 0
        for _, dlt in ev:
            live += dlt
            maxc = max(maxc, live)
    # k-loop and whole-tile cycles by how many workgroups shared the tile's CU during the launch
    cnt = {c: int((cu == c).sum()) for c in set(cu.tolist())}
    share = torch.tensor([cnt[c] for c in cu.tolist()])
    for n in sorted(set(share.tolist())):
        sel = s[share == n]
        per_cu = [(s[cu == c][:, 4].max() - s[cu == c][:, 0].min()).item() for c in cnt if cnt[c] == n]
        print("   CUs holding %d tile(s): %4d tiles, k-loop %7.0f, entry->exit %7.0f per tile, first entry -> last exit on the CU %7.0f"
              % (n, sel.shape[0], (sel[:, 2] - sel[:, 1]).mean().item(), (sel[:, 4] - sel[:, 0]).mean().item(),
                 sum(per_cu) / len(per_cu)), flush=True)
    # the cycle counters of the eight XCDs are not synchronised: first entry -> last exit per XCD
    spans = [(s[xcc == x][:, 4].max() - s[xcc == x][:, 0].min()).item() for x in sorted(set(xcc.tolist()))]
    late = [(s[xcc == x][:, 0].max() - s[xcc == x][:, 0].min()).item() for x in sorted(set(xcc.tolist()))]
    span = sum(spans) / len(spans)
    print("   per XCD: first entry -> last exit %s cycles; last entry - first entry %s" % (
        " ".join("%.0f" % v for v in spans), " ".join("%.0f" % v for v in late)), flush=True)
    print("%-44s %7.1f us | %4d tiles, span %8.0f cyc (%.0f cyc/us) | prologue %6.0f  k-loop %7.0f  epi0 %6.0f  epi1 %6.0f | "
          "(staging %5.0f) CUs used %d, max co-resident %d" % (label, t_us, s.shape[0], span, span / t_us, d[0], d[1], d[2], d[3], stg,
                                                                ncu, maxc),
          flush=True)


def ablation():
    """kernel time with the epilogue's chunk loop (1) or the whole epilogue (2) removed"""
    groups = {
        "FFN2-bwd group": [("wgrad", 3072, 768, 1152), ("dgrad", 1152, 3072, 768), ("wgrad", 3072, 768, 640), ("dgrad", 640, 3072, 768)],
        "FFN1-fwd pair": [("fwd", 1152, 3072, 768), ("fwd", 640, 3072, 768)],
        "QKV-fwd pair": [("fwd", 1152, 2304, 768), ("fwd", 640, 2304, 768)],
        "attn-out fwd pair": [("fwd", 1152, 768, 768), ("fwd", 640, 768, 768)],
    }
    for code, name in ((3, "128x128"), (2, "128x64"), (1, "64x64")):
        _lib.lib.xggm_gemm_set_group_tile(code)
        for gname, shapes in groups.items():
            made = [problem(f, M, N, K) for f, M, N, K in shapes]
            ps = [m[0] for m in made]
            row = []
            for bits in (0, 1, 2):
                _lib.lib.xggm_gemm_set_ablate(bits)
                row.append(timeit(lambda: ops.gemm_group(BF, ps)) * 1e6)
            _lib.lib.xggm_gemm_set_ablate(0)
            print("%-8s %-20s full %6.1f us | no chunk loop %6.1f | no epilogue %6.1f" % (name, gname, *row), flush=True)
    _lib.lib.xggm_gemm_set_group_tile(0)


def main():
    ablation()
    for code, name in ((3, "128x128"), (1, "64x64")):
        for K in (64, 768):
            run(code, [("fwd", 2048, 1024, K)] * 2, "%s fwd 2x(2048x1024x%d)" % (name, K))
        run(code, [("wgrad", 3072, 768, 1152), ("dgrad", 1152, 3072, 768), ("wgrad", 3072, 768, 640), ("dgrad", 640, 3072, 768)],
            "%s FFN2-bwd group" % name)
        run(code, [("fwd", 1152, 3072, 768), ("fwd", 640, 3072, 768)], "%s FFN1-fwd pair" % name)
    _lib.lib.xggm_gemm_set_group_tile(0)


if __name__ == "__main__":
    main()
