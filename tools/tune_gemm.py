"""Measured tile choice for the grouped GEMM launches of the training iteration (writes x-ggm_amd/gemm_tiles_gfx950.json).

Runs the bench's model for a few un-captured iterations; in front of EVERY grouped launch the same launch is timed under
every tile (1: 64 x 64, 2: 128 x 64, 3: 128 x 128 on four waves, 4: 128 x 128 on eight waves) and under the library's own
choice (0), in the step's real conditions: operands as the previous kernels left them, weights not in cache.  Launches
are aggregated by signature (ops.gemm_signature); a tile enters the table when its median beats the library's choice by
more than 8 %.  Re-running a launch is harmless for the measurement run (accumulating weight-gradient launches add
their product more than once: the run's numbers are thrown away).

    python tools/tune_gemm.py [--order gqa] [--dtype fp8]     (on an MI355X)
"""
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["XGGM_TILE_TABLE"] = "0"  # tune against the library's own choices
import bench  # noqa: E402
from xggm_amd import ops, _lib  # noqa: E402
import ctypes as ct  # noqa: E402

PINS = (0, 1, 2, 3, 4, 7, 8, 9)  # 7 / 8 / 9: the role k-loop (four loader + four compute waves) on 128 x 128 / 128 x 64 / 64 x 64
REPS = 3


def main():
    args = bench.parse(sys.argv[1:] + ["--steps", "1", "--warmup", "0"])
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    from xggm_amd.engine import CapturedTrainer
    model, optim, batch = bench.build(args, dev)
    tr = CapturedTrainer(model, optim, batch, sigma=1.0, order=args.order, use_graph=False)
    for kind in ("plain", "rel", "node"):
        tr._eager_pass(kind)  # arena, fp8 calibration, schedules
    torch.cuda.synchronize()
    times = {}  # signature -> pin -> [us]

    def hook(dt, chunk, sig, arr):
        name = "xggm_gemm_grouped_fp8e4m3" if dt == "e4m3" else "xggm_gemm_grouped_" + ops.sfx(dt)
        fn = getattr(_lib.lib, name)
        st = ops.stream()
        rec = times.setdefault(sig, {p: [] for p in PINS})
        for pin in PINS:
            _lib.lib.xggm_gemm_set_group_tile(pin)
            e = [torch.cuda.Event(enable_timing=True) for _ in range(REPS + 1)]
            e[0].record()
            for r in range(REPS):
                rc = fn(ct.cast(arr, ct.c_void_p), len(chunk), st)
                assert rc == 0, _lib.last_error()
                e[r + 1].record()
            rec[pin].append(e)
        _lib.lib.xggm_gemm_set_group_tile(0)
        return 0

    ops.TILE_HOOK = hook
    try:
        for it in range(2):
            for kind in ("plain", "rel", "node"):
                torch.cuda.synchronize()
                torch.cuda._sleep(int(5e7))  # the host runs ahead: the events bracket back-to-back GPU work
                tr._eager_pass(kind)
    finally:
        ops.TILE_HOOK = None
    torch.cuda.synchronize()
    table, report = {}, []
    for sig, rec in times.items():
        med = {}
        for pin, runs in rec.items():
            us = [1000.0 * ev[r].elapsed_time(ev[r + 1]) for ev in runs for r in range(1, REPS)]  # first repetition: warm-up
            med[pin] = statistics.median(us)
        best = min((p for p in PINS if p), key=lambda p: med[p])
        n = len(rec[0])
        gain = med[0] - med[best]
        report.append((gain * n, sig, n, med, best))
        if med[best] < 0.92 * med[0]:  # marginal wins measured eagerly did not survive the replayed step (same-box A/B)
            table[sig] = best
    report.sort(reverse=True)
    for tot, sig, n, med, best in report[:40]:
        print("%-100s n=%3d  lib %6.1f | %s | best %d%s" % (sig[:100], n, med[0], " ".join("%6.1f" % med[p] for p in PINS[1:]), best,
                                                          "  <- table" if sig in table else ""))
    saved = sum(max(0.0, med[0] - med[b]) * n for _, sig, n, med, b in report if sig in table)
    print("launch signatures: %d, in the table: %d, modelled saving over the measured passes (2 x plain, rel, node): %.0f us"
          % (len(report), len(table), saved))
    suffix = "" if (args.order == "vqa" and args.dtype == "bf16") else "_%s_%s" % (args.order, args.dtype)
    out = os.path.join(ROOT, "gpurun_out", "gemm_tiles_gfx950%s.json" % suffix)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump({"device": torch.cuda.get_device_name(0), "what": "tools/tune_gemm.py: tile per grouped-launch signature where a "
               "measurement beat the library's cost model by > 8 %", "tiles": table}, open(out, "w"), indent=1, sort_keys=True)
    print("written:", out)


if __name__ == "__main__":
    main()
