"""families, launches and the GEMM roofline fraction out of a rocprofv3 --kernel-trace --stats CSV of bench.py
(python tools/stats_summary.py profiles/r03_bench_kernel_stats.csv): iterations are counted from the update launches
(bertadam_multi: one per pass, two passes per iteration; before round 4's single update launch: from rng_advance, which
has since moved into the norm's finishing launch); 2.04 TFLOP of GEMM work per iteration at 32 samples."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
flop_per_iter = float(sys.argv[2]) if len(sys.argv) > 2 else 2.04e12


TRIMMED = []
SETUP = []


def calls_us(r):
    """(calls, total us) of one kernel with an outlier record removed: the stats CSV keeps only Total / Average / Max per
    kernel, so ONE record is recognisable -- a maximum above 100 x the average of the others (round 3: a 19.57 ms
    record of a 16 us kernel, a launch queued behind torch's spin_kernel, moved the family's fraction from 0.130 to 0.124)."""
    c, tot, mx = int(r["Calls"]), float(r["TotalDurationNs"]), float(r.get("MaxNs") or 0)
    if c > 1 and mx > 100.0 * (tot - mx) / (c - 1):
        TRIMMED.append((r["Name"][:70], mx / 1e3))
        return c - 1, (tot - mx) / 1e3
    return c, tot / 1e3


iters = sum(int(r["Calls"]) for r in rows if "bertadam_multi" in r["Name"]) / 2.0
if not iters:  # profiles of rounds 1-3
    iters = sum(int(r["Calls"]) for r in rows if "rng_advance" in r["Name"]) / 2.0
fam = {}
for r in rows:
    n = r["Name"]
    if "spin_kernel" in n:
        continue
    if "__amd_rocclr_" in n:
        # the runtime's copy / fill kernels: the parameters moving into the arena when the model is built (837 copies of
        # ~1 MB whatever --steps is; tools/find_aten.py finds none inside a pass)
        SETUP.append((n[:40], int(r["Calls"])))
        continue
    k = ("gemm" if "gemm_" in n else "bertadam + norm" if ("bertadam" in n or "sqnorm" in n or "clip_norm" in n) else
         "layernorm" if ("ln_fwd" in n or "ln_bwd" in n) else "attention core" if "attn_" in n else "everything else")
    f = fam.setdefault(k, [0, 0.0])
    c, t = calls_us(r)
    f[0] += c
    f[1] += t
tot = sum(t for _, t in fam.values())
for name, mx in TRIMMED:
    print("trimmed one outlier record: %.1f us of %s" % (mx, name))
for name, c in SETUP:
    print("not counted (model set-up, outside the iterations): %d x %s" % (c, name))
print("iterations %.1f, kernel time %.3f ms per iteration, %.0f launches per iteration" % (iters, tot / iters / 1e3, sum(c for c, _ in fam.values()) / iters))
for k, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    print("%-16s %6.1f launches/iter %8.1f us/iter %5.1f %%   %6.1f us per launch" % (k, c / iters, t / iters, 100 * t / tot, t / c))
g = fam["gemm"]
tf = flop_per_iter / (g[1] / iters * 1e-6) / 1e12
print("GEMM: %.1f TFLOP/s = %.4f of the 2500 TFLOP/s dense bf16 peak (%.0f launches, %.1f us average)" % (tf, tf / 2500.0, g[0], g[1] / g[0]))
