"""families, launches and the GEMM roofline fraction out of a rocprofv3 --kernel-trace --stats CSV of bench.py
(python tools/stats_summary.py profiles/r03_bench_kernel_stats.csv): iterations are counted from the rng_advance
launches (one per pass, two passes per iteration); 2.04 TFLOP of GEMM work per iteration at 32 samples."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
flop_per_iter = float(sys.argv[2]) if len(sys.argv) > 2 else 2.04e12


def us(r):
    return float(r["TotalDurationNs"]) / 1e3


iters = sum(int(r["Calls"]) for r in rows if "rng_advance" in r["Name"]) / 2.0
fam = {}
for r in rows:
    n = r["Name"]
    if "spin_kernel" in n:
        continue
    k = ("gemm" if "gemm_" in n else "bertadam + norm" if ("bertadam" in n or "sqnorm" in n) else
         "layernorm" if ("ln_fwd" in n or "ln_bwd" in n) else "attention core" if "attn_" in n else "everything else")
    f = fam.setdefault(k, [0, 0.0])
    f[0] += int(r["Calls"])
    f[1] += us(r)
tot = sum(t for _, t in fam.values())
print("iterations %.1f, kernel time %.3f ms per iteration, %.0f launches per iteration" % (iters, tot / iters / 1e3, sum(c for c, _ in fam.values()) / iters))
for k, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    print("%-16s %6.1f launches/iter %8.1f us/iter %5.1f %%   %6.1f us per launch" % (k, c / iters, t / iters, 100 * t / tot, t / c))
g = fam["gemm"]
tf = flop_per_iter / (g[1] / iters * 1e-6) / 1e12
print("GEMM: %.1f TFLOP/s = %.4f of the 2500 TFLOP/s dense bf16 peak (%.0f launches, %.1f us average)" % (tf, tf / 2500.0, g[0], g[1] / g[0]))
