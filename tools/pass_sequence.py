"""The ordered kernel sequence of the last two passes of a run, from a rocprofv3 --kernel-trace CSV (directory given):
which launches sit between the GEMM / LayerNorm / attention launches, to see what could share a launch.
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/seq -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline \
        --no-ref-batch --no-loader --no-kernel-timing;  python tools/pass_sequence.py gpurun_out/seq"""
import csv
import glob
import re
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    m = re.match(r"([A-Za-z0-9_:]+(<[^(]*>)?)", n)
    return (m.group(1) if m else n)[:60]


def main():
    f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
    rows.sort()
    adv = [i + 1 for i, r in enumerate(rows) if "bertadam_multi" in r[2]]  # a pass ends with its update
    for p in (3, 2):
        seg = rows[adv[-p]:adv[-p + 1]]
        print("---- pass of %d launches, %.3f ms" % (len(seg), (seg[-1][1] - seg[0][0]) / 1e6))
        run, last = 0, None
        for s, e, n in seg:
            k = short(n)
            big = "gemm_" in k or "ln_fwd" in k or "ln_bwd" in k or "attn_" in k
            if big:
                run += 1
                continue
            if run:
                print("      ... %d gemm / layernorm / attention launches" % run)
                run = 0
            print("  %7.1f us  %s" % ((e - s) / 1e3, k))
        if run:
            print("      ... %d gemm / layernorm / attention launches" % run)


if __name__ == "__main__":
    main()
