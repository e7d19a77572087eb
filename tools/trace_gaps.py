"""Gaps between consecutive kernels inside replayed hipGraphs, from a rocprofv3 --kernel-trace CSV (directory given):
how much of an iteration is NOT inside any kernel (dispatch / dependency latency between graph nodes)."""
import csv
import glob
import sys


def main():
    f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
    rows.sort()
    # the last iterations of the run are graph replays: take the last passes (delimited by their update launches)
    adv = [i for i, r in enumerate(rows) if "bertadam_multi" in r[2]]  # a pass ends with its update
    n_pass = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    lo, hi = adv[-n_pass - 1] + 1, adv[-1] + 1
    seg = rows[lo:hi]
    busy = sum(e - s for s, e, _ in seg)
    span = seg[-1][1] - seg[0][0]
    gaps = [seg[i + 1][0] - seg[i][1] for i in range(len(seg) - 1)]
    pos = [g for g in gaps if g > 0]
    print("%d kernels over %d passes: span %.3f ms, inside kernels %.3f ms (%.1f %%), gaps %.3f ms = %.2f us per kernel boundary "
          "(median %.2f us, 95th percentile %.2f us), overlapped boundaries %d"
          % (len(seg), n_pass, span / 1e6, busy / 1e6, 100.0 * busy / span, sum(pos) / 1e6, sum(pos) / max(1, len(pos)) / 1e3,
             sorted(pos)[len(pos) // 2] / 1e3, sorted(pos)[int(len(pos) * 0.95)] / 1e3, len(gaps) - len(pos)))
    big = sorted(((g, seg[i][2][:50], seg[i + 1][2][:50]) for i, g in enumerate(gaps)), reverse=True)[:int(sys.argv[3]) if len(sys.argv) > 3 else 8]
    print("   gaps above 5 us: %d, together %.3f ms" % (sum(1 for g in pos if g > 5000), sum(g for g in pos if g > 5000) / 1e6))
    for g, a, b in big:
        print("   %8.2f us between %s -> %s" % (g / 1e3, a, b))


if __name__ == "__main__":
    main()
