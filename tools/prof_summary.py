"""Print the top rows of a rocprofv3 kernel_stats.csv (found under the directory given)."""
import csv
import glob
import sys


def main():
    root = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    f = glob.glob(root + "/**/*kernel_stats.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    tot = sum(int(r["TotalDurationNs"]) for r in rows)
    print("total kernel time %.3f ms over %d kernels" % (tot / 1e6, len(rows)))
    for r in rows[:top]:
        print("%-88s %6s %12s %10.1f %6s" % (r["Name"][:88], r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]),
                                             r["Percentage"]))


if __name__ == "__main__":
    main()
