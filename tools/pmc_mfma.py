"""MFMA pipe occupancy per kernel from one rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass (CSV dir given).
MfmaUtil = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (GRBM_GUI_ACTIVE per XCD * 1024 SIMDs); the CSV reports GRBM_GUI_ACTIVE
summed over the 8 XCDs, hence the division by 8."""
import collections
import csv
import glob
import sys


def main():
    f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            calls[r["Kernel_Name"]] += 1
    rows = []
    for k, a in acc.items():
        if a["SQ_VALU_MFMA_BUSY_CYCLES"] <= 0:
            continue
        act = a["GRBM_GUI_ACTIVE"] / 8.0
        rows.append((a["SQ_VALU_MFMA_BUSY_CYCLES"], k, calls[k], act / max(1, calls[k]), 100.0 * a["SQ_VALU_MFMA_BUSY_CYCLES"] / (act * 1024.0)))
    rows.sort(reverse=True)
    print("%-70s %6s %14s %10s" % ("kernel", "calls", "cycles/launch", "MfmaUtil %"))
    for _, k, n, c, u in rows[:14]:
        print("%-70s %6d %14.0f %10.1f" % (k[:70], n, c, u))


if __name__ == "__main__":
    main()
