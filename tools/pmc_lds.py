"""LDS bank-conflict share per kernel from one rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE pass (CSV dir given)."""
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
        if r["Counter_Name"] == "SQ_LDS_IDX_ACTIVE":
            calls[r["Kernel_Name"]] += 1
    rows = [(a["SQ_LDS_IDX_ACTIVE"], k, calls[k], a["SQ_LDS_BANK_CONFLICT"]) for k, a in acc.items() if a["SQ_LDS_IDX_ACTIVE"] > 0]
    rows.sort(reverse=True)
    print("%-72s %6s %16s %16s %8s" % ("kernel", "calls", "LDS active/launch", "conflict/launch", "share %"))
    for act, k, n, conf in rows[:14]:
        print("%-72s %6d %16.0f %16.0f %8.2f" % (k[:72], n, act / n, conf / n, 100.0 * conf / act))


if __name__ == "__main__":
    main()
