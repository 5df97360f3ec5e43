"""Print per-launch averages of the counters of one kernel from rocprofv3 --pmc CSV passes.

    python tools/pmc_show.py <kernel-name-substring> dir1 [dir2 ...]
"""
import collections
import csv
import glob
import os
import sys


def main():
    key, dirs = sys.argv[1], sys.argv[2:]
    agg = collections.defaultdict(list)
    for d in dirs:
        for p in glob.glob(os.path.join(d, "*counter_collection.csv")):
            per = collections.defaultdict(float)
            for r in csv.DictReader(open(p)):
                if key in r["Kernel_Name"]:
                    per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
            for (_, c), v in per.items():
                agg[c].append(v)
    for c, v in sorted(agg.items()):
        print(f"{c:28s} {sum(v) / len(v):.4g}")


if __name__ == "__main__":
    main()
