#!/usr/bin/env python3
"""Per-kernel sums of a rocprofv3 --pmc CSV (counter_collection.csv)."""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(k_\w+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:40]


def main(path):
    agg = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(int)
    seen = set()
    for row in csv.DictReader(open(path)):
        k = short(row["Kernel_Name"])
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        key = (row["Dispatch_Id"], k)
        if key not in seen:
            seen.add(key); calls[k] += 1
    for k in sorted(agg, key=lambda k: -sum(agg[k].values())):
        print("%s  calls=%d" % (k, calls[k]))
        for c, v in sorted(agg[k].items()):
            print("    %-40s %.6g   (per call %.6g)" % (c, v, v / max(calls[k], 1)))


if __name__ == "__main__":
    main(sys.argv[1])
