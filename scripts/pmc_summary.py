#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel name, mean counter value per
dispatch (optionally only kernels matching a substring)."""
import collections
import csv
import glob
import sys


def main(root, pattern=""):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name", "")
                if pattern and pattern not in name:
                    continue
                acc[name[:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for name, counters in acc.items():
        print(name)
        for c, vals in sorted(counters.items()):
            print("   %-28s n=%d mean=%.4g" % (c, len(vals), sum(vals) / len(vals)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
