#!/usr/bin/env python
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per kernel name."""
import collections
import csv
import glob
import sys


def summarise(path):
    rows = list(csv.DictReader(open(path)))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = []
    for k, v in sorted(agg.items()):
        short = k.replace("(sg::StageArgs)", "").replace("void ", "")
        out.append((short, {c: sum(x) / len(x) for c, x in v.items()}, len(next(iter(v.values())))))
    return out


if __name__ == "__main__":
    for d in sys.argv[1:]:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            print("#", f)
            for name, vals, n in summarise(f):
                print("%-60s n=%-4d %s" % (name[:60], n, "  ".join("%s=%.6g" % kv for kv in sorted(vals.items()))))
