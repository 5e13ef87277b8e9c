#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel name, mean counter value per dispatch."""
import csv
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void ldit::(anonymous namespace)::", "").replace("ldit::(anonymous namespace)::", "")
    return name.split("(")[0][:48]


def main(paths):
    agg = defaultdict(lambda: defaultdict(list))
    for p in paths:
        for r in csv.DictReader(open(p)):
            k = short(r["Kernel_Name"])
            if "at::native" in k or "rocclr" in k:
                continue
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    counters = sorted({c for k in agg for c in agg[k]})
    print("kernel".ljust(50), " ".join(c[-22:].rjust(22) for c in counters))
    for k in sorted(agg):
        vals = []
        for c in counters:
            v = agg[k].get(c)
            vals.append(("%.4g" % (sum(v) / len(v))).rjust(22) if v else "-".rjust(22))
        print(k.ljust(50), " ".join(vals), " n=%d" % max(len(v) for v in agg[k].values()))


if __name__ == "__main__":
    main(sys.argv[1:])
