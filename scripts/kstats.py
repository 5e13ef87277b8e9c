#!/usr/bin/env python3
"""Per-step kernel breakdown of a rocprofv3 --kernel-trace --stats CSV: python scripts/kstats.py <kernel_stats.csv> [steps=7] [top=24]"""
import csv, sys
path = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 7
top = int(sys.argv[3]) if len(sys.argv) > 3 else 24
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:top]:
    name = r["Name"].replace("ldit::(anonymous namespace)::", "").replace("void ", "")
    print(f"{float(r['TotalDurationNs']) / steps / 1e3:8.1f} us/step {int(r['Calls']) // steps:4d} calls  avg {float(r['AverageNs']) / 1e3:7.1f} us  {name[:100]}")
print(f"total {tot / steps / 1e6:.3f} ms/step")
