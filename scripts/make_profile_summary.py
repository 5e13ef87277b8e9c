#!/usr/bin/env python3
"""Turn gpurun_out/{prof,pmc}_<tag> into the tracked files under profiles/: kernel stats CSV, PMC summary text, and
profiles/<out>_traffic.json (HBM bytes per GEMM launch for bench.py's roofline.traffic).

FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts a wide coalesced read at half its bytes
(MI355X_MICROARCH.md, HBM section), so read bytes = 2 * FETCH_SIZE * 1024, write bytes = WRITE_SIZE * 1024."""
import csv
import json
import os
import shutil
import subprocess
import sys
from collections import defaultdict

tag, out = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = os.path.join(root, "gpurun_out")
p = os.path.join(root, "profiles")
os.makedirs(p, exist_ok=True)
shutil.copy(os.path.join(g, f"prof_{tag}", f"{tag}_kernel_stats.csv"), os.path.join(p, f"{out}_kernel_stats.csv"))
files = [os.path.join(g, f"pmc_{tag}_{n}", f"{n}_counter_collection.csv") for n in ("sq", "fetch", "write", "lds", "tcc")]
with open(os.path.join(p, f"{out}_pmc_summary.txt"), "w") as f:
    for group in ([files[0]], files[1:3], [files[3]], [files[4]]):
        f.write(subprocess.run([sys.executable, os.path.join(root, "scripts", "pmc_summary.py")] + group,
                               capture_output=True, text=True).stdout + "\n")


def per_kernel(path, counter):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and "gemm" in r["Kernel_Name"]:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


fetch, write = per_kernel(files[1], "FETCH_SIZE"), per_kernel(files[2], "WRITE_SIZE")
n = sum(len(v) for v in fetch.values())
rd = 2.0 * 1024.0 * sum(sum(v) for v in fetch.values()) / n
wr = 1024.0 * sum(sum(v) for v in write.values()) / sum(len(v) for v in write.values())
sq = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(files[0])):
    if "gemm" in r["Kernel_Name"]:
        sq[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
busy = sum(sum(v["SQ_VALU_MFMA_BUSY_CYCLES"]) for v in sq.values())
act = sum(sum(v["GRBM_GUI_ACTIVE"]) for v in sq.values())
util = busy / (act / 8.0 * 1024.0)
json.dump({"gemm_hbm_bytes_per_launch": rd + wr, "read_bytes": rd, "write_bytes": wr, "launches_sampled": n,
           "gemm_mfma_util_pmc": util,
           "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over bench.py; read = 2*FETCH_SIZE KiB "
                     "(gfx950 half-count correction), write = WRITE_SIZE KiB; mean over all GEMM dispatches. MFMA util = "
                     "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs)."},
          open(os.path.join(p, f"{out}_traffic.json"), "w"), indent=1)
print(open(os.path.join(p, f"{out}_traffic.json")).read())
