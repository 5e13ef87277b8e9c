#!/usr/bin/env python3
"""Turn gpurun_out/{<tag>_bench.json, <tag>_kernel_stats.csv, pmc_<tag>_*} (scripts/prof_config.sh) into the tracked files
under profiles/: <out>_bench.json, <out>_kernel_stats.csv, <out>_pmc_summary.txt, and one entry of profiles/r04_traffic.json
(HBM bytes per GEMM launch + PMC MFMA utilisation, keyed by workload, labelled with the commit it was measured on) that
bench.py reports as roofline.traffic.

    python scripts/make_profile_summary.py <tag> <out> <mode:model:size:batch:dtype> <commit>

FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts a wide coalesced read at half its bytes
(MI355X_MICROARCH.md, HBM section), so read bytes = 2 * FETCH_SIZE * 1024, write bytes = WRITE_SIZE * 1024.

Round 3: only the GEMM kernels OF THE WORKLOAD'S DTYPE are summed (gemm_bf16* / gemm_fp8* / the fp32 family).  The round-2
sums took every kernel with "gemm" in its name, so the fp32 calibration GEMMs of the fp8 build (gemm_f32_mfma /
gemm_panel_f32, ~80 % MFMA busy, large tiles) and the fp32 patch-embed GEMM of the bf16 builds contaminated the fp8 / bf16
rows: the "51.7 % MFMA busy, 113 MB / launch" quoted for the fp8 GEMMs was really 13.4 % and ~57 MB."""
import csv
import json
import os
import shutil
import subprocess
import sys
from collections import defaultdict

tag, out, key, commit = sys.argv[1:5]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = os.path.join(root, "gpurun_out")
p = os.path.join(root, "profiles")
os.makedirs(p, exist_ok=True)
shutil.copy(os.path.join(g, f"{tag}_kernel_stats.csv"), os.path.join(p, f"{out}_kernel_stats.csv"))
shutil.copy(os.path.join(g, f"{tag}_bench.json"), os.path.join(p, f"{out}_bench.json"))
files = [os.path.join(g, f"pmc_{tag}_{n}", f"{n}_counter_collection.csv") for n in ("sq", "fetch", "write", "lds")]
with open(os.path.join(p, f"{out}_pmc_summary.txt"), "w") as f:
    for group in ([files[0]], files[1:3], [files[3]]):
        f.write(subprocess.run([sys.executable, os.path.join(root, "scripts", "pmc_summary.py")] + group,
                               capture_output=True, text=True).stdout + "\n")


dtype = key.split(":")[-1]
FAMILY = {"f32": ("gemm_panel_f32", "gemm_f32_mfma", "gemm_thin_f32"), "bf16": ("gemm_bf16",), "fp8": ("gemm_fp8",),
          "f32x3": ("gemm_bf16",), "f32x6": ("gemm_bf16",)}[dtype]      # split-fp32 builds: their GEMMs ARE the bf16 kernels


def in_family(name: str) -> bool:
    return any(f in name for f in FAMILY)


def per_kernel(path, counter):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and in_family(r["Kernel_Name"]):
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


fetch, write = per_kernel(files[1], "FETCH_SIZE"), per_kernel(files[2], "WRITE_SIZE")
n = sum(len(v) for v in fetch.values())
rd = 2.0 * 1024.0 * sum(sum(v) for v in fetch.values()) / n
wr = 1024.0 * sum(sum(v) for v in write.values()) / sum(len(v) for v in write.values())
sq = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(files[0])):
    if in_family(r["Kernel_Name"]):
        sq[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
busy = sum(sum(v["SQ_VALU_MFMA_BUSY_CYCLES"]) for v in sq.values())
act = sum(sum(v["GRBM_GUI_ACTIVE"]) for v in sq.values())
util = busy / (act / 8.0 * 1024.0)
lds = defaultdict(float)
for r in csv.DictReader(open(files[3])):
    if in_family(r["Kernel_Name"]):
        lds[r["Counter_Name"]] += float(r["Counter_Value"])
per_kernel_util = {}
for name, v in sq.items():
    short = name.replace("void ldit::(anonymous namespace)::", "").split("(")[0]
    per_kernel_util[short] = {"launches": len(v["SQ_VALU_MFMA_BUSY_CYCLES"]),
                              "mfma_util": sum(v["SQ_VALU_MFMA_BUSY_CYCLES"]) / (sum(v["GRBM_GUI_ACTIVE"]) / 8.0 * 1024.0)}
path = os.path.join(p, os.environ.get("TRAFFIC_JSON", "r04_traffic.json"))     # bench.py reads the newest one
table = json.load(open(path)) if os.path.exists(path) else {}
table[key] = {"commit": commit, "gemm_hbm_bytes_per_launch": rd + wr, "read_bytes": rd, "write_bytes": wr, "launches_sampled": n,
              "gemm_mfma_util_pmc": util, "kernels": sorted(FAMILY), "per_kernel_mfma_util": per_kernel_util,
              "gemm_lds_bank_conflict_frac": lds["SQ_LDS_BANK_CONFLICT"] / max(lds["SQ_LDS_IDX_ACTIVE"], 1.0),
              "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over bench.py; read = 2*FETCH_SIZE KiB "
                        "(gfx950 half-count correction), write = WRITE_SIZE KiB; mean over the dispatches of the workload's own GEMM kernels "
                        "(`kernels`; calibration / patch-embed GEMMs of another dtype excluded). MFMA util = "
                        "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs); LDS conflicts = SQ_LDS_BANK_CONFLICT / "
                        "SQ_LDS_IDX_ACTIVE summed over the GEMM dispatches."}
json.dump(table, open(path, "w"), indent=1)
print(json.dumps(table[key], indent=1))
