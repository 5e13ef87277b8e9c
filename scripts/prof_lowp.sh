#!/bin/bash
# Kernel trace of a low-precision bench line ($DT, $MODEL, $SIZE, $BS): per-kernel durations and the gaps between them
# (GPU box).  The *_kernel_stats.csv it leaves under gpurun_out/prof_<tag>/ is what profiles/r01_cfg*_kernel_stats.csv hold.
set -o pipefail
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-fp8}; DT=${DT:-fp8}; BS=${BS:-32}; MODEL=${MODEL:-base}; SIZE=${SIZE:-224}
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o $TAG -- python3 bench.py --model $MODEL --size $SIZE --dtype $DT --batch $BS --steps 5 --warmup 2 --cpu-sample 0 --no-roofline-pass > gpurun_out/prof_bench_$TAG.log 2>&1
echo "rocprofv3 rc=$?"
python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/prof_$TAG/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
n = len(rows)
last = rows[-100:]       # about one forward
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last)
span = int(last[-1]["End_Timestamp"]) - int(last[0]["Start_Timestamp"])
print(f"{n} kernels; last 100: span {span/1e3:.1f} us, busy {busy/1e3:.1f} us, gaps {(span-busy)/1e3:.1f} us")
agg = collections.defaultdict(list)
for r in last:
    agg[r["Kernel_Name"][:90]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{sum(v):9.1f} us  n={len(v):3d}  avg {sum(v)/len(v):7.1f}  {k}")
PY
