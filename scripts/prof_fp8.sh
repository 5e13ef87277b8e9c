#!/bin/bash
# Kernel trace of the fp8 (or $DT) ViT-B bs=$BS forward: per-kernel durations and the gaps between them (GPU box).
set -o pipefail
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-fp8}; DT=${DT:-fp8}; BS=${BS:-32}
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o $TAG -- python3 bench.py --dtype $DT --batch $BS --steps 5 --warmup 2 --cpu-sample 0 --no-roofline-pass > gpurun_out/prof_bench_$TAG.log 2>&1
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
