#!/bin/bash
# kernel breakdown of the fp32 forward at a serving-size batch:  bash scripts/prof_bs.sh <batch> [tag]
set -o pipefail
B=${1:-1}; TAG=${2:-bs$B}
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_$TAG -o $TAG -- python3 $ROOT/bench.py --batch $B --steps 20 --warmup 3 --cpu-sample 0 --no-roofline-pass --no-split-fp32 > $ROOT/gpurun_out/prof_$TAG.log 2>&1
echo "rc=$?"
find $ROOT/gpurun_out/prof_$TAG -name '*kernel_trace.csv' -delete
python3 - <<PY
import csv, glob
f = glob.glob("$ROOT/gpurun_out/prof_$TAG/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:10]:
    print(r['Name'][:78].ljust(78), r['Calls'].rjust(5), f"{float(r['AverageNs'])/1e3:8.1f} us", r['Percentage'])
PY
