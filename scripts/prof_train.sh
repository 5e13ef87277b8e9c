#!/bin/bash
# rocprofv3 kernel trace of the train-step bench line (BASELINE configs[2]); summary -> gpurun_out/<tag>_train_kernel_stats.csv
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_train_$tag
rm -rf $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --config 2 --steps 5 --warmup 2 --no-roofline-pass > $GRAFT_REPO_ROOT/gpurun_out/${tag}_train_prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/${tag}_train_prof.err
f=$(find $out -name "*kernel_stats.csv" | head -1)
cp "$f" $GRAFT_REPO_ROOT/gpurun_out/${tag}_train_kernel_stats.csv
head -30 $GRAFT_REPO_ROOT/gpurun_out/${tag}_train_kernel_stats.csv | cut -c1-220
