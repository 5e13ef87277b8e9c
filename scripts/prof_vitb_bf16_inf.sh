ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/prof_x
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_x -o s -- python3 $ROOT/bench.py --config 1 --dtype bf16 --steps 5 --warmup 2 --cpu-sample 0 --no-roofline-pass --no-split-fp32 > $ROOT/gpurun_out/prof_x.log 2>&1
cp "$(find $ROOT/gpurun_out/prof_x -name '*kernel_stats.csv' | head -1)" $ROOT/gpurun_out/r04m_vitb_bf16_inf_kernel_stats.csv
rm -rf $ROOT/gpurun_out/prof_x
cd $ROOT && python scripts/kstats.py gpurun_out/r04m_vitb_bf16_inf_kernel_stats.csv 7 12
bash scripts/run_cfg2_check.sh 2>&1 | tail -28
