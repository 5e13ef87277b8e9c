python -m pytest tests/test_gpu_train.py tests/test_gpu_dp.py tests/test_gpu_fpn.py -x -q > gpurun_out/r04l_tests.log 2>&1; echo rc=$?; tail -4 gpurun_out/r04l_tests.log
for i in 1 2; do python bench.py --config 2 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r04l_cfg2_$i.json; python -c "
import json
d=json.load(open('gpurun_out/r04l_cfg2_$i.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['kernel_ms_per_step'])"; done
bash scripts/prof_stats.sh r04l 2 > /dev/null 2>&1
python scripts/kstats.py gpurun_out/r04l_cfg2_kernel_stats.csv 7 22
