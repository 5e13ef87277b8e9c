for c in 1 2 3; do bash scripts/prof_config.sh r04c$c --config $c > gpurun_out/r04c${c}_prof.log 2>&1; tail -2 gpurun_out/r04c${c}_prof.log | cut -c1-200; done
