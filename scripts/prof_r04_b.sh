for c in 4 5 6; do bash scripts/prof_config.sh r04c$c --config $c > gpurun_out/r04c${c}_prof.log 2>&1; tail -2 gpurun_out/r04c${c}_prof.log | cut -c1-200; done
