# In-model A/B of one diagnostic environment switch of the shipped library on one box, alternating processes (as ab_model_lib.sh):
#   bash scripts/ab_model_env.sh LDIT_TRAIN_SIDE_STREAM=1 <config> [<config> ...]
KV=$1; shift
for r in 1 2 3; do for c in "$@"; do for v in off on; do
  if [ $v = on ]; then export "$KV"; else unset "${KV%%=*}"; fi
  python bench.py --config $c --steps 20 --warmup 5 --no-roofline-pass --cpu-sample 0 --no-split-fp32 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cfg',$c,'$KV $v', d['value'], d['ms_per_step'])"
done; done; done
