# In-model A/B of two builds of the library on one box, alternating processes: bench.py of the given configs with the shipped library
# and with layoutdit_amd/csrc/build/libldit_<name>.so (scripts/build_alt.sh; LDIT_LIB_PATH).  Isolated kernel timings on this chip
# overstate what a change is worth inside a model (a kernel run back to back throttles the clock differently): decide here.
#   bash scripts/ab_model_lib.sh <name> <config> [<config> ...]
NAME=$1; shift
B=layoutdit_amd/csrc/build
for r in 1 2 3; do for c in "$@"; do for v in shipped $NAME; do
  if [ $v = shipped ]; then unset LDIT_LIB_PATH; else export LDIT_LIB_PATH=$PWD/$B/libldit_$v.so; fi
  python bench.py --config $c --steps 20 --warmup 5 --no-roofline-pass --cpu-sample 0 --no-split-fp32 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cfg',$c,'$v', d['value'], d['ms_per_step'])"
done; done; done
