B=layoutdit_amd/csrc/build
for r in 1 2 3; do
for c in 3 5 2; do
for v in shipped m32; do
  if [ $v = shipped ]; then unset LDIT_LIB_PATH; else export LDIT_LIB_PATH=$PWD/$B/libldit_$v.so; fi
  python bench.py --config $c --steps 20 --warmup 5 --no-roofline-pass 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cfg',$c,'$v', d['value'], d['ms_per_step'])"
done; done; done
