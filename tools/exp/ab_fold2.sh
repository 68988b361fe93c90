#!/bin/bash
F="--steps 10 --warmup 3 --no-cpu-baseline --no-warm --no-split-bf16 --no-configs --no-dp-selftest --no-h2d --no-instep --no-audit"
for v in 1 0; do
  DENET_BN_FINAL_FOLD=$v python bench.py $F 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('fold=$v', d['value'], d['ms_per_step'])
for k,v in d['roofline']['all_igemm'].items(): print('   %-40s %3d %8.3f ms %7.2f TF' % (k, v['launches_per_step'], v['ms_per_step'], v['tflops']))
"
done
