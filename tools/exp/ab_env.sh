#!/bin/bash
# A/B of one environment switch inside the training step, interleaved on ONE box: bash tools/exp/ab_env.sh DENET_WINO4G 0 1
VAR=$1; shift
for rep in 1 2; do for v in "$@"; do env $VAR=$v python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-warm --no-split-bf16 --no-configs --no-dp-selftest 2>&1 | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read()); a = d['roofline']['all_igemm']
print('$VAR=$v', d['value'], d['ms_per_step'], {k: (v['ms_per_step'], v['tflops']) for k, v in a.items() if 'wino4' in k or k.startswith('igemm_kernel<2, 128, 128, 2, 2, 1')})"; done; done
