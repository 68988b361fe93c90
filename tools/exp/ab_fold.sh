#!/bin/bash
# A / B on ONE box, interleaved: the batch-norm reductions finished in the producing launches (default) against the separate
# final kernels (DENET_BN_FINAL_FOLD=0)
F="--steps 20 --warmup 3 --no-cpu-baseline --no-warm --no-split-bf16 --no-configs --no-dp-selftest --no-h2d --no-instep --no-audit --no-roofline"
for i in 1 2 3; do
  for v in 3 2 0; do
    DENET_BN_FINAL_FOLD=$v python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('fold=$v', d['value'], d['ms_per_step'])"
  done
done
