#!/bin/bash
# A / B of environment settings on the headline leg: bash tools/exp/ab_env.sh "VAR=a" "VAR=b" [pairs]
A="$1"; B="$2"; N=${3:-4}
CMD="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-warm --no-split-bf16 --no-configs --no-dp-selftest --no-h2d --no-instep --no-audit --no-roofline --no-inference"
val() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        print(json.loads(l)['value'])
"; }
for i in $(seq 1 $N); do
  echo "A[$A] $(env $A $CMD 2>/dev/null | val)   B[$B] $(env $B $CMD 2>/dev/null | val)"
done
