#!/bin/bash
# A / B of two source trees on the headline leg (A = tools/exp/_old_tree, a `git archive` of the commit before + its library; B = this tree)
N=${1:-5}
ARGS="--steps 20 --warmup 3 --no-cpu-baseline --no-warm --no-split-bf16 --no-configs --no-dp-selftest --no-h2d --no-instep --no-audit --no-roofline"
val() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        print(json.loads(l)['value'])
"; }
for i in $(seq 1 $N); do
  echo "A $(python tools/exp/_old_tree/bench.py $ARGS 2>/dev/null | val)   B $(python bench.py $ARGS 2>/dev/null | val)"
done
