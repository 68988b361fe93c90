#!/bin/bash
# A / B of two builds of libdenet_hip.so on the headline leg: bash tools/exp/ab_lib.sh <libA.so> <libB.so> [pairs]
A="$1"; B="$2"; N=${3:-4}
run() { python -c "
import sys, runpy
import denet_amd.lib as l
l.LIB_PATH = '$1'
sys.argv = ['bench.py'] + '--steps 20 --warmup 3 --no-cpu-baseline --no-warm --no-split-bf16 --no-configs --no-dp-selftest --no-h2d --no-instep --no-audit --no-roofline'.split()
runpy.run_path('bench.py', run_name='__main__')
" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        print(json.loads(l)['value'])
"; }
for i in $(seq 1 $N); do
  echo "A $(run $A)   B $(run $B)"
done
