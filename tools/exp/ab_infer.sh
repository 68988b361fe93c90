#!/bin/bash
# A / B of two tuned files on the inference leg (get_detections, B = 32, hard NMS): bash tools/exp/ab_infer.sh <fileA> <fileB> [pairs]
A="$1"; B="$2"; N=${3:-3}
CMD="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-warm --no-split-bf16 --no-configs --no-dp-selftest --no-h2d --no-instep --no-audit --no-roofline"
val() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l)['inference']; print(d.get('b1_nms',{}).get('value'), d.get('b32_nms',{}).get('value'), d.get('b32_soft_nms',{}).get('value'), d.get('error',''))
"; }
for i in $(seq 1 $N); do
  echo "A $(DENET_TUNE_CACHE=$A $CMD 2>/dev/null | val)   B $(DENET_TUNE_CACHE=$B $CMD 2>/dev/null | val)"
done
