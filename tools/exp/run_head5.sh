#!/bin/bash
# headline leg only, five fresh processes: the spread of `value` on one box
for i in 1 2 3 4 5; do
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-warm --no-split-bf16 --no-configs --no-dp-selftest --no-h2d --no-instep --no-audit --no-roofline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'])
"
done
