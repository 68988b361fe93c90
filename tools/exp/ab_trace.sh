#!/bin/bash
# A/B of an environment switch under rocprofv3 --kernel-trace: in-step time of the kernels matching $2, per value of the switch $1
# usage: ab_trace.sh ENVVAR kernel_substring value1 value2 ...
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
VAR=$1; PAT=$2; shift 2
for v in "$@"; do
  rm -rf /tmp/ab; env $VAR=$v rocprofv3 --kernel-trace -d /tmp/ab -o ab -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-warm --no-split-bf16 --no-roofline > /tmp/ab.log 2>&1
  DB=$(find /tmp/ab -name "*.db" | head -1)
  python $R/tools/step_trace.py $DB > /tmp/ab_trace.txt
  echo "$VAR=$v: $(head -1 /tmp/ab_trace.txt) | $(grep -o '"value": [0-9.]*' /tmp/ab.log | head -1) | $PAT: $(grep "$PAT" /tmp/ab_trace.txt | awk '{s+=$2; n++} END {printf "%d launches, %.0f us", n, s}')"
done
