#!/bin/bash
# PMC counters of the first-layer kernels (tools/exp/stem_time.py), one counter group per pass
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for G in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC"; do
  rm -rf /tmp/pm; rocprofv3 --kernel-trace --pmc $G -d /tmp/pm -o pm --output-format csv -- python $R/tools/exp/stem_time.py > /tmp/pm.log 2>&1
  python - <<PY
import csv, glob, collections, re
f = glob.glob('/tmp/pm/**/*counter_collection.csv', recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in f:
    for r in csv.DictReader(open(fn)):
        k = r['Kernel_Name']
        m = re.search(r'stem_\w+|igemm_kernel', k)
        if m:
            agg[m.group(0)][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
done
