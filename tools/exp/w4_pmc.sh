#!/bin/bash
# counters of the fused F(4x4) kernel (and the un-fused kernels beside it) on the geometries of tools/exp/w4_time.py
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out/w4_pmc
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $O/m -o m -- python tools/exp/w4_time.py ${1:-32} > $O/m.log 2>&1
python tools/pmc_mfma.py $(find $O/m -name "*.db" | head -1) > gpurun_out/w4_pmc_mfma.json
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $O/l -o l -- python tools/exp/w4_time.py ${1:-32} > $O/l.log 2>&1
python tools/pmc_summary.py $(find $O/l -name "*.db" | head -1) all | grep -i "ino4f\|igemm" > gpurun_out/w4_pmc_lds.txt
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES -d $O/i -o i -- python tools/exp/w4_time.py ${1:-32} > $O/i.log 2>&1
python tools/pmc_summary.py $(find $O/i -name "*.db" | head -1) all | grep -i "ino4f\|igemm" >> gpurun_out/w4_pmc_lds.txt
rm -rf $O
