cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ITERS=3 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d gpurun_out/r02_pmcC -o pmc -- python tools/exp/bgemm_test.py > gpurun_out/r02_pmcC.log 2>&1
python tools/pmc_by_grid.py $(find gpurun_out/r02_pmcC -name "*.db" | head -1) bgemm > gpurun_out/r02_pmcC_summary.txt 2>&1
ITERS=3 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM -d gpurun_out/r02_pmcD -o pmc -- python tools/exp/bgemm_test.py > gpurun_out/r02_pmcD.log 2>&1
python tools/pmc_by_grid.py $(find gpurun_out/r02_pmcD -name "*.db" | head -1) bgemm > gpurun_out/r02_pmcD_summary.txt 2>&1
rm -rf gpurun_out/r02_pmcC gpurun_out/r02_pmcD
cat gpurun_out/r02_pmcC_summary.txt gpurun_out/r02_pmcD_summary.txt
