R=$GRAFT_REPO_ROOT
cd $R; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d gpurun_out/g3_pmcA -o pmc -- python tools/exp/gemm3b_test.py > gpurun_out/g3_pmcA.log 2>&1
python tools/pmc_by_grid.py $(find gpurun_out/g3_pmcA -name "*.db" | head -1) gemm3b | head -4
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM -d gpurun_out/g3_pmcB -o pmc -- python tools/exp/gemm3b_test.py > gpurun_out/g3_pmcB.log 2>&1
python tools/pmc_by_grid.py $(find gpurun_out/g3_pmcB -name "*.db" | head -1) gemm3b | head -4
rm -rf gpurun_out/g3_pmcA gpurun_out/g3_pmcB
