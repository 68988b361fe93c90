# kernel trace + PMC passes over the fused F(2x2) filter-gradient kernel alone (tools/exp/w2g_test.py)
R=$GRAFT_REPO_ROOT
cd $R; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d gpurun_out/w2g_kt -o kt -- python tools/exp/w2g_test.py > gpurun_out/w2g_kt.log 2>&1
python tools/kt_by_grid.py $(find gpurun_out/w2g_kt -name "*.db" | head -1) w2g
python tools/kt_by_grid.py $(find gpurun_out/w2g_kt -name "*.db" | head -1) wino2f_wgrad
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d gpurun_out/w2g_pmcA -o pmc -- python tools/exp/w2g_test.py > gpurun_out/w2g_pmcA.log 2>&1
python tools/pmc_by_grid.py $(find gpurun_out/w2g_pmcA -name "*.db" | head -1) wino2f_wgrad
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_LDS_ADDR_CONFLICT -d gpurun_out/w2g_pmcB -o pmc -- python tools/exp/w2g_test.py > gpurun_out/w2g_pmcB.log 2>&1
python tools/pmc_by_grid.py $(find gpurun_out/w2g_pmcB -name "*.db" | head -1) wino2f_wgrad
rm -rf gpurun_out/w2g_kt gpurun_out/w2g_pmcA gpurun_out/w2g_pmcB
