cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 -L > $R/gpurun_out/r02_counters.txt 2>&1
cd $R
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d gpurun_out/r02_pmcA -o pmc -- python tools/bench_wino_gemm.py > gpurun_out/r02_pmcA.log 2>&1
python tools/pmc_summary.py $(ls gpurun_out/r02_pmcA/*/*.db | head -1) > gpurun_out/r02_pmcA_summary.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_MISC -d gpurun_out/r02_pmcB -o pmc -- python tools/bench_wino_gemm.py > gpurun_out/r02_pmcB.log 2>&1
python tools/pmc_summary.py $(ls gpurun_out/r02_pmcB/*/*.db | head -1) > gpurun_out/r02_pmcB_summary.txt 2>&1
rm -rf gpurun_out/r02_pmcA/*/*.db gpurun_out/r02_pmcB/*/*.db
tail -30 gpurun_out/r02_pmcA_summary.txt
