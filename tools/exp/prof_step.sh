cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rocprofv3 --kernel-trace -d gpurun_out/r02_step -o kt -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r02_step.log 2>&1
DB=$(find gpurun_out/r02_step -name "*.db" | head -1)
python tools/rocpd_stats.py $DB --last-steps 5 > gpurun_out/r02_step_last5.md
python tools/rocpd_stats.py $DB --skip-steps 3 > gpurun_out/r02_step_timed.md
rm -rf gpurun_out/r02_step
head -60 gpurun_out/r02_step_last5.md | cut -c1-150
