cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rocprofv3 --kernel-trace -d gpurun_out/r02_kt -o kt -- python bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-warm > gpurun_out/r02_kt.log 2>&1
DB=$(find gpurun_out/r02_kt -name "*.db" | head -1)
python tools/kt_by_grid.py $DB igemm_kernel
rm -rf gpurun_out/r02_kt
