cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rocprofv3 --kernel-trace -d gpurun_out/r02_tl -o kt -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-warm --no-roofline > gpurun_out/r02_tl.log 2>&1
DB=$(find gpurun_out/r02_tl -name "*.db" | head -1)
python tools/exp/fwd_gaps.py $DB 8
python tools/timeline.py $DB 8 | head -3
rm -rf gpurun_out/r02_tl
