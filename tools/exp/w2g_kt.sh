R=$GRAFT_REPO_ROOT
cd $R; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d gpurun_out/w2g_kt -o kt -- python tools/exp/w2g_time.py > gpurun_out/w2g_kt.log 2>&1
python tools/kt_by_grid.py $(find gpurun_out/w2g_kt -name "*.db" | head -1) w2g
python tools/kt_by_grid.py $(find gpurun_out/w2g_kt -name "*.db" | head -1) wino2f_wgrad
rm -rf gpurun_out/w2g_kt
