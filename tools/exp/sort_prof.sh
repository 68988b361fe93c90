cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python tools/exp/sort_time.py 2>&1 | grep -v amdgpu
rocprofv3 --kernel-trace -d gpurun_out/r02_sortkt -o kt -- python tools/exp/sort_time.py > /dev/null 2>&1
python tools/kt_by_grid.py $(find gpurun_out/r02_sortkt -name "*.db" | head -1) sparse
rm -rf gpurun_out/r02_sortkt
