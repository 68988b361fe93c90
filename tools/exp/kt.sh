cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for sh in l2 l3 l4 up1 up2; do
for v in 0 1; do
SHAPE=$sh DENET_BGEMM=$v rocprofv3 --kernel-trace -d gpurun_out/r02_kt$v -o kt -- python tools/bench_wino_gemm.py > gpurun_out/r02_kt$v.log 2>&1
echo "== $sh bgemm=$v: $(grep fwd gpurun_out/r02_kt$v.log)"
python tools/kt_by_grid.py $(find gpurun_out/r02_kt$v -name "*.db" | head -1) GLOBAL | grep -v "filter" 
rm -rf gpurun_out/r02_kt$v
done; done
