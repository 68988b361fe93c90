for t in 1 2; do for n in 1 2; do echo "== tile $t nbuf $n"; DENET_IGEMM_TILE=$t DENET_IGEMM_NBUF=$n python tools/exp/gemm_shapes.py 2>&1 | grep -v amdgpu.ids; done; done
