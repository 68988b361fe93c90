"""Does a memory-bound kernel overlap a matrix-bound one when they run on two streams? GEMM-shaped 1x1 convolutions (the
implicit-GEMM kernel of the head / the Winograd component products) on the compute stream, elementwise adds (3 HBM passes) on
a side stream: alone, alone, together."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from denet_amd import ops

torch.cuda.set_device(0)
ops.init_streams()
side = ops.side_stream(0)
main = torch.cuda.current_stream()
g = torch.Generator().manual_seed(1)

def run(label, M, C, K, n_gemm, elems, n_add):
    x = torch.randn(1, 1, M, C, generator=g).cuda()
    w = (torch.randn(K, 1, 1, C, generator=g) * 0.05).cuda()
    y = ops.empty(1, 1, M, K)
    a = torch.randn(elems, generator=g).cuda(); b = torch.randn(elems, generator=g).cuda(); c = torch.empty_like(a)
    def gemms():
        for _ in range(n_gemm):
            ops.conv_fwd(x, w, out=y)
    def adds():
        with torch.cuda.stream(side):
            for _ in range(n_add):
                ops.add(a, b, out=c)
    def timed(fns):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for f in fns:
            f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3
    for f in (gemms, adds):
        f()
    tg = min(timed([gemms]) for _ in range(3))
    ta = min(timed([adds]) for _ in range(3))
    tb = min(timed([adds, gemms]) for _ in range(3))
    fl = 2.0 * M * C * K * n_gemm
    print("%-28s gemm alone %.2f ms (%.0f TFLOP/s, %s) | adds alone %.2f ms (%.2f TB/s) | together %.2f ms | sum %.2f | overlap gain %.0f %%" % (
        label, tg, fl / tg / 1e9, ops._last_igemm_name(), ta, 3.0 * 4 * elems * n_add / ta / 1e9, tb, tg + ta, 100 * (tg + ta - tb) / min(tg, ta)))

run("head1 18432x4736x1536", 18432, 4736, 1536, 4, 32 << 20, 100)
run("comp-like 294912x128x128", 294912, 128, 128, 40, 32 << 20, 55)
run("comp-like 73728x256x256", 73728, 256, 256, 40, 32 << 20, 50)
run("comp-like 18432x512x512", 18432, 512, 512, 40, 32 << 20, 50)
