"""Calibration for a Winograd F(2x2,3x3) path: the 16 component GEMMs of a layer have the FLOPs / tile count of ONE
1x1 convolution over 16x the tiles. Times that GEMM with the existing fwd kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from denet_amd import ops
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
for name, HW, C, K, direct_ms in [("l2", 64, 128, 128, 0.30), ("l3", 32, 256, 256, 0.292), ("l4", 16, 512, 512, 0.305), ("up1", 32, 512, 256, 0.567), ("up2", 64, 256, 128, 0.58)]:
    NX, MO = (36, 4) if len(sys.argv) > 1 and sys.argv[1] == "4" else (16, 2)
    T = 32 * (HW // MO) * (HW // MO)
    x = torch.randn(1, NX, T, C, device="cuda"); w = torch.randn(K, 1, 1, C, device="cuda") * 0.05
    t = timeit(lambda: ops.conv_fwd(x, w))
    flop = 2.0 * NX * T * C * K
    v_mb = NX * T * C * 4 / 1e6; m_mb = NX * T * K * 4 / 1e6
    tr = (v_mb * 1.3 + m_mb * 1.25) / 5.0e3    # ms at 5 TB/s: input transform (read x + write V), output transform (read M + write y)
    print("%-4s GEMM %.3f ms %.1f TF | V %.0f MB M %.0f MB transforms ~%.3f ms | winograd ~%.3f ms vs direct %.3f ms" % (name, t, flop / t / 1e9, v_mb, m_mb, tr, t + tr, direct_ms))
