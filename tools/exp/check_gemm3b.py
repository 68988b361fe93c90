"""3-term bf16-split GEMM (csrc/gemm3b.hip, opt-in): error against fp64 next to the exact-fp32 kernel's, and time, on the four
1x1 convolutions of the detection head (B=32: M = 18432)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from denet_amd import lib, ops
from wino2f_test import timeit
L = lib.load()
M = int(os.environ.get("M", 18432))
for K, N in ((4736, 1536), (1536, 1024), (1024, 768), (768, 512)):
    g = torch.Generator().manual_seed(K)
    a = torch.randn(M, K, generator=g).cuda()
    b = (torch.randn(N, K, generator=g) * (2.0 / K) ** 0.5).cuda()
    bias = torch.randn(N, generator=g).cuda()
    c = torch.empty(M, N, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    f = lambda: L.denet_gemm_bf16x3_nt(a.data_ptr(), b.data_ptr(), bias.data_ptr(), c.data_ptr(), M, N, K, s)
    assert f() == 0, lib.last_error()
    rows = slice(0, 512)
    ref = a[rows].double() @ b.double().T + bias.double()
    e3 = float((c[rows].double() - ref).abs().max() / ref.abs().max())
    x4 = a.view(M // 576, 24, 24, K) if M % 576 == 0 else a.view(1, 1, M, K)
    w4 = b.view(N, 1, 1, K)
    y = ops.conv_fwd(x4, w4, bias=bias, stride=1, pad=0)
    y = ops.conv_fwd(x4, w4, bias=bias, stride=1, pad=0)
    e1 = float((y.reshape(M, N)[rows].double() - ref).abs().max() / ref.abs().max())
    t3 = timeit(f, 10)
    t1 = timeit(lambda: ops.conv_fwd(x4, w4, bias=bias, stride=1, pad=0), 10)
    flop = 2.0 * M * N * K
    print("K %4d N %4d: bf16x3 %7.1f us (%.0f TF fp32-equivalent) max-norm err %.2e | exact fp32 %7.1f us (%.0f TF) err %.2e" % (
        K, N, t3, flop / t3 / 1e6, e3, t1, flop / t1 / 1e6, e1), flush=True)
