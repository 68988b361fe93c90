"""Times the batched component GEMMs of the Winograd passes (C_b[M,N] = A_b[M,K] B_b[N,K]^T, 36 members) through the
product's own launcher; DENET_IGEMM_TILE=1|2 (128x128 | 128x64) and DENET_IGEMM_NBUF=1|2 force the configuration."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from denet_amd import lib, ops
L = lib.load()
f = getattr(L, "_Z21denet_gemm_batched_ntPKfS0_PfiiiilllP12ihipStream_t")
f.restype = ctypes.c_int
f.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 4 + [ctypes.c_long] * 3 + [ctypes.c_void_p]
SHAPES = [("l1", 32768, 64, 64), ("l2", 8192, 128, 128), ("l3", 2048, 256, 256), ("l4", 512, 512, 512), ("up1f", 2048, 256, 512),
          ("up1d", 2048, 512, 256), ("up2f", 8192, 128, 256), ("up2d", 8192, 256, 128)]
NX = int(os.environ.get("NX", 36))
iters = int(os.environ.get("ITERS", 20))
for name, M, N, K in SHAPES:
    M = M * (36 // NX)
    a = torch.randn(NX, M, K, device="cuda")
    b = torch.randn(NX, N, K, device="cuda") * 0.05
    c = torch.empty(NX, M, N, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    run = lambda: f(a.data_ptr(), b.data_ptr(), c.data_ptr(), NX, M, N, K, M * K, N * K, M * N, s)
    assert run() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    flop = 2.0 * NX * M * N * K
    ref = (a[NX // 12, :64].double() @ b[NX // 12].double().T)
    err = float((c[NX // 12, :64].double() - ref).abs().max() / ref.abs().max())
    print("%-5s M %6d N %4d K %4d  %s  %8.1f us  %6.1f TF  bytes %.0f MB -> %.2f TB/s  err %.1e" % (
        name, M, N, K, ops._last_igemm_name(), ms * 1e3, flop / ms / 1e9, (a.numel() + c.numel()) * 4 / 1e6,
        (a.numel() + c.numel()) * 4 / ms / 1e9, err), flush=True)
