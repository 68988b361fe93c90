"""Correctness + timing of the persistent stream-K batched GEMM (csrc/bgemm.hip) on the Winograd component shapes."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from denet_amd import lib
L = lib.load()
f = getattr(L, "_Z11denet_bgemmPKfS0_PfiiiilllPvmiiP12ihipStream_t")
f.restype = ctypes.c_int
f.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 4 + [ctypes.c_long] * 3 + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
wsb = getattr(L, "_Z27denet_bgemm_workspace_bytesv")
wsb.restype = ctypes.c_size_t
nws = wsb()
ws = torch.zeros(nws, dtype=torch.uint8, device="cuda")
ONLY = os.environ.get("SHAPES")
SHAPES = [("l1", 32768, 64, 64), ("l2", 8192, 128, 128), ("l3", 2048, 256, 256), ("l4", 512, 512, 512), ("up1f", 2048, 256, 512),
          ("up1d", 2048, 512, 256), ("up2f", 8192, 128, 256), ("up2d", 8192, 256, 128), ("odd", 1000, 96, 160), ("tiny", 70, 32, 32)]
NX = int(os.environ.get("NX", 36))
iters = int(os.environ.get("ITERS", 20))
s = torch.cuda.current_stream().cuda_stream
for name, M, N, K in SHAPES:
    if ONLY and name not in ONLY.split(','): continue
    a = torch.randn(NX, M, K, device="cuda")
    b = torch.randn(NX, N, K, device="cuda") * 0.05
    ref = torch.bmm(a.double(), b.double().transpose(1, 2))
    for tile in (0, 1):
        for wg in (1, 2):
            c = torch.full((NX, M, N), float("nan"), device="cuda")
            run = lambda: f(a.data_ptr(), b.data_ptr(), c.data_ptr(), NX, M, N, K, M * K, N * K, M * N, ws.data_ptr(), nws, tile, wg, s)
            rc = run()
            assert rc == 0, L.denet_last_error()
            torch.cuda.synchronize()
            err = float((c.double() - ref).abs().max() / ref.abs().max())
            flag_err = int(ws[16380:16384].view(torch.int32)[0])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / iters
            c2 = c.clone()
            run(); torch.cuda.synchronize()
            same = bool(torch.equal(c, c2))
            print("%-5s M %6d N %4d K %4d tile %d wg %d  %8.1f us %6.1f TF  err %.1e  timeout %d  deterministic %s" % (
                name, M, N, K, tile, wg, ms * 1e3, 2.0 * NX * M * N * K / ms / 1e9, err, flag_err, same), flush=True)
            assert os.environ.get('DENET_BGEMM_DBG') or (err < 5e-6 and flag_err == 0 and same)
