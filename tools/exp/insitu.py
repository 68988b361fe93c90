"""GEMM timing with the A operand freshly written by another kernel before every launch (as in the Winograd pass, where
the input transform writes V right before the component GEMMs), igemm vs bgemm, event-timed per launch."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from denet_amd import lib
L = lib.load()
fb = getattr(L, "_Z11denet_bgemmPKfS0_PfiiiilllPvmiiP12ihipStream_t")
fb.restype = ctypes.c_int
fb.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 4 + [ctypes.c_long] * 3 + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
fi = getattr(L, "_Z21denet_gemm_batched_ntPKfS0_PfiiiilllPvmP12ihipStream_t")
fi.restype = ctypes.c_int
fi.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 4 + [ctypes.c_long] * 3 + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
wsb = getattr(L, "_Z27denet_bgemm_workspace_bytesv"); wsb.restype = ctypes.c_size_t
nws = wsb()
ws = torch.zeros(nws, dtype=torch.uint8, device="cuda")
SHAPES = [("l2", 8192, 128, 128), ("l3", 2048, 256, 256), ("l4", 512, 512, 512)]
NX = 36
s = torch.cuda.current_stream().cuda_stream
for name, M, N, K in SHAPES:
    a = torch.randn(NX, M, K, device="cuda"); a2 = a.clone()
    b = torch.randn(NX, N, K, device="cuda") * 0.05
    c = torch.empty(NX, M, N, device="cuda")
    runs = {"igemm": lambda: fi(a.data_ptr(), b.data_ptr(), c.data_ptr(), NX, M, N, K, M * K, N * K, M * N, None, 0, s)}
    for tile in (0, 1):
        for wg in (1, 2):
            runs["bgemm t%d w%d" % (tile, wg)] = (lambda tile=tile, wg=wg: fb(a.data_ptr(), b.data_ptr(), c.data_ptr(), NX, M, N, K, M * K, N * K, M * N, ws.data_ptr(), nws, tile, wg, s))
    for label, run in runs.items():
        for mode in ("hot", "fresh"):
            ts = []
            for it in range(12):
                if mode == "fresh":
                    a.copy_(a2)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); assert run() == 0; e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            ts = sorted(ts[2:])
            print("%-4s %-12s %-5s median %7.1f us min %7.1f us" % (name, label, mode, ts[len(ts) // 2], ts[0]), flush=True)
