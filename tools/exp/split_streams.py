"""Does a component GEMM run faster as two concurrent half-batches on two streams than as one launch?"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from denet_amd import lib
L = lib.load()
fi = getattr(L, "_Z21denet_gemm_batched_ntPKfS0_PfiiiilllPvmP12ihipStream_t")
fi.restype = ctypes.c_int
fi.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 4 + [ctypes.c_long] * 3 + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
s0 = torch.cuda.current_stream()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
NX = 36
for name, M, N, K in [("l2", 8192, 128, 128), ("l3", 2048, 256, 256), ("l4", 512, 512, 512), ("up1f", 2048, 256, 512)]:
    a = torch.randn(NX, M, K, device="cuda"); b = torch.randn(NX, N, K, device="cuda") * 0.05
    c = torch.empty(NX, M, N, device="cuda")
    def one():
        fi(a.data_ptr(), b.data_ptr(), c.data_ptr(), NX, M, N, K, M * K, N * K, M * N, None, 0, s0.cuda_stream)
    def two(parts=2):
        ev = torch.cuda.Event(); ev.record(s0)
        h = NX // parts
        for i, st in enumerate([s1, s2][:parts]):
            st.wait_event(ev)
            fi(a[i * h].data_ptr(), b[i * h].data_ptr(), c[i * h].data_ptr(), h, M, N, K, M * K, N * K, M * N, None, 0, st.cuda_stream)
            e2 = torch.cuda.Event(); e2.record(st); s0.wait_event(e2)
    for label, fn in (("one launch", one), ("two streams", two)):
        ts = []
        for it in range(14):
            a.mul_(1.0)        # a memory-bound kernel in front, like the input transform
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s0); fn(); e1.record(s0)
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts = sorted(ts[2:])
        print("%-5s %-12s median %7.1f us min %7.1f" % (name, label, ts[len(ts) // 2], ts[0]), flush=True)
