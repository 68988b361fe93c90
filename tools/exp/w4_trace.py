"""in-kernel s_memtime stamps of workgroup 0 of the fused F(4x4) kernel (debug build of csrc/wino4f.hip)"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from denet_amd import ops
from denet_amd.lib import load
L = load()
lib = ctypes.CDLL(os.path.join(os.path.dirname(ops.__file__), "csrc", "libdenet_hip.so"))
B, H, W, C, K = 32, 64, 64, 128, 128
x = torch.randn(B, H, W, C, device="cuda"); w = torch.randn(K, 3, 3, C, device="cuda") * 0.05
ud = ops.conv_wino_filter(w, 4, dgrad=True)
dy = torch.randn(B, H, W, K, device="cuda")
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 64
L.denet_conv_wino4f_mode(mode)
for _ in range(3): ops.conv_wino_dgrad(dy, w, tile=4, u=ud)
torch.cuda.synchronize()
dbg = torch.zeros(16 * 640, dtype=torch.int64, device="cuda")
lib.denet_conv_wino4f_debug(ctypes.c_void_p(dbg.data_ptr()))
ops.conv_wino_dgrad(dy, w, tile=4, u=ud)
torch.cuda.synchronize()
lib.denet_conv_wino4f_debug(ctypes.c_void_p(0))
d = dbg.cpu().view(16, 640)
for wv in (0, 1, 5, 15 if mode == 64 else 7):
    r = d[wv]; n = int((r != 0).sum())
    st = r[:n].tolist()
    t0 = st[0]
    # triples per chunk: before wait, after wait, after barrier
    ch = [(st[i] - t0, st[i + 1] - st[i], st[i + 2] - st[i + 1]) for i in range(0, n - 2, 3)]
    per = [ch[i + 1][0] - ch[i][0] for i in range(len(ch) - 1)]
    print("wave", wv, "stamps", n, "total", st[-1] - st[0], "(100 MHz ticks?)")
    print("  chunk period:", per[:12], "... mean %.1f" % (sum(per) / len(per)))
    print("  vmcnt wait:", [c[1] for c in ch[:12]], "mean %.1f" % (sum(c[1] for c in ch) / len(ch)))
    print("  barrier wait:", [c[2] for c in ch[:12]], "mean %.1f" % (sum(c[2] for c in ch) / len(ch)))
    print("  tail:", [st[i] - st[i - 1] for i in range(n - 2, n)])
