import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools/exp")
import torch
from wino2f_test import *
N, H, W = 32, 128, 128
x = torch.randn(N, H, W, 64, device="cuda")
w = torch.randn(64, 3, 3, 64, device="cuda") * 0.06
u = filt(w, 0)
y = torch.empty(N, H, W, 64, device="cuda")
s = torch.cuda.current_stream().cuda_stream
import ctypes
rows = ctypes.c_int(0)
f = lambda: L.denet_conv_wino2f(x.data_ptr(), u.data_ptr(), None, None, y.data_ptr(), 0, None, 0, ctypes.byref(rows), N, H, W, 64, 64, s)
print("DBG", os.environ.get("DENET_W2F_DBG"), "%.1f us" % timeit(f, 50))
