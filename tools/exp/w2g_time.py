import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from denet_amd import lib
from wino2f_test import timeit
L = lib.load()
N, H, W = 32, 128, 128
x = torch.randn(N, H, W, 64, device="cuda")
dy = torch.randn(N, H, W, 64, device="cuda")
nb = L.denet_conv_wino2f_wgrad_workspace_bytes(N, H, W)
ws = torch.zeros(nb // 4, device="cuda")
dw = torch.empty(64, 3, 3, 64, device="cuda")
s = torch.cuda.current_stream().cuda_stream
t = timeit(lambda: L.denet_conv_wino2f_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), nb, N, H, W, 64, 64, s), 50)
print("fused F2 l1 wgrad (3 kernels) %7.1f us" % t)
