"""Runs one conv shape repeatedly (for rocprofv3 PMC passes). usage: conv_one.py mode H W C K R stride pad [B]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from denet_amd import ops
mode = sys.argv[1]
H, W, C, K, R, st, pad = [int(v) for v in sys.argv[2:9]]
B = int(sys.argv[9]) if len(sys.argv) > 9 else 32
x = torch.randn(B, H, W, C, device="cuda")
w = torch.randn(K, R, R, C, device="cuda") * 0.05
OH = (H + 2 * pad - R) // st + 1
dy = torch.randn(B, OH, OH, K, device="cuda")
for _ in range(5):
    if mode == "fwd": ops.conv_fwd(x, w, stride=st, pad=pad)
    elif mode == "dgrad": ops.conv_dgrad(dy, w, tuple(x.shape), stride=st, pad=pad)
    else: ops.conv_wgrad(x, dy, tuple(w.shape), stride=st, pad=pad)
torch.cuda.synchronize()
