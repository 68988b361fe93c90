"""Fused F(2x2) filter-gradient kernel (csrc/wino2f.hip): correctness against fp64 and time beside the current wgrad path"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import torch.nn.functional as Fn
from denet_amd import lib, ops
from wino2f_test import timeit
L = lib.load()


def run(x, dy):
    N, H, W, _ = x.shape
    nb = L.denet_conv_wino2f_wgrad_workspace_bytes(N, H, W)
    ws = torch.zeros(nb // 4, device="cuda")
    dw = torch.empty(64, 3, 3, 64, device="cuda")
    rc = L.denet_conv_wino2f_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), nb, N, H, W, 64, 64,
                                   torch.cuda.current_stream().cuda_stream)
    assert rc == 0, lib.last_error()
    return dw


def check(N, H, W, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, H, W, 64, generator=g).cuda()
    dy = torch.randn(N, H, W, 64, generator=g).cuda()
    w = torch.zeros(64, 64, 3, 3, dtype=torch.float64, device="cuda", requires_grad=True)
    y = Fn.conv2d(x.double().permute(0, 3, 1, 2), w, None, padding=1)
    ref = torch.autograd.grad(y, w, dy.double().permute(0, 3, 1, 2))[0].permute(0, 2, 3, 1)     # [k][r][s][c]
    dw = run(x, dy)
    e = float((dw.double() - ref).abs().max() / ref.abs().max())
    print("N %d H %d W %d: wgrad max-norm err %.2e" % (N, H, W, e), flush=True)
    assert e < 2e-5, e


if __name__ == "__main__":
    check(1, 16, 16)
    check(2, 32, 48, 1)
    check(3, 64, 64, 2)
    check(5, 128, 128, 3)
    N, H, W = 32, 128, 128
    x = torch.randn(N, H, W, 64, device="cuda")
    dy = torch.randn(N, H, W, 64, device="cuda")
    nb = L.denet_conv_wino2f_wgrad_workspace_bytes(N, H, W)
    ws = torch.zeros(nb // 4, device="cuda")
    dw = torch.empty(64, 3, 3, 64, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    t = timeit(lambda: L.denet_conv_wino2f_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), nb, N, H, W, 64, 64, s))
    print("fused F2 l1 wgrad %7.1f us" % t)
    for tuned, name in ((False, "direct"), (True, "tuned")):
        ops.AUTOTUNE = tuned
        ops._WINO.clear(); ops._TUNED.clear()
        for _ in range(2):
            ops.conv_wgrad(x, dy, (64, 3, 3, 64), stride=1, pad=1, s_real=3)
        t = timeit(lambda: ops.conv_wgrad(x, dy, (64, 3, 3, 64), stride=1, pad=1, s_real=3))
        print("%-6s l1 wgrad %7.1f us (winograd tile %s)" % (name, t, ops._WINO.get((2, ops.conv_geom(x.shape, (64, 3, 3, 64), 1, 1, 3)))))
