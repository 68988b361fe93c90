"""the F(4x4) filter gradient with the slice sum folded into the adjoint filter transform: bit-identical to the build before it
(tools/exp/_base_lib.so = that build) and timed, at the headline's geometries"""
import ctypes
import sys

import torch

sys.path.insert(0, ".")
from denet_amd import lib as dlib
from denet_amd import ops

L = ops._L()
old = ctypes.CDLL("tools/exp/_base_lib.so")
sig = dlib._SIGNATURES["denet_conv_wino_wgrad"] if hasattr(dlib, "_SIGNATURES") else None
fo = old.denet_conv_wino_wgrad
fn = L.denet_conv_wino_wgrad
fo.restype, fo.argtypes = fn.restype, fn.argtypes
for (N, H, W, C, K) in [(32, 64, 64, 128, 128), (32, 32, 32, 256, 256), (32, 16, 16, 512, 512), (32, 64, 64, 256, 128), (16, 128, 128, 512, 256)]:
    torch.manual_seed(H + C)
    x = torch.randn(N, H, W, C, device="cuda")
    dy = torch.randn(N, H, W, K, device="cuda")
    ws = ops._wino_ws(4, N, H, W, C, K)
    sws = ops.WS.get("wgrad", ops.WGRAD_WS_BYTES)
    outs = []
    for f in (fn, fo):
        dw = torch.empty(K, 3, 3, C, device="cuda")
        for _ in range(2):
            rc = f(ops.ptr(x), ops.ptr(dy), None, ops.ptr(dw), ops.ptr(ws), ws.numel(), ops.ptr(sws), sws.numel(), 4, N, H, W, C, K, ops.stream_ptr())
            assert rc == 0, rc
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f(ops.ptr(x), ops.ptr(dy), None, ops.ptr(dw), ops.ptr(ws), ws.numel(), ops.ptr(sws), sws.numel(), 4, N, H, W, C, K, ops.stream_ptr())
        e1.record()
        torch.cuda.synchronize()
        outs.append((dw, e0.elapsed_time(e1) / 20 * 1e3))
    print((N, H, W, C, K), "splits", L.denet_wino4g_splits(4, N * (H // 4) * (W // 4), C, K) if hasattr(L, "denet_wino4g_splits") else "?",
          "identical:", torch.equal(outs[0][0], outs[1][0]), "new %.1f us  old %.1f us (transforms included)" % (outs[0][1], outs[1][1]))
