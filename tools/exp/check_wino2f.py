"""Fused F(2x2,3x3) kernel (csrc/wino2f.hip): correctness against an fp64 convolution and time beside the un-fused Winograd
and direct passes at the l1 geometry (B=32, 128x128, 64->64)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as Fn
from denet_amd import lib, ops
L = lib.load()


def filt(w, dgrad):
    K, _, _, C = w.shape
    u = torch.empty(16, K * C, device="cuda")
    assert L.denet_conv_wino_filter(w.data_ptr(), u.data_ptr(), 2, dgrad, C, K, torch.cuda.current_stream().cuda_stream) == 0
    return u


def run(x, u, Co, bias=None, add=None, stats=False):
    N, H, W, Ci = x.shape
    y = torch.empty(N, H, W, Co, device="cuda")
    rows = ctypes.c_int(0)
    st = torch.zeros(N * (H // 16) * (W // 16) * 2 * Co, dtype=torch.float64, device="cuda") if stats else None
    rc = L.denet_conv_wino2f(x.data_ptr(), u.data_ptr(), bias.data_ptr() if bias is not None else None,
                             add.data_ptr() if add is not None else None, y.data_ptr(), 0, st.data_ptr() if stats else None,
                             st.numel() * 8 if stats else 0, ctypes.byref(rows), N, H, W, Ci, Co,
                             torch.cuda.current_stream().cuda_stream)
    assert rc == 0, lib.last_error()
    return y, (st.view(rows.value, 2, Co).sum(0) if stats else None)


def check(N, H, W, Co, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, H, W, 64, generator=g).cuda()
    w = (torch.randn(Co, 3, 3, 64, generator=g) * (2.0 / 576) ** 0.5).cuda()
    bias = torch.randn(Co, generator=g).cuda()
    add = torch.randn(N, H, W, Co, generator=g).cuda()
    ref = Fn.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), None, padding=1).permute(0, 2, 3, 1)
    y, _ = run(x, filt(w, 0), Co)
    e0 = float((y.double() - ref).abs().max() / ref.abs().max())
    y2, st = run(x, filt(w, 0), Co, bias, add, True)
    ref2 = ref + bias.double() + add.double()
    e1 = float((y2.double() - ref2).abs().max() / ref2.abs().max())
    r2 = ref2.reshape(-1, Co)
    es = float((st[0] - r2.sum(0)).abs().max() / r2.abs().sum(0).max())
    eq = float((st[1] - (r2 * r2).sum(0)).abs().max() / (r2 * r2).sum(0).max())
    # data gradient: dy [N,H,W,Co] -> dx [N,H,W,64] needs the output side to be 64 channels: K = 64 filters of C = Co
    e2 = -1.0
    if Co == 64:
        dy = torch.randn(N, H, W, 64, generator=g).cuda()
        refdx = torch.autograd.grad(Fn.conv2d(x.double().permute(0, 3, 1, 2).requires_grad_(True), w.double().permute(0, 3, 1, 2), None, padding=1),
                                    [], None, allow_unused=True) if False else None
        xx = x.double().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
        yy = Fn.conv2d(xx, w.double().permute(0, 3, 1, 2), None, padding=1)
        refdx = torch.autograd.grad(yy, xx, dy.double().permute(0, 3, 1, 2))[0].permute(0, 2, 3, 1)
        dx, _ = run(dy, filt(w, 1), 64)
        e2 = float((dx.double() - refdx).abs().max() / refdx.abs().max())
    print("N %d H %d W %d Co %d: fwd %.2e  fwd+bias+add %.2e  sums %.1e %.1e  dgrad %.2e" % (N, H, W, Co, e0, e1, es, eq, e2), flush=True)
    assert max(e0, e1, e2) < 2e-5 and es < 1e-5 and eq < 1e-5


def timeit(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


if __name__ == "__main__":
    check(1, 16, 16, 64)
    check(2, 32, 48, 64, 1)
    check(1, 16, 32, 128, 2)
    check(3, 64, 64, 64, 3)
    check(5, 128, 128, 64, 4)          # 320 work items: the persistent loop, uneven
    check(3, 128, 64, 128, 5)         # 192 blocks x 2 channel chunks
    N, H, W = 32, 128, 128
    x = torch.randn(N, H, W, 64, device="cuda")
    w = torch.randn(64, 3, 3, 64, device="cuda") * 0.06
    u = filt(w, 0)
    add = torch.randn(N, H, W, 64, device="cuda")
    t = timeit(lambda: run(x, u, 64))
    flop = 2.0 * N * H * W * 64 * 64 * 9
    print("fused F2 l1 fwd        %7.1f us  (%.1f direct-equivalent TF, %.2f TB/s of x+y)" % (t, flop / t / 1e6, 2 * x.numel() * 4 / t / 1e6))
    t = timeit(lambda: run(x, u, 64, None, add, True))
    print("fused F2 l1 +add+stats %7.1f us" % t)
    for tuned, name in ((False, "direct"), (True, "tuned")):
        ops.AUTOTUNE = tuned
        ops._WINO.clear(); ops._TUNED.clear()
        for _ in range(2):
            ops.conv_fwd(x, w, stride=1, pad=1, s_real=3)
        t = timeit(lambda: ops.conv_fwd(x, w, stride=1, pad=1, s_real=3))
        print("%-6s l1 fwd %7.1f us" % (name, t))
