"""Soak test of the fused F(2x2) kernels: random batch sizes / image sizes, each result against the direct implicit-GEMM kernels,
while a second stream keeps HBM busy with large copies (stretches the LDS-DMA latencies the refill protocol has to cover)."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from denet_amd import ops
ops.init_streams()
rng = random.Random(5)
side = torch.cuda.Stream()
big_a = torch.empty(1 << 28, device="cuda")
big_b = torch.empty(1 << 28, device="cuda")
iters = int(os.environ.get("ITERS", 150))
worst = [0.0, 0.0, 0.0]
for it in range(iters):
    N = rng.choice([3, 8, 9, 17, 24, 32, 40, 48])
    H = rng.choice([64, 96, 128, 160])
    W = rng.choice([64, 128, 144])
    g = torch.Generator().manual_seed(it)
    x = torch.randn(N, H, W, 64, generator=g).cuda()
    dy = torch.randn(N, H, W, 64, generator=g).cuda()
    w = (torch.randn(64, 3, 3, 64, generator=g) * 0.06).cuda()
    geom = ops.conv_geom(x.shape, w.shape, 1, 1, None)
    res = {}
    for algo in (0, ops.FUSED2):
        ops._WINO[(0, geom)] = ops._WINO[(1, geom)] = ops._WINO[(2, geom)] = algo
        if algo and it % 2:
            with torch.cuda.stream(side):            # HBM traffic beside the fused kernels
                for _ in range(6):
                    big_b.copy_(big_a, non_blocking=True)
        y = ops.conv_fwd(x, w, stride=1, pad=1)
        dx = ops.conv_dgrad(dy, w, tuple(x.shape), stride=1, pad=1)
        dw = ops.conv_wgrad(x, dy, tuple(w.shape), stride=1, pad=1)
        res[algo] = (y, dx, dw)
    torch.cuda.synchronize()
    for k in range(3):
        a, b = res[0][k], res[ops.FUSED2][k]
        e = float((a - b).abs().max() / a.abs().max())
        worst[k] = max(worst[k], e)
        assert e < 1e-5, ("iteration %d N %d H %d W %d pass %d: %.3e" % (it, N, H, W, k, e))
print("%d iterations: worst relative difference to the direct kernels fwd %.2e dgrad %.2e wgrad %.2e" % (iters, *worst))
