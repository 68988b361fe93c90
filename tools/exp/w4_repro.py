"""two calls of the fused F(4x4) kernel on the same inputs: where do they differ? (debugging aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from denet_amd import ops
ops.init_streams()
L = ops._L()
gen = torch.Generator().manual_seed(11)
PRESS = int(os.environ.get("PRESS", "1"))
side = torch.cuda.Stream()
big_a = torch.empty(1 << 27, device="cuda")
big_b = torch.empty(1 << 27, device="cuda")
for (N, H, W, C, K, tb) in [(16, 64, 64, 128, 128, 64), (16, 32, 32, 256, 256, 32), (8, 64, 64, 256, 128, 33)]:
    x = torch.randn(N, H, W, C, generator=gen).cuda()
    w = (torch.randn(K, 3, 3, C, generator=gen) * 0.03).cuda()
    u = ops.conv_wino_filter(w, 4, dgrad=False)
    L.denet_conv_wino4f_mode(0)
    ref = ops.conv_wino_fwd(x, w, tile=4, u=u).clone()
    L.denet_conv_wino4f_mode(tb)
    outs = []
    for r in range(4):
        if r and PRESS:
            with torch.cuda.stream(side):
                for _ in range(8):
                    big_b.copy_(big_a, non_blocking=True)
        y = ops.conv_wino_fwd(x, w, tile=4, u=u)
        torch.cuda.synchronize()
        outs.append(y.clone())
    for r in range(4):
        d = (outs[r] != outs[0])
        e = (outs[r] - ref).abs()
        print(tb, r, "nan", int(torch.isnan(outs[r]).sum()), "differs from call 0:", int(d.sum()), "max err vs unfused %.3e" % float(e.max() / ref.abs().max()))
        if d.any():
            idx = d.nonzero()
            good, bad = outs[0], outs[r]
            for q in idx[:6].tolist() + idx[-3:].tolist():
                n_, y_, x_, k_ = q
                got = float(bad[n_, y_, x_, k_]); exp = float(good[n_, y_, x_, k_])
                ty, tx = y_ // 4, x_ // 4
                blk = good[n_, 4 * ty:4 * ty + 4, 4 * tx:4 * tx + 4, k_]
                where = (blk == got).nonzero().tolist()
                anyw = (good[n_, :, :, k_] == got).nonzero().tolist()[:3]
                print("   ", q, "got %.6f expected %.6f" % (got, exp), "same value inside the tile at", where, "elsewhere in the image/channel at", anyw)
            print("  first differing (n,y,x,k):", idx[0].tolist(), "last:", idx[-1].tolist(), "rows%4:", sorted(set((idx[:, 1] % 4).tolist())), "k/16:", sorted(set((idx[:, 3] // 16).tolist()))[:16],
                  "cols%4:", sorted(set((idx[:, 2] % 4).tolist())), "n:", sorted(set(idx[:, 0].tolist()))[:8], "maxdiff %.3e" % float((outs[r] - outs[0]).abs().max()))
L.denet_conv_wino4f_mode(-1)
