"""fused 64-channel data-gradient pass with add + backward sums at several shapes against the direct kernel + the reduction pass"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from denet_amd import ops

g = torch.Generator().manual_seed(5)
for (N, H, W) in [(2, 16, 16), (2, 32, 32), (3, 48, 32), (32, 128, 128)]:
    for inplace in (False, True):
        for with_y in (False, True):
            C = K = 64
            dy = torch.randn(N, H, W, K, generator=g).cuda()
            w = (torch.randn(K, 3, 3, C, generator=g) * 0.05).cuda()
            addt = torch.randn(N, H, W, C, generator=g).cuda()
            x = (torch.randn(N, H, W, C, generator=g) * 1.5 + 0.3).cuda()
            gamma, beta = (torch.rand(C, generator=g) + 0.5).cuda(), torch.randn(C, generator=g).cuda()
            rm, rs = torch.zeros(C).cuda(), torch.ones(C).cuda()
            res = torch.randn(N, H, W, C, generator=g).cuda() if with_y else None
            y, sm, si = ops.bn_fwd_train(x, gamma, beta, rm, rs, relu=True, res=res)
            geom = ops.conv_geom((N, H, W, C), w.shape, 1, 1, None)
            ops.BWD_SUMS = 3
            ops._WINO[(1, geom)] = 0
            ref = ops.conv_dgrad(dy, w, (N, H, W, C), add=addt.clone(), stride=1, pad=1, cache={})
            ops._WINO[(1, geom)] = ops.FUSED2
            sums = ops.BnSums(x, y if with_y else None, gamma, beta, sm, si, True)
            a = addt.clone()
            dz = ops.conv_dgrad(dy, w, (N, H, W, C), add=a, stride=1, pad=1, cache={}, sums=sums, **({"out": a} if inplace else {}))
            torch.cuda.synchronize()
            e1 = float((dz - ref).abs().max()) / float(ref.abs().max())
            yy = y if with_y else None
            la, _ = ops.bn_bwd_link(x, yy, dz, gamma, sm, si, relu=True, beta=beta, pre=sums.partial)
            lb, _ = ops.bn_bwd_link(x, yy, dz, gamma, sm, si, relu=True, beta=beta)
            e2 = float((la.coef - lb.coef).abs().max()) / float(lb.coef.abs().max())
            print((N, H, W), "inplace" if inplace else "separate", "mask from y" if with_y else "recomputed", "dx err %.2e  coef err %.2e" % (e1, e2))
