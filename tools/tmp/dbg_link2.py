import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from denet_amd import ops
g = torch.Generator().manual_seed(3)
for (N, H, C, K, tile, with_res) in [(2, 16, 128, 128, 4, False), (2, 16, 128, 128, 4, True), (2, 8, 256, 256, 4, True), (2, 4, 512, 512, 4, False),
                                     (2, 32, 64, 64, 4, True), (2, 16, 128, 128, 2, True), (2, 64, 64, 64, 4, False)]:
    x = torch.randn(N, H, H, C, generator=g).cuda()
    res = torch.randn(N, H, H, C, generator=g).cuda() if with_res else None
    w = (torch.randn(K, 3, 3, C, generator=g) * 0.05).cuda()
    gamma = (torch.rand(C, generator=g) + 0.5).cuda(); beta = torch.randn(C, generator=g).cuda()
    rm = torch.zeros(C).cuda(); rs = torch.ones(C).cuda()
    geom = ops.conv_geom(x.shape, w.shape, 1, 1, 3)
    ops._WINO[(0, geom)] = tile; ops._WINO[(2, geom)] = 0
    ops._TUNED.add((0, geom)) if hasattr(ops._TUNED, "add") else None
    # reference: stats + apply, then the plain Winograd pass
    y_ref, sm, si = ops.bn_fwd_train(x, gamma, beta, rm.clone(), rs.clone(), relu=True, res=res)
    cache = {"train": True}
    out_ref = ops.conv_fwd(y_ref, w, stride=1, pad=1, s_real=3, cache=cache, bn_stats=True)
    st_ref = cache["bn_stats"]
    link = ops.BnLink(False, x, res, None, gamma, beta, sm, si, None, True)
    cache2 = {"train": True}
    out = ops.conv_fwd(None, w, stride=1, pad=1, s_real=3, cache=cache2, bn_stats=True, link=link)
    torch.cuda.synchronize()
    print((N, H, C, K, tile, with_res), "linked" if link.result is not y_ref and ops.LINK_COUNT[0] else "NOT LINKED", "act equal", torch.equal(link.result, y_ref), "conv equal", torch.equal(out, out_ref),
          "max diff act %.3g conv %.3g" % (float((link.result - y_ref).abs().max()), float((out - out_ref).abs().max())),
          "stats rows", st_ref[1] if st_ref else None, cache2["bn_stats"][1] if cache2.get("bn_stats") else None,
          "stats equal", (st_ref is None and cache2.get("bn_stats") is None) or torch.equal(st_ref[0][:st_ref[1] * 2 * K], cache2["bn_stats"][0][:st_ref[1] * 2 * K]))
