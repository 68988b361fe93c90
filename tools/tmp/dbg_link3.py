import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from denet_amd import ops
g = torch.Generator().manual_seed(3)
N, H, C, K, tile = 1, 8, 32, 32, 4
x = torch.randn(N, H, H, C, generator=g).cuda()
w = (torch.randn(K, 3, 3, C, generator=g) * 0.05).cuda()
gamma = (torch.rand(C, generator=g) + 0.5).cuda(); beta = torch.randn(C, generator=g).cuda()
rm = torch.zeros(C).cuda(); rs = torch.ones(C).cuda()
geom = ops.conv_geom(x.shape, w.shape, 1, 1, 3)
ops._WINO[(0, geom)] = tile; ops._WINO[(2, geom)] = tile
y_ref, sm, si = ops.bn_fwd_train(x, gamma, beta, rm.clone(), rs.clone(), relu=True)
c1 = {"train": True}; ops.conv_fwd(y_ref, w, stride=1, pad=1, s_real=3, cache=c1)
link = ops.BnLink(False, x, None, None, gamma, beta, sm, si, None, True)
c2 = {"train": True}; ops.conv_fwd(None, w, stride=1, pad=1, s_real=3, cache=c2, link=link)
torch.cuda.synchronize()
T = N * (H // 4) * (H // 4)
V1 = c1["V"].view(36, T, C); V2 = c2["V"].view(36, T, C)
d = (V1 != V2)
print("differing elements", int(d.sum()), "of", d.numel())
idx = d.nonzero()[:12].tolist()
for xi, t, c in idx:
    print("xi", xi, "(i,j)=", divmod(xi, 6), "tile", t, "c", c, V1[xi, t, c].item(), V2[xi, t, c].item())
print("components with differences:", sorted(set(int(v) for v in d.nonzero()[:, 0].tolist())))
print("tiles with differences:", sorted(set(int(v) for v in d.nonzero()[:, 1].tolist())))
