import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
from denet_amd import ops
L = ops._L()
B, H, W, rois, gs = 32, 64, 64, 576, 7
n = rois * gs * gs
rng = np.random.RandomState(0)
# realistic taps: boxes -> 7x7 grids of cells
taps = np.zeros((B, rois, 49), np.int32)
for b in range(B):
    for r in range(rois):
        x0, y0 = rng.uniform(0, 0.8, 2); w, h = rng.uniform(0.05, 0.2, 2)
        xs = np.clip(np.rint((x0 + np.arange(7) * w / 6) * W), 0, W - 1).astype(int)
        ys = np.clip(np.rint((y0 + np.arange(7) * h / 6) * H), 0, H - 1).astype(int)
        taps[b, r] = (ys[:, None] * W + xs[None, :]).reshape(-1)
td = torch.from_numpy(taps.reshape(B, n)).cuda()
nb = L.denet_sparse_sort_workspace_bytes(B, H, W, rois, gs)
ws = torch.zeros(nb, dtype=torch.uint8, device="cuda")
run = lambda: ops.check(L.denet_sparse_sort(ops.ptr(td), ops.ptr(ws), nb, B, H, W, rois, gs, ops.stream_ptr()), "sort")
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
print("sparse_sort (3 kernels): %.1f us" % (e0.elapsed_time(e1) / 20 * 1e3))
