"""the RoI gather's backward pass (denet_sparse_bwd with presorted taps) at the step's size: 32 images, 64 x 64 cells, 576 RoIs,
7 x 7 taps, 96 features; time alone and a checksum of the result's bits (the summation order per cell is part of the contract)"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from denet_amd import ops

B, H, W, F, rois, gs = 32, 64, 64, 96, 576, 7
CP, coff = 128, 0
KP = ((gs * gs * F + 2 + 31) // 32) * 32
rng = np.random.RandomState(0)
for name in ("random boxes", "detector-like boxes (clustered)"):
    M = B * rois
    if name.startswith("random"):
        x0, y0 = rng.uniform(0, 1, M), rng.uniform(0, 1, M)
        bbox = np.stack([x0, y0, x0 + (1 - x0) * rng.uniform(0, 1, M), y0 + (1 - y0) * rng.uniform(0, 1, M)], 1).astype(np.float32)
    else:
        cx, cy = rng.normal(0.5, 0.08, M), rng.normal(0.5, 0.08, M)
        w, h = rng.uniform(0.02, 0.2, M), rng.uniform(0.02, 0.2, M)
        bbox = np.clip(np.stack([cx - w, cy - h, cx + w, cy + h], 1), 0, 1).astype(np.float32)
    fmap = torch.randn(B, H, W, CP, device="cuda")
    out, taps = ops.sparse_fwd(fmap, torch.from_numpy(bbox).cuda(), coff, F, rois, gs, KP, 0)
    dy = torch.randn(M, KP, device="cuda")
    dfmap = torch.zeros(B, H, W, CP, device="cuda")
    ev = ops.sparse_sort_async(taps, B, H, W, rois, gs)
    torch.cuda.synchronize()
    for _ in range(3):
        ops.sparse_bwd(dy, taps, dfmap, coff, F, rois, gs, coff + F, presorted=ev)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        ops.sparse_bwd(dy, taps, dfmap, coff, F, rois, gs, coff + F, presorted=ev)
    b.record()
    torch.cuda.synchronize()
    bits = dfmap.view(torch.int32).to(torch.int64)
    print("%s: %.0f us, checksum %d / %d" % (name, a.elapsed_time(b) / 20 * 1e3, int(bits.sum()), int((bits * (torch.arange(bits.numel(), device="cuda").view_as(bits) % 1009)).sum())))
