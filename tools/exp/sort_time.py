"""alone time of the tap sort (denet_sparse_sort) at config 5's and the headline's sizes"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import os
from denet_amd import lib as dlib
if os.environ.get("OLD_LIB"):
    dlib.LIB_PATH = os.environ["OLD_LIB"]
from denet_amd import ops
L = ops._L()
for case in [(16, 128, 128, 2304, 7), (32, 64, 64, 576, 7), (32, 128, 128, 576, 7)]:
    B, H, W, rois, gs = case
    n = rois * gs * gs
    rng = np.random.RandomState(1)
    td = torch.from_numpy(rng.randint(0, H * W, (B, n)).astype(np.int32)).cuda()
    nbytes = L.denet_sparse_sort_workspace_bytes(B, H, W, rois, gs)
    ws = torch.zeros(nbytes // 4, dtype=torch.int32, device="cuda")
    for _ in range(3):
        ops.check(L.denet_sparse_sort(ops.ptr(td), ops.ptr(ws), nbytes, B, H, W, rois, gs, ops.stream_ptr()), "sort")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.check(L.denet_sparse_sort(ops.ptr(td), ops.ptr(ws), nbytes, B, H, W, rois, gs, ops.stream_ptr()), "sort")
    e1.record()
    torch.cuda.synchronize()
    print(case, "single" if L.denet_sparse_sort_is_single(B, H, W, rois, gs) else "three", "%.1f us" % (e0.elapsed_time(e1) * 1e3 / 20))
