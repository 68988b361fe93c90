"""In-kernel timeline of the fused F(2x2) kernel (csrc/wino2f.hip built with -DW2F_TRACE): s_memtime stamps of workgroup 0,
third work item, per wave: item start, per half (arrive at barrier, leave barrier, end of half), epilogue end."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
from wino2f_test import L, filt, timeit
N, H, W = 32, 128, 128
x = torch.randn(N, H, W, 64, device="cuda")
w = torch.randn(64, 3, 3, 64, device="cuda") * 0.06
u = filt(w, 0)
y = torch.empty(N, H, W, 64, device="cuda")
s = torch.cuda.current_stream().cuda_stream
rows = ctypes.c_int(0)
f = lambda: L.denet_conv_wino2f(x.data_ptr(), u.data_ptr(), None, None, y.data_ptr(), 0, None, 0, ctypes.byref(rows), N, H, W, 64, 64, s)
print("%.1f us" % timeit(f, 20))
f(); torch.cuda.synchronize()
buf = (ctypes.c_uint * 256)()
L.denet_w2f_trace.restype = ctypes.c_int
assert L.denet_w2f_trace(buf) == 0
full = np.array(buf[:], dtype=np.int64).reshape(8, 32)
print("whole kernel, wave 0: %d ticks" % ((full[0, 27] - full[0, 26]) & 0xFFFFFFFF))
t = full[:, :26]
t0 = t[:, 0].min()
t = (t - t0) & 0xFFFFFFFF
names = ["start"] + ["%s%d%s" % (p, g, h) for g in range(4) for h in "AB" for p in ("arr", "lv", "end")] + ["epi"]
for wv in range(8):
    print("wave %d: " % wv + " ".join("%s=%d" % (n, v) for n, v in zip(names, t[wv])))
d = t[:, 1:].astype(float)
print("per wave-mean: item %.0f ticks; barrier wait (lv-arr) per half: %s" % ((t[:, 25] - t[:, 0]).mean(),
      " ".join("%.0f" % (t[:, 2 + 3 * h] - t[:, 1 + 3 * h]).mean() for h in range(8))))
print("half duration (end - prev end): %s" % " ".join("%.0f" % ((t[:, 3 + 3 * h] - (t[:, 3 * h] if h else t[:, 0])).mean()) for h in range(8)))
print("pre-barrier part (arr - prev end): %s" % " ".join("%.0f" % ((t[:, 1 + 3 * h] - (t[:, 3 * h] if h else t[:, 0])).mean()) for h in range(8)))
print("post-barrier part (end - lv): %s" % " ".join("%.0f" % ((t[:, 3 + 3 * h] - t[:, 2 + 3 * h]).mean()) for h in range(8)))
print("epilogue: %.0f" % (t[:, 25] - t[:, 24]).mean())
