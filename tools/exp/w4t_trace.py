"""s_memtime stamps of wino4t workgroups (build: bash tools/exp/w4t_variants.sh trace:-DT_TRACE; run with
W4T_LIB=tools/exp/_w4t/libdenet_hip_trace.so): per phase of a workgroup the time in s_memtime TICKS, for every 64th workgroup of one
launch on the 64-channel stage's geometry. The counter's rate is not the 100 MHz of older chips: the launch of 2026-09-30 (145 us,
four items of a workgroup slot after one another, ~580-750 ticks each) fits ~21 ticks per microsecond; read the columns as ratios.
Start values of different XCDs are not comparable (a counter per XCD)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import denet_amd.lib as _lib
_lib.LIB_PATH = os.path.abspath(os.environ["W4T_LIB"])
from denet_amd import ops
from denet_amd.lib import load, ptr, stream_ptr, check

L = load()
dbgfn = ctypes.CDLL(_lib.LIB_PATH).denet_conv_wino4t_debug
dbgfn.argtypes = [ctypes.c_void_p]
B, H, W, C, K = 32, 128, 128, 64, 64
x = torch.randn(B, H, W, C, device="cuda")
w = torch.randn(K, 3, 3, C, device="cuda") * 0.05
u = ops.conv_wino_filter(w, 4, dgrad=False)
pk = torch.empty_like(u)
check(L.denet_conv_wino4t_pack(ptr(u), ptr(pk), C, K, stream_ptr()), "pack")
y = torch.empty(B, H, W, K, device="cuda")
st = torch.zeros(L.denet_conv_wino4t_stats_rows(B, H, W) * 2 * K, dtype=torch.float64, device="cuda")
rows = ctypes.c_int(0)
run = lambda: check(L.denet_conv_wino4t_sums(ptr(x), ptr(pk), None, None, ptr(y), 0, ptr(st), st.numel() * 8, ctypes.byref(rows), None,
                                             B, H, W, C, K, stream_ptr()), "w4t")
for _ in range(3):
    run()
torch.cuda.synchronize()
nwg = B * (H // 8) * (W // 32)
dbg = torch.zeros((nwg // 64) * 32, dtype=torch.int64, device="cuda")
dbgfn(dbg.data_ptr())
run()
torch.cuda.synchronize()
dbgfn(None)
t = dbg.cpu().view(-1, 32).numpy()
t0 = t[:, 0].min()
names = ["start"] + sum([["dma%d" % s, "bar", "transf", "bar"] for s in range(4)], []) + ["mfma3", "outT", "stores", "stats"]
print("stamps per workgroup: start, then per chunk (pieces landed | barrier | transform done | barrier), end of products, output transform, stores, statistics")
print("columns (hundreds of ticks): per chunk [products of the chunk before +] wait for the pieces, barrier, transform, barrier; the last chunk's products; output transform, stores, statistics")
for i, r in enumerate(t):
    n = int((r > 0).sum())
    us = [(v - t0) / 100.0 for v in r[:n]]          # hundreds of ticks
    d = [us[k + 1] - us[k] for k in range(n - 1)]
    # per chunk: stamps 1+4s .. 4+4s; the products of chunk s lie between stamp 4+4s and stamp 5+4s (next chunk's first stamp / end)
    parts = []
    for s in range(4):
        b = 4 * s          # d[b]: wait for the chunk's pieces (s > 0: + the products of the chunk before), then barrier, transform, barrier
        parts.append("%5.1f %4.1f %4.1f %4.1f" % (d[b], d[b + 1], d[b + 2], d[b + 3]))
    print("wg %4d | %s | last products %5.1f | outT %4.1f stores %5.1f stats %4.1f | total %5.1f" % (
        64 * i, " | ".join(parts), d[16], d[17], d[18], d[19] if len(d) > 19 else 0.0, us[-1] - us[0]))
