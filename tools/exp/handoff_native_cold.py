"""the native part of the fast RoI hand-off with COLD caches (what a training step sees: the host has spun in a polling loop for ~19 ms):
between calls a 256 MB buffer is streamed through; variants: nothing / the generator stretch touched right before the call"""
import ctypes
import sys
import time

import numpy

sys.path.insert(0, ".")
from denet_amd import lib as dlib

L = dlib.load()
B, S, H, W = 32, 576, 64, 64
n_keep = S - int(0.25 * S)
rng = numpy.random.RandomState(1)
stream = rng.randint(0, 2 ** 32, 8 * B * S + 8192, dtype=numpy.uint64).astype(numpy.uint32)
cnt = numpy.where(rng.rand(B) < 0.8, S, 500).astype(numpy.int32)
x0 = rng.randint(0, W - 1, (B, S)); y0 = rng.randint(0, H - 1, (B, S))
box = numpy.ascontiguousarray(numpy.stack([x0, y0, numpy.minimum(W - 1, x0 + 5), numpy.minimum(H - 1, y0 + 5)], -1).astype(numpy.int32))
gt = rng.rand(B * 3, 4); off = (numpy.arange(B + 1) * 3).astype(numpy.int32)
ws = numpy.empty(2 * S, numpy.int32); out = numpy.empty((B, S, 4), numpy.float32)
cur, dry = ctypes.c_long(0), ctypes.c_int(0)
fn = L.denet_host_handoff_boxes_stream
evict = numpy.zeros(256 << 20, numpy.uint8)


def call():
    cur.value = 0
    fn(stream.ctypes.data, stream.size, ctypes.byref(cur), ctypes.byref(dry), box.ctypes.data, cnt.ctypes.data, H, W, B, S, n_keep,
       gt.ctypes.data, off.ctypes.data, 1, ws.ctypes.data, out.ctypes.data)


for mode in ("hot", "cold", "cold, stretch touched", "cold, a dry run on the old proposal first"):
    ts = []
    for rep in range(12):
        if mode != "hot":
            evict[::64] += 1
        if mode == "cold, stretch touched":
            int(stream[::16].sum())
        if mode.startswith("cold, a dry run"):
            call()
        t = time.perf_counter()
        call()
        ts.append(time.perf_counter() - t)
    ts.sort()
    print("%-45s median %.1f us  min %.1f us" % (mode, 1e6 * ts[len(ts) // 2], 1e6 * ts[0]))
