"""host time of denet_host_handoff_boxes_stream (the native part of the fast RoI hand-off) on synthetic proposals; runs without a GPU"""
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
for full in (1.0, 0.5, 0.0):
    cnt = numpy.where(rng.rand(B) < full, S, 200).astype(numpy.int32)
    x0 = rng.randint(0, W - 1, (B, S)); y0 = rng.randint(0, H - 1, (B, S))
    box = numpy.stack([x0, y0, numpy.minimum(W - 1, x0 + rng.randint(0, 20, (B, S))), numpy.minimum(H - 1, y0 + rng.randint(0, 20, (B, S)))], -1).astype(numpy.int32)
    box = numpy.ascontiguousarray(box)
    gt = rng.rand(B * 3, 4)
    off = (numpy.arange(B + 1) * 3).astype(numpy.int32)
    ws = numpy.empty(2 * S, numpy.int32)
    out = numpy.empty((B, S, 4), numpy.float32)
    cur, dry = ctypes.c_long(0), ctypes.c_int(0)
    fn = L.denet_host_handoff_boxes_stream
    try:
        fu = L.denet_host_handoff_boxes_stream_u
    except AttributeError:
        fu = None
    uni = numpy.empty(stream.size, numpy.float64)
    if fu is not None:
        assert L.denet_host_mt_uniforms(stream.ctypes.data, stream.size, uni.ctypes.data) == 0
    out2 = numpy.empty((B, S, 4), numpy.float32)

    def call_u():
        cur.value = 0
        rc = fu(stream.ctypes.data, stream.size, ctypes.byref(cur), ctypes.byref(dry), box.ctypes.data, cnt.ctypes.data, H, W, B, S, n_keep,
                gt.ctypes.data, off.ctypes.data, 1, ws.ctypes.data, out2.ctypes.data, uni.ctypes.data)
        assert rc == 0 and not dry.value

    def call():
        cur.value = 0
        rc = fn(stream.ctypes.data, stream.size, ctypes.byref(cur), ctypes.byref(dry), box.ctypes.data, cnt.ctypes.data, H, W, B, S, n_keep,
                gt.ctypes.data, off.ctypes.data, 1, ws.ctypes.data, out.ctypes.data)
        assert rc == 0 and not dry.value
    call()
    dt = 1e9
    for _ in range(10):
        t = time.perf_counter()
        for _ in range(20):
            call()
        dt = min(dt, (time.perf_counter() - t) / 20)
    if fu is None:
        print("share of images with a full list %.1f: %.1f us per call (library without the table form)" % (full, 1e6 * dt))
        continue
    call_u()
    c_u = cur.value
    dtu = 1e9
    for _ in range(10):
        t = time.perf_counter()
        for _ in range(20):
            call_u()
        dtu = min(dtu, (time.perf_counter() - t) / 20)
    t = time.perf_counter()
    for _ in range(20):
        L.denet_host_mt_uniforms(stream.ctypes.data, stream.size, uni.ctypes.data)
    dtt = (time.perf_counter() - t) / 20
    call()
    print("  with the table: %.1f us per call (table itself %.1f us, ahead of the hand-off), identical: %s, cursor %s" % (1e6 * dtu, 1e6 * dtt, numpy.array_equal(out.view(numpy.uint32), out2.view(numpy.uint32)), c_u == cur.value))
    print("share of images with a full list %.1f: %.1f us per call, cursor %d, checksum %.6f" % (full, 1e6 * dt, cur.value, float(out.astype(numpy.float64).sum())))
