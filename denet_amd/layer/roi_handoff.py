"""The RoI hand-off of the `DNS` layer: device proposal -> edited bbox array -> gather (reference: DeNetSparseLayer.get_target,
denet/layer/denet_sparse.py:164-206, the editing loop :184-201, set_samples :155-161). The reference edits the list in Python
with the GPU idle; here the generator outputs the editing will draw are drawn AHEAD (while the device runs the backbone) and the
bbox array is written either on the device or by one native host call; the Python-side list, which the detection targets read a
few layers later, is produced beside the device's gather. Everything is exact including the stdlib generator's position
(tests/test_parity_gpu.py::test_device_side_editing_equals_the_host_list, tests/test_host.py::test_prefetched_generator_*)."""
import math
import os
import random

import numpy

from .. import common
from .. import ops

# The RoI hand-off (device proposal -> edited bbox array -> gather) has two short forms and the ordinary host path:
#   device_edit  no image proposes more RoIs than the list keeps (no random.sample): the bbox array is written on the device
#   fast         any other batch: ONE native host call writes the bbox array, its bookkeeping runs beside the gather
#   host         what remains (clustering, coverage logging, a generator that moved, CPU runs): edit + upload on the host
# DENET_SHORT_HANDOFF=0 is the one switch: the ordinary path only. The three names below are test hooks, not configuration.
SHORT_HANDOFF = os.environ.get("DENET_SHORT_HANDOFF", "1") != "0"
PREFETCH_RANDOM = DEVICE_EDIT = FAST_HANDOFF = SHORT_HANDOFF
WARM_HANDOFF = SHORT_HANDOFF          # dry runs of the fast form's native call while the host waits (test hook like the others)
# seconds between two dry runs while the host waits for the proposal (DENET_HANDOFF_WARM_MS; 0: none)
WARM_PERIOD = float(os.environ.get("DENET_HANDOFF_WARM_MS", "0.25")) * 1e-3


class PyRandomMirror:
    """Vectorised draws from the stdlib `random` stream. numpy's RandomState is the same MT19937 with the same
    53-bit double construction as CPython's `random`, so the generator state can be moved across, doubles drawn
    in bulk, and the advanced state moved back: bit-identical to calling random.random() n times. sample() runs
    CPython's random.sample(range(n), k) algorithm on the same state in native host code
    (denet_host_py_random_sample)."""
    _rs = None

    def __init__(self):
        # one generator object for the process: an unseeded RandomState() gathers OS entropy on construction
        if PyRandomMirror._rs is None:
            PyRandomMirror._rs = numpy.random.RandomState(0)
        self.rs = PyRandomMirror._rs
        self.pull()

    def pull(self):
        """adopt the current state of the stdlib generator"""
        st = random.getstate()
        self._version, self._gauss = st[0], st[2]
        self.key = numpy.array(st[1][:-1], dtype=numpy.uint32)
        self.pos = numpy.array([st[1][-1]], dtype=numpy.int32)
        self._finger = (st[1][-1], st[1][:4])

    def fresh(self):
        """True if the stdlib generator has not moved since pull() (any draw advances the position word or, on a
        refill, rewrites the first state words)"""
        st = random.getstate()[1]
        return self._finger == (st[-1], st[:4])

    def push(self):
        """hand the advanced state back to the stdlib generator"""
        random.setstate((self._version, tuple(self.key.tolist()) + (int(self.pos[0]),), self._gauss))

    def doubles(self, n):
        if n <= 0:
            return numpy.zeros((0,), dtype=numpy.float64)
        self.rs.set_state(("MT19937", self.key, int(self.pos[0])))
        v = self.rs.random_sample(n)
        ns = self.rs.get_state()
        self.key = numpy.ascontiguousarray(ns[1], dtype=numpy.uint32)
        self.pos[0] = int(ns[2])
        return v

    def sample(self, n, k):
        """indices chosen by random.sample(range(n), k) (== the positions random.sample(list, k) picks)"""
        from .. import lib as _lib
        out = numpy.empty(k, dtype=numpy.int32)
        pool = numpy.empty(n, dtype=numpy.int32)
        _lib.check(_lib.load().denet_host_py_random_sample(self.key.ctypes.data, self.pos.ctypes.data, int(n), int(k),
                                                           pool.ctypes.data, out.ctypes.data), "py_random_sample")
        return out


def py_random_doubles(n):
    """the next n values of random.random(), drawn in one call"""
    m = PyRandomMirror()
    v = m.doubles(n)
    m.push()
    return v


class RoiHandoff:
    """the short forms of the hand-off, mixed into DeNetSparseLayer (which owns the proposal, the ordinary path and the lists)"""

    def _prefetch_random(self):
        """The editing draws 8 generator outputs per random box (4 doubles) and a few per element kept by random.sample; what it
        draws does not depend on the device's proposal, only how much. The outputs of the step are therefore drawn HERE, while
        the device runs the backbone (denet_host_mt_prefetch, on a copy of the state), and the hand-off only walks through
        them (denet_host_edit_samples_stream: ~4x less host time between the device's proposal and the gather). The state
        after the consumed outputs is handed back to `random`; if the generator moved meanwhile, or the stretch runs dry, the
        ordinary path runs."""
        import ctypes
        from .. import lib as _lib
        B, S = self.batch_size, self.sample_count
        n = 8 * B * S + 8192
        max_snaps = n // 624 + 3
        buf = getattr(self, "_pf_buf", None)
        if buf is None or buf[0].size != n:
            if self._on_device():
                import torch
                # pinned: the stretch is also uploaded for the device-side editing (_upload_for_device_edit)
                self._pf_pinned = torch.empty(n, dtype=torch.int32).pin_memory()
                out0 = self._pf_pinned.numpy().view(numpy.uint32)
            else:
                out0 = numpy.empty(n, dtype=numpy.uint32)
            buf = self._pf_buf = (out0, numpy.empty((max_snaps, 624), dtype=numpy.uint32), numpy.empty(max_snaps, dtype=numpy.int64))
            self._pf_uniforms = numpy.empty(n, dtype=numpy.float64)
        prev = self.__dict__.pop("_pf_upload", None)
        if prev is not None:
            prev.synchronize()            # the previous step's upload out of this buffer (a whole step old: done)
        out, snaps, first = buf
        mirror = PyRandomMirror()
        key, pos = mirror.key.copy(), mirror.pos.copy()
        ns = ctypes.c_int(0)
        _lib.check(_lib.load().denet_host_mt_prefetch(key.ctypes.data, pos.ctypes.data, n, out.ctypes.data, snaps.ctypes.data,
                                                      first.ctypes.data, max_snaps, ctypes.byref(ns)), "mt_prefetch")
        # the doubles random.random() would return from every position of the stretch (the random boxes of the fast hand-off read
        # them from this table: denet_host_handoff_boxes_stream_u) - made here, while the device runs the backbone
        _lib.check(_lib.load().denet_host_mt_uniforms(out.ctypes.data, n, self._pf_uniforms.ctypes.data), "mt_uniforms")
        self._prefetch = {"mirror": mirror, "n": n, "ns": ns.value, "pos0": int(mirror.pos[0])}

    def _upload_for_device_edit(self, metas, prep):
        """what the device-side editing reads, sent while the device runs the backbone: the generator outputs drawn ahead and the
        ground-truth boxes"""
        import torch
        pf = self._prefetch
        B, S = self.batch_size, self.sample_count
        st = self.__dict__.get("_de_static")
        if st is None:
            st = self._de_static = {"gt": torch.empty((B * S + 1, 4), dtype=torch.float64).pin_memory(),
                                    "off": torch.empty(B + 1, dtype=torch.int32).pin_memory(),
                                    "status": torch.zeros(2, dtype=torch.int32, device="cuda"),
                                    "status_host": torch.zeros(2, dtype=torch.int32).pin_memory()}
        off, gt = prep["off"], prep["gt"]
        ng = int(off[-1])
        if ng > B * S:
            return
        st["off"].numpy()[:] = off
        st["gt"].numpy()[:max(ng, 1)] = gt[:max(ng, 1)]
        side = ops.side_stream(2)
        with torch.cuda.stream(side):
            mt = self._pf_pinned.cuda(non_blocking=True)
            gtd = st["gt"][:max(ng, 1)].cuda(non_blocking=True)
            offd = st["off"].cuda(non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(side)
        self._pf_upload = ev
        self._dev_edit = {"metas": metas, "mt": mt, "gt": gtd, "off": offd, "ev": ev, "n": pf["n"]}

    def _short_handoff_ready(self):
        """what the two short forms of the hand-off share, checked BEFORE the host starts waiting for the proposal (nothing between
        here and the hand-off draws from the generator: the host only waits): the prefetched stretch, the prepared ground truth,
        a generator that has not moved since. (pf, prep) or None"""
        pf, prep = self.__dict__.get("_prefetch"), self.__dict__.get("_prep")
        if pf is None or prep is None or self.cluster or self.proposal_count != self.sample_count or not self._on_device():
            return None
        if not (DEVICE_EDIT or FAST_HANDOFF) or not pf["mirror"].fresh():
            return None
        de = self.__dict__.get("_dev_edit")
        if DEVICE_EDIT and de is not None and de["metas"] is prep["metas"] and not de.get("staged"):
            # the device-side editing, should the counts allow it: its stream dependencies and its output buffer are set up here
            # as well (the uploads it waits for were queued when the step began)
            import torch
            cur = torch.cuda.current_stream()
            cur.wait_event(de["ev"])
            for t in (de["mt"], de["gt"], de["off"]):
                t.record_stream(cur)
            de["out"] = torch.empty((self.batch_size * self.sample_count, 4), dtype=torch.float32, device="cuda")
            de["staged"] = True
        return pf, prep

    def _short_handoff(self, hc, tot, ready=False):
        """the two short forms of the hand-off; `ready`: what _short_handoff_ready returned ahead of the wait (False: check now)"""
        if ready is False:
            ready = self._short_handoff_ready()
        if ready is None:
            return False
        pf, prep = ready
        return self._device_edit(hc, tot, pf, prep) or self._fast_handoff(hc, pf, prep)

    def _device_edit(self, hc, tot, pf, prep):
        """The hand-off when no image proposes more RoIs than the list keeps (no random.sample: every step of a detector early in
        training): denet_edit_samples_device writes the bbox array from the proposal as it lies on the device, the uploaded
        generator outputs and the ground truth - the gather starts behind one small device-to-host copy. The host's own editing
        (same stretch of outputs, same values: test_device_side_editing_equals_the_host_list) is left to the first reader of the
        Python-side list. False: not this case, the ordinary path runs."""
        import torch
        de = self.__dict__.get("_dev_edit")
        B, S = self.batch_size, self.sample_count
        n_keep = S - math.floor(self.random_sample * S)
        if not DEVICE_EDIT or de is None or de["metas"] is not prep["metas"]:
            return False
        if int(hc.max()) > n_keep:
            return False                 # random.sample would trim a list: the fast hand-off's case
        from .. import lib as _lib
        cl = self.corner_layer
        if de.get("staged"):
            out = de["out"]
        else:
            cur = torch.cuda.current_stream()
            cur.wait_event(de["ev"])
            for t in (de["mt"], de["gt"], de["off"]):
                t.record_stream(cur)
            out = torch.empty((B * S, 4), dtype=torch.float32, device="cuda")
        r, st = self._res_dev, self._de_static
        _lib.check(_lib.load().denet_edit_samples_device(
            _lib.ptr(r[:B * S * 4]), _lib.ptr(r[B * S * 5:]), cl.height, cl.width, _lib.ptr(de["mt"]), de["n"], 0,
            _lib.ptr(de["gt"]), _lib.ptr(de["off"]), int(bool(self.sample_gt)), B, S, n_keep, _lib.ptr(out), _lib.ptr(st["status"]),
            _lib.stream_ptr()), "edit_samples_device")
        st["status_host"].copy_(st["status"], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._de_status_ev = [ev, None]          # [1]: the outputs the HOST's editing of the same batch uses (the job below)
        pend = self._de_status_ev
        self.sample_bbox = out
        self._dev_edit = None
        self.device_edits = getattr(self, "device_edits", 0) + 1
        self.handoff_modes["device_edit"] += 1

        def job():
            # the host's share, beside the device's gather and head: sample tuples, then the editing on the SAME stretch of
            # generator outputs (no other choice is consistent with what the device has used)
            timer = common.Timer()
            self._finish_samples(timer, True, log=False)
            det, cnt = self._raw_samples if self._raw_samples is not None else (numpy.zeros((B, S, 5), dtype=numpy.float32),
                                                                                 numpy.zeros(B, dtype=numpy.int32))      # cold detector
            if self._pinned is None:
                self._pinned = torch.empty((B * S, 4), dtype=torch.float32).pin_memory()
            f32 = self._pinned.numpy()
            if not pf["mirror"].fresh():
                raise RuntimeError("the stdlib generator moved between the device-side RoI editing and the host's: the two lists "
                                   "would differ")
            self._prefetch = None
            done = self._native_edit_stream(pf, numpy.ascontiguousarray(det, dtype=numpy.float32),
                                            numpy.ascontiguousarray(cnt, dtype=numpy.int32), prep, f32)
            if done is None:
                raise RuntimeError("the prefetched generator outputs ran dry in the host's editing of a batch the device has edited")
            out_pr, out_box, mirror = done
            pend[1] = self._last_cursor
            mirror.push()
            self._prep = None
            self._sample_pr, self._sample_boxes = list(out_pr), list(out_box)
            self._sample_bbox_f32 = f32.reshape(B, S, 4)
            # the device's status word (copied back right behind its kernel, long done): a bad one stops the step BEFORE the
            # solver applies an update trained on a wrong bbox array (train_step resolves this job ahead of the solver)
            self._check_device_edit_status()
        self._lazy_edit = job
        return True

    def _warm_fast_handoff(self, ready):
        """A dry run of the fast hand-off's native call, repeated WHILE the host waits for the proposal (ops.wait_stream(idle=...)):
        the same code over the step's generator stretch, the previous step's counts and whatever the result buffer holds, into a
        scratch array. The call takes ~55 us with warm caches and 190-350 us cold - and cold is how the hand-off found it, with the
        device idle meanwhile: the host has queued a whole forward pass and polled for ~19 ms since it drew the stretch
        (`tools/exp/handoff_native_cold.py`, `handoff_host.py`: one dry run ahead of the wait does not last, one every quarter
        millisecond does). Nothing of a dry run is kept (its cursor, its status and its output are its own)."""
        bufs = self.__dict__.get("_ho_bufs")
        if not WARM_HANDOFF or not FAST_HANDOFF or ready is None or bufs is None or self.__dict__.get("_pf_uniforms") is None:
            return
        import ctypes
        pf, prep = ready
        out = self._pf_buf[0]
        cursor, dry = ctypes.c_long(0), ctypes.c_int(0)
        H, W = bufs["hw"]
        cnt = bufs["cnt"][bufs["turn"]]
        bufs["fn"](out.ctypes.data, pf["n"], ctypes.byref(cursor), ctypes.byref(dry), bufs["hp"], cnt.ctypes.data, H, W, self.batch_size,
                   self.sample_count, bufs["n_keep"], prep["gt"].ctypes.data, prep["off"].ctypes.data, int(bool(self.sample_gt)),
                   bufs["ws"].ctypes.data, bufs["scratch"].ctypes.data, self._pf_uniforms.ctypes.data)       # (return value ignored)

    def _fast_handoff(self, hc, pf, prep):
        """The hand-off with the device idle as short as the host can make it: ONE native call that writes the bbox array alone
        (denet_host_handoff_boxes_stream: the selection and the random boxes on the prefetched generator outputs, no score
        arithmetic, no lists) and its upload into a device buffer that already exists. Everything else the rest of the step wants
        from the host - the sample tuples, the lists (denet_host_handoff_stream over the same outputs), the generator's state
        handed back to `random` - is left to the first reader (_resolve_edit), who runs beside the device's gather and head.
        False: the stretch ran dry: the ordinary path."""
        import ctypes
        import torch
        from .. import lib as _lib
        if not FAST_HANDOFF:
            return False
        B, S = self.batch_size, self.sample_count
        bufs = self.__dict__.get("_ho_bufs")
        if bufs is None:
            cl = self.corner_layer
            if self._pinned is None:
                self._pinned = torch.empty((B * S, 4), dtype=torch.float32).pin_memory()
            bufs = self._ho_bufs = {
                "ws": numpy.empty(2 * S, dtype=numpy.int32), "turn": 0, "scratch": numpy.empty(B * S * 4, dtype=numpy.float32),
                "det": [numpy.empty((B, S, 5), dtype=numpy.float32) for _ in range(2)],
                "cnt": [numpy.empty(B, dtype=numpy.int32) for _ in range(2)],
                "dev": [torch.empty((B * S, 4), dtype=torch.float32, device="cuda") for _ in range(2)],
                "cursor": ctypes.c_long(0), "dry": ctypes.c_int(0), "fn": _lib.load().denet_host_handoff_boxes_stream_u,
                "hp": self._res_host.data_ptr(), "f32": self._pinned.numpy(), "hw": (cl.height, cl.width),
                "n_keep": S - math.floor(self.random_sample * S)}
        bufs["turn"] ^= 1
        turn = bufs["turn"]
        cnt = bufs["cnt"][turn]
        cnt[:] = hc
        out, snaps, first = self._pf_buf
        cursor, dry = bufs["cursor"], bufs["dry"]
        cursor.value = 0
        off, gt = prep["off"], prep["gt"]
        H, W = bufs["hw"]
        _lib.check(bufs["fn"](out.ctypes.data, pf["n"], ctypes.byref(cursor), ctypes.byref(dry), bufs["hp"], cnt.ctypes.data, H, W, B, S,
                              bufs["n_keep"], gt.ctypes.data, off.ctypes.data, int(bool(self.sample_gt)), bufs["ws"].ctypes.data,
                              bufs["f32"].ctypes.data, self._pf_uniforms.ctypes.data), "handoff_boxes_stream")
        if dry.value:
            return False
        dev = bufs["dev"][turn]
        dev.copy_(self._pinned, non_blocking=True)
        self.sample_bbox = dev
        self._prefetch = None
        c = cursor.value
        self.fast_handoffs = getattr(self, "fast_handoffs", 0) + 1
        self.handoff_modes["fast"] += 1

        def job():
            # the lists: sample tuples + the editing once more, over the same outputs (float32 array into a scratch buffer: the
            # pinned one may still be on its way to the device)
            if not pf["mirror"].fresh():
                raise RuntimeError("the stdlib generator moved between the RoI hand-off and the bookkeeping of its draws")
            det = bufs["det"][turn]
            out_pr, out_box = self._edit_out()
            c2, dry2 = ctypes.c_long(0), ctypes.c_int(0)
            hp = bufs["hp"]
            _lib.check(_lib.load().denet_host_handoff_stream(
                out.ctypes.data, pf["n"], ctypes.byref(c2), ctypes.byref(dry2), hp, hp + 4 * B * S * 4, cnt.ctypes.data, H, W, B, S,
                bufs["n_keep"], gt.ctypes.data, off.ctypes.data, int(bool(self.sample_gt)), bufs["ws"].ctypes.data, det.ctypes.data,
                out_pr.ctypes.data, out_box.ctypes.data, bufs["scratch"].ctypes.data), "handoff_stream")
            assert not dry2.value and c2.value == c, (c2.value, c)
            # the state after c outputs: the snapshot they ended in, with CPython's lazy refill (a position of 624 stays 624)
            j = max(0, int(numpy.searchsorted(first[:pf["ns"]], c, side="left")) - 1)
            mirror = pf["mirror"]
            mirror.key = snaps[j].copy()
            mirror.pos[0] = (pf["pos0"] if j == 0 else 0) + (c - int(first[j]))
            assert 0 <= int(mirror.pos[0]) <= 624
            mirror.push()
            self._prep = None
            self._raw_samples = (det, cnt)
            self._sample_pr, self._sample_boxes = list(out_pr), list(out_box)
            self._sample_bbox_f32 = bufs["f32"].reshape(B, S, 4)
        self._lazy_edit = job
        return True

    def _check_device_edit_status(self):
        """the status word of the previous step's device-side editing (copied back asynchronously; a step old by now)"""
        pend = self.__dict__.pop("_de_status_ev", None)
        if pend is None:
            return
        ev, used = pend
        ev.synchronize()
        flag, dev_used = [int(v) for v in self._de_static["status_host"].tolist()]
        if flag != 0 or (used is not None and dev_used != used):
            raise RuntimeError("device-side RoI editing: status %d, %d generator outputs used (host: %s)" % (flag, dev_used, used))

    def _native_edit_stream(self, pf, det, cnt, prep, out_f32):
        """the editing on the prefetched outputs; returns (out_pr, out_box, mirror holding the state after them) or None when the
        stretch ran dry"""
        import ctypes
        from .. import lib as _lib
        B, S = self.batch_size, self.sample_count
        n_keep = S - math.floor(self.random_sample * S)
        ws = numpy.empty(2 * S, dtype=numpy.int32)
        off, gt = prep["off"], prep["gt"]
        out_pr, out_box = self._edit_out()
        out, snaps, first = self._pf_buf
        cursor, dry = ctypes.c_long(0), ctypes.c_int(0)
        _lib.check(_lib.load().denet_host_edit_samples_stream(
            out.ctypes.data, pf["n"], ctypes.byref(cursor), ctypes.byref(dry), det.ctypes.data, cnt.ctypes.data, B, S, n_keep,
            gt.ctypes.data, off.ctypes.data, int(bool(self.sample_gt)), ws.ctypes.data, out_pr.ctypes.data,
            out_box.ctypes.data, out_f32.ctypes.data), "edit_samples_stream")
        if dry.value:
            return None
        # the state after c outputs: the snapshot they ended in, with CPython's lazy refill (a position of 624 stays 624)
        c = self._last_cursor = cursor.value
        j = max(0, int(numpy.searchsorted(first[:pf["ns"]], c, side="left")) - 1)
        mirror = pf["mirror"]
        mirror.key = snaps[j].copy()
        mirror.pos[0] = (pf["pos0"] if j == 0 else 0) + (c - int(first[j]))
        assert 0 <= int(mirror.pos[0]) <= 624
        return out_pr, out_box, mirror

    def _edit_out(self):
        """the (pr [B,S], box [B,S,4]) float64 arrays an editing call writes: four sets used in turn (fresh 0.7 MB allocations
        cost page faults inside the hand-off; the lists of a step are read until its targets are built, long before the
        set comes round again)"""
        B, S = self.batch_size, self.sample_count
        ring = self.__dict__.setdefault("_edit_ring", [])
        if len(ring) < 4 or ring[0][0].shape != (B, S):
            ring[:] = [(numpy.empty((B, S), dtype=numpy.float64), numpy.empty((B, S, 4), dtype=numpy.float64)) for _ in range(4)]
            self._edit_turn = 0
        self._edit_turn = (self._edit_turn + 1) % 4
        return ring[self._edit_turn]

    def _native_edit(self, mirror, det, cnt, prep, out_f32):
        """denet_host_edit_samples on the generator state held by `mirror` (advanced in place)"""
        from .. import lib as _lib
        B, S = self.batch_size, self.sample_count
        n_keep = S - math.floor(self.random_sample * S)
        ws = numpy.empty(2 * S, dtype=numpy.int32)
        off, gt = prep["off"], prep["gt"]
        out_pr, out_box = self._edit_out()
        assert out_f32.dtype == numpy.float32 and out_f32.size == B * S * 4 and out_f32.flags.c_contiguous
        _lib.check(_lib.load().denet_host_edit_samples(
            mirror.key.ctypes.data, mirror.pos.ctypes.data, det.ctypes.data, cnt.ctypes.data, B, S, n_keep,
            gt.ctypes.data, off.ctypes.data, int(bool(self.sample_gt)), ws.ctypes.data, out_pr.ctypes.data,
            out_box.ctypes.data, out_f32.ctypes.data), "edit_samples")
        return out_pr, out_box
