"""GPU executor of augmentation plans (denet_amd/dataset/plan.py): the host decodes each image to u8 and plans; crop,
border, Pillow-exact resampling, /255, colour jitter, normalisation, mirror and the NHWC layout run on the device
(csrc/image.hip) and land directly in the training batch. One pinned staging buffer and one H2D copy per batch carry
the decoded images and the fixed-point coefficient tables; there is no host fallback: an unsupported plan raises.

Same pixels as the host path (`plan.render_pil` == `image_loader.load_sample_proc`): bit-identical u8 images after
resampling, bit-identical fp32 batch except where the `contrast` jitter is drawn (its grey mean comes from exact
integer sums instead of numpy's fp32 pairwise mean: relative difference ~1e-7)."""
import ctypes

import numpy
from PIL import Image

from . import plan as planmod

# Pillow's resampling filter -> the native table builder's (denet_host_resample_coeffs): NEAREST is a one-tap index table there
FILTER_ID = {int(Image.LANCZOS): 1, int(Image.BILINEAR): 2, int(Image.BICUBIC): 3, int(Image.NEAREST): 4}


def _align(n, a=16):
    return (n + a - 1) // a * a


class DeviceRenderer:
    def __init__(self, crop, cp=4):
        self.crop = int(crop)
        self.cp = int(cp)
        self._pinned = None
        self._dev = None
        self._scratch = [None, None]
        self._sums = None
        self._copy_done = None       # event: the previous batch's H2D copy has read the pinned staging buffer

    # ---- host side: decode + coefficient tables -------------------------------------------------------------------
    @staticmethod
    def decode(fname):
        with Image.open(fname) as im:
            if im.mode not in ("RGB", "L"):
                raise NotImplementedError("device rendering expects RGB or greyscale images (got mode %s)" % im.mode)
            return numpy.ascontiguousarray(numpy.asarray(im.convert("RGB"), dtype=numpy.uint8))

    def _expand(self, size, steps):
        """plan steps -> device ops (kind, 7 ints), window sizes, box ends; returns (ops, wh, in1, staging ints, scratch px)
        kinds: 0 crop (sw, sh, px, py, x0, y0) + (w, h); 1 reduce (in_w, in_h, fx, fy); 2 pass (horizontal, in_w, in_h,
        out_n, filter) + in1 (upper end of the source box, fractional after a reduce)"""
        ops, wh, in1s = [], [], []
        cur, need, tab_ints = tuple(size), size[0] * size[1], 0

        def add_pass(horizontal, cur, out_n, filt, box_end):
            fid = FILTER_ID.get(int(filt))
            if fid is None:
                raise NotImplementedError("device rendering supports the NEAREST / LANCZOS / BILINEAR / BICUBIC filters only")
            ops.append((2, horizontal, cur[0], cur[1], out_n, fid, 0, 0))
            wh.append((0, 0))
            in1s.append(box_end)
            support = {1: 3.0, 2: 1.0, 3: 2.0, 4: 0.0}[fid] * max(box_end / out_n, 1.0)
            return out_n * (2 + int(numpy.ceil(support)) * 2 + 1) + 8

        for st in steps:
            if st[0] == "crop":
                _, x0, y0, x1, y1, px, py, cw, ch = st
                ops.append((0, cur[0], cur[1], px, py, x0, y0, 0))
                wh.append((x1 - x0, y1 - y0))
                in1s.append(0.0)
                cur = (x1 - x0, y1 - y0)
            else:
                if st[0] == "thumbnail":
                    t = planmod.thumbnail_size(cur[0], cur[1], st[1])
                    if t is None or t == cur:
                        continue
                    new, filt = t, st[2]
                    # Image.thumbnail resizes with reducing_gap=2.0: when the shrink factor reaches 4 a box-filter
                    # reduce() comes first and the convolution maps the fractional box (w / fx, h / fy), which Pillow
                    # hands to its C code as float32
                    fx, fy = int(cur[0] / t[0] / 2.0) or 1, int(cur[1] / t[1] / 2.0) or 1
                    if int(filt) == int(Image.NEAREST):       # Image.resize: "reducing_gap is not None and resample != NEAREST"
                        fx = fy = 1
                    box = (float(cur[0]), float(cur[1]))
                    if fx > 1 or fy > 1:
                        ops.append((1, cur[0], cur[1], fx, fy, 0, 0, 0))
                        wh.append((0, 0))
                        in1s.append(0.0)
                        box = (float(numpy.float32(cur[0] / fx)), float(numpy.float32(cur[1] / fy)))
                        cur = ((cur[0] + fx - 1) // fx, (cur[1] + fy - 1) // fy)
                else:
                    new, filt = (st[1], st[2]), st[3]
                    box = (float(cur[0]), float(cur[1]))
                if new[0] != cur[0] or box[0] != cur[0]:
                    tab_ints += add_pass(1, cur, new[0], filt, box[0])
                    cur = (new[0], cur[1])
                    need = max(need, cur[0] * cur[1])
                if new[1] != cur[1] or box[1] != cur[1]:
                    tab_ints += add_pass(0, cur, new[1], filt, box[1])
                    cur = (cur[0], new[1])
            need = max(need, cur[0] * cur[1])
        if cur != (self.crop, self.crop):
            raise Exception("plan renders %dx%d, expected %dx%d" % (cur + (self.crop, self.crop)))
        return ops, wh, in1s, tab_ints, need

    # ---- device side -------------------------------------------------------------------------------------------------
    def render_batch(self, plans, images=None, out=None):
        """plans: list of plan dictionaries; images: optional pre-decoded u8 (H, W, 3) arrays.
        -> torch fp32 tensor [B, crop, crop, cp] on the current device / stream. One native call per batch
        (denet_image_render_batch: staging copy, coefficient tables, one H2D copy, all kernel launches)"""
        import torch
        from .. import ops as dops
        L = dops._L()
        B = len(plans)
        if images is None:
            images = [self.decode(p["fname"]) for p in plans]
        ops_all, wh_all, in1_all, op_off = [], [], [], [0]
        staging, scratch_px = 0, 1
        for p, a in zip(plans, images):
            cached = p.get("_device_ops")          # expanded at planning time (DeviceImageLoader.plan)
            o, wh, in1, tab_ints, need = cached if cached is not None else self._expand((a.shape[1], a.shape[0]), p["steps"])
            ops_all += o
            wh_all += wh
            in1_all += in1
            op_off.append(len(ops_all))
            staging += _align(a.size) + 4 * tab_ints + 64
            scratch_px = max(scratch_px, need)
        if self._pinned is None or self._pinned.numel() < staging:
            if self._copy_done is not None:
                self._copy_done.synchronize()
            self._pinned = torch.empty(max(staging, 1 << 20), dtype=torch.uint8).pin_memory()
            self._dev = torch.empty(self._pinned.numel(), dtype=torch.uint8, device="cuda")
        for i in range(2):
            if self._scratch[i] is None or self._scratch[i].numel() < 4 * scratch_px:
                self._scratch[i] = torch.empty(4 * scratch_px, dtype=torch.uint8, device="cuda")
        if self._sums is None:
            self._sums = torch.zeros(4, dtype=torch.int64, device="cuda")
        if out is None:
            out = torch.empty(B, self.crop, self.crop, self.cp, dtype=torch.float32, device="cuda")
        i32 = lambda v, shape: numpy.ascontiguousarray(numpy.array(v, dtype=numpy.int32).reshape(shape))
        ops_a = i32(ops_all if ops_all else [[0] * 8], (-1, 8))
        wh_a = i32(wh_all if wh_all else [[0, 0]], (-1, 2))
        in1_a = numpy.ascontiguousarray(numpy.array(in1_all if in1_all else [0.0], dtype=numpy.float64))
        off_a = i32(op_off, (-1,))
        src_wh = i32([[a.shape[1], a.shape[0]] for a in images], (-1, 2))
        src_ptrs = (ctypes.c_void_p * B)(*[a.ctypes.data for a in images])
        photo_n = i32([len(p["photo"]) for p in plans], (-1,))
        photo_ops = i32([[op for op, _ in p["photo"]] + [0] * (3 - len(p["photo"])) for p in plans], (-1, 3))
        photo_alpha = numpy.ascontiguousarray(numpy.array(
            [[al for _, al in p["photo"]] + [0.0] * (3 - len(p["photo"])) for p in plans], dtype=numpy.float64))
        has_noise = numpy.array([p["noise"] is not None for p in plans], dtype=numpy.uint8)
        noise = numpy.ascontiguousarray(numpy.array(
            [[float(v) for v in p["noise"]] if p["noise"] is not None else [0.0] * 3 for p in plans], dtype=numpy.float64))
        ms_list = [p["mean_std"] for p in plans]
        if any(m is not None for m in ms_list) and any(m != ms_list[0] for m in ms_list):
            raise NotImplementedError("one mean / std normalisation per batch")
        mean_std = numpy.array(ms_list[0], dtype=numpy.float32) if ms_list[0] is not None else None
        mirror = numpy.array([bool(p["mirror"]) for p in plans], dtype=numpy.uint8)
        ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p) if a is not None else None
        if self._copy_done is not None:
            self._copy_done.synchronize()          # the previous batch's copy has finished reading the staging buffer
        dops.check(L.denet_image_render_batch(
            B, ctypes.cast(src_ptrs, ctypes.c_void_p), ptr(src_wh), ptr(off_a), ptr(ops_a), ptr(wh_a), ptr(in1_a), ptr(photo_n),
            ptr(photo_ops), ptr(photo_alpha), ptr(noise), ptr(has_noise), ptr(mean_std), ptr(mirror), self.crop, self.cp,
            out.data_ptr(), self._pinned.data_ptr(), self._pinned.numel(), self._dev.data_ptr(), self._scratch[0].data_ptr(),
            self._scratch[1].data_ptr(), self._scratch[0].numel(), self._sums.data_ptr(), dops.stream_ptr()), "image_render_batch")
        self._copy_done = torch.cuda.Event()
        self._copy_done.record()
        return out


_WORKER_SHM = {}


def _decode_into_shm(job):
    """pool worker: decode one image straight into the shared staging block (no pixels travel through a pipe)"""
    from multiprocessing import resource_tracker, shared_memory
    fname, shm_name, offset, w, h = job
    shm = _WORKER_SHM.get(shm_name)
    if shm is None:
        shm = shared_memory.SharedMemory(name=shm_name)
        try:        # the parent owns the segment: keep this process's resource tracker from unlinking it at exit
            resource_tracker.unregister(shm._name, "shared_memory")
        except Exception:
            pass
        _WORKER_SHM[shm_name] = shm
    a = DeviceRenderer.decode(fname)
    assert a.shape == (h, w, 3), (fname, a.shape, (h, w))
    numpy.frombuffer(shm.buf, dtype=numpy.uint8, count=w * h * 3, offset=offset).reshape(h, w, 3)[...] = a
    return True


class DeviceImageLoader:
    """ImageLoader whose pixels are rendered on the GPU: same format_params, same per-image seeds drawn from the parent's
    random stream, same metas; `load_batch` returns the fp32 NHWC batch already in HBM instead of host arrays.
    Planning is serial (it owns the global random streams, microseconds per image). JPEG decoding runs on `thread_num`
    workers: threads (decode="thread": Pillow releases the interpreter lock while decoding, but not while handing the
    bytes to numpy) or processes writing into a shared-memory block (decode="process": nothing but two integers per image
    crosses a pipe and the training thread keeps the interpreter to itself)."""

    def __init__(self, thread_num, is_training, format_params={}, cp=4, decode="thread", params=None):
        from .image_loader import ImageLoader
        # `params`: an existing ImageLoader (e.g. a dataset's, with its colour statistics) used for make_args only
        self.params = params if params is not None else ImageLoader(1, is_training, format_params)
        self.renderer = DeviceRenderer(self.params.crop, cp)
        self.decode_mode = decode
        self.workers = max(1, int(thread_num))
        if decode == "process":
            import multiprocessing as mp
            self.pool = mp.get_context("spawn").Pool(self.workers)
            self._shm = [None, None]
            self._turn = 0
        else:
            from concurrent.futures import ThreadPoolExecutor
            self.pool = ThreadPoolExecutor(self.workers)

    def close(self):
        if self.decode_mode == "process":
            self.pool.terminate()
            for shm in self._shm:
                if shm is not None:
                    shm.close()
                    shm.unlink()
            self._shm = [None, None]
        else:
            self.pool.shutdown()

    def plan(self, images):
        """seeds + plans of a list of images; consumes the parent's random stream exactly like ImageLoader.load (one
        randint per image, nothing else)"""
        import random
        args_list = [self.params.make_args(image) for image in images]
        state, np_state = random.getstate(), numpy.random.get_state()
        plans = [p for a in args_list for p in planmod.plan_views(a)]      # ten per image under test-time multicrop
        random.setstate(state)
        numpy.random.set_state(np_state)
        for p in plans:       # the device program of each plan, so that the prefetch thread has no Python work left to do
            p["_device_ops"] = self.renderer._expand(tuple(p["meta"]["image_size"]), p["steps"])
        return plans

    def _decode(self, plans):
        """decoded u8 image of every plan; a file that several plans of the batch share (the ten multicrop views) is decoded once"""
        first = {}
        uniq = [p for p in plans if first.setdefault(p["fname"], len(first)) == len(first) - 1]
        if len(uniq) < len(plans):
            images = self._decode(uniq)
            return [images[first[p["fname"]]] for p in plans]
        if self.decode_mode != "process":
            return list(self.pool.map(DeviceRenderer.decode, [p["fname"] for p in plans]))
        from multiprocessing import shared_memory
        sizes = [p["meta"]["image_size"] for p in plans]
        offs, off = [], 0
        for w, h in sizes:
            offs.append(off)
            off = _align(off + w * h * 3, 64)
        slot = self._turn
        self._turn ^= 1
        shm = self._shm[slot]
        if shm is None or shm.size < off:
            if shm is not None:
                shm.close()
                shm.unlink()
            shm = self._shm[slot] = shared_memory.SharedMemory(create=True, size=max(off, 32 << 20))
        jobs = [(p["fname"], shm.name, o, w, h) for p, o, (w, h) in zip(plans, offs, sizes)]
        self.pool.map(_decode_into_shm, jobs, chunksize=max(1, len(jobs) // (2 * self.workers)))
        return [numpy.frombuffer(shm.buf, dtype=numpy.uint8, count=w * h * 3, offset=o).reshape(h, w, 3)
                for o, (w, h) in zip(offs, sizes)]

    def render(self, plans, out=None):
        """decode (workers) + render (current stream) a batch of plans -> torch fp32 [B, crop, crop, cp] on the device.
        Uses no random numbers: safe on a background thread while the main thread trains"""
        return self.renderer.render_batch(plans, images=self._decode(plans), out=out)

    def load_batch(self, images, out=None):
        """-> (torch fp32 [len(images), crop, crop, cp] on the device, list of metas)"""
        plans = self.plan(images)
        return self.render(plans, out=out), [p["meta"] for p in plans]

    def iterate(self, images, batch_size):
        """yields (x_dev, metas) batches over `images`; the last batch is padded with samples drawn by random.randint
        like DatasetAbstract.export; batch k+1 is decoded and rendered on a side stream while batch k is consumed"""
        import math
        import random
        import torch
        from concurrent.futures import ThreadPoolExecutor
        plans = self.plan(images)
        n = len(plans)
        size = batch_size * math.ceil(n / batch_size)
        index = list(range(n)) + [random.randint(0, n - 1) for _ in range(size - n)]
        from .. import ops
        side = ops.side_stream(1)        # a stream that really runs beside the compute stream (ops.init_streams)
        bufs = [torch.empty(batch_size, self.renderer.crop, self.renderer.crop, self.renderer.cp, device="cuda") for _ in range(2)]

        def job(k):
            batch = [plans[i] for i in index[k * batch_size:(k + 1) * batch_size]]
            with torch.cuda.stream(side):
                x = self.render(batch, out=bufs[k % 2])
                ev = torch.cuda.Event()
                ev.record(side)
            return x, [p["meta"] for p in batch], ev

        nb = size // batch_size
        with ThreadPoolExecutor(1) as bg:
            fut = bg.submit(job, 0)
            for k in range(nb):
                x, metas, ev = fut.result()
                torch.cuda.current_stream().wait_event(ev)
                if k + 1 < nb:
                    # the buffer batch k+1 is rendered into was consumed by batch k-1: order the side stream behind it
                    done = torch.cuda.Event()
                    done.record()
                    side.wait_event(done)
                    fut = bg.submit(job, k + 1)
                yield x, metas
