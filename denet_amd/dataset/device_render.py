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

from .. import lib
from . import plan as planmod

FILTER_ID = {int(Image.LANCZOS): 1, int(Image.BILINEAR): 2, int(Image.BICUBIC): 3}


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

    # ---- host side: decode + coefficient tables -------------------------------------------------------------------
    @staticmethod
    def decode(fname):
        with Image.open(fname) as im:
            if im.mode not in ("RGB", "L"):
                raise NotImplementedError("device rendering expects RGB or greyscale images (got mode %s)" % im.mode)
            return numpy.ascontiguousarray(numpy.asarray(im.convert("RGB"), dtype=numpy.uint8))

    @staticmethod
    def _coeffs(in_size, out_size, filt, in1=None):
        L = lib.load()
        fid = FILTER_ID.get(int(filt))
        if fid is None:
            raise NotImplementedError("device rendering supports the LANCZOS / BILINEAR / BICUBIC filters only")
        in1 = float(in_size) if in1 is None else float(in1)
        scale = max(in1 / out_size, 1.0)
        cap = out_size * (int(numpy.ceil(3.0 * scale)) * 2 + 1)
        bounds = numpy.empty(2 * out_size, dtype=numpy.int32)
        kk = numpy.empty(cap, dtype=numpy.int32)
        ks = L.denet_host_resample_coeffs(in_size, 0.0, in1, out_size, fid,
                                          bounds.ctypes.data_as(ctypes.c_void_p), kk.ctypes.data_as(ctypes.c_void_p), cap)
        if ks <= 0:
            raise lib.DenetHipError("resample_coeffs: " + L.denet_last_error().decode())
        return bounds, kk[:out_size * ks], ks

    def _expand(self, size, steps):
        """plan steps -> device ops with sizes: ("crop", ...) | ("pass", horizontal, in_w, in_h, out_n, tables)"""
        ops, cur, need = [], tuple(size), size[0] * size[1]
        for st in steps:
            if st[0] == "crop":
                _, x0, y0, x1, y1, px, py, cw, ch = st
                ops.append(("crop", cur[0], cur[1], px, py, x0, y0, x1 - x0, y1 - y0))
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
                    box = (float(cur[0]), float(cur[1]))
                    if fx > 1 or fy > 1:
                        ops.append(("reduce", cur[0], cur[1], fx, fy))
                        box = (float(numpy.float32(cur[0] / fx)), float(numpy.float32(cur[1] / fy)))
                        cur = ((cur[0] + fx - 1) // fx, (cur[1] + fy - 1) // fy)
                        need = max(need, cur[0] * cur[1])
                else:
                    new, filt = (st[1], st[2]), st[3]
                    box = (float(cur[0]), float(cur[1]))
                if new[0] != cur[0] or box[0] != cur[0]:
                    ops.append(("pass", 1, cur[0], cur[1], new[0], self._coeffs(cur[0], new[0], filt, box[0])))
                    cur = (new[0], cur[1])
                    need = max(need, cur[0] * cur[1])
                if new[1] != cur[1] or box[1] != cur[1]:
                    ops.append(("pass", 0, cur[0], cur[1], new[1], self._coeffs(cur[1], new[1], filt, box[1])))
                    cur = (cur[0], new[1])
            need = max(need, cur[0] * cur[1])
        if cur != (self.crop, self.crop):
            raise Exception("plan renders %dx%d, expected %dx%d" % (cur + (self.crop, self.crop)))
        return ops, need

    # ---- device side -------------------------------------------------------------------------------------------------
    def render_batch(self, plans, images=None, out=None):
        """plans: list of plan dictionaries; images: optional pre-decoded u8 (H, W, 3) arrays.
        -> torch fp32 tensor [B, crop, crop, cp] on the current device / stream"""
        import torch
        from .. import ops as dops
        L = dops._L()
        B = len(plans)
        if images is None:
            images = [self.decode(p["fname"]) for p in plans]
        progs, off, layout, scratch_px = [], 0, [], 1
        for p, a in zip(plans, images):
            ops_, need = self._expand((a.shape[1], a.shape[0]), p["steps"])
            scratch_px = max(scratch_px, need)
            img_off = off
            off = _align(off + a.size)
            tabs = []
            for o in ops_:
                if o[0] == "pass":
                    bounds, kk, ks = o[5]
                    tabs.append((off, off + bounds.nbytes))
                    layout.append((off, bounds))
                    layout.append((off + bounds.nbytes, kk))
                    off = _align(off + bounds.nbytes + kk.nbytes)
                else:
                    tabs.append(None)
            layout.append((img_off, a.reshape(-1)))
            progs.append((img_off, ops_, tabs))
        # one staging buffer, one copy
        if self._pinned is None or self._pinned.numel() < off:
            self._pinned = torch.empty(max(off, 1 << 20), dtype=torch.uint8).pin_memory()
            self._dev = torch.empty(self._pinned.numel(), dtype=torch.uint8, device="cuda")
        host = self._pinned.numpy()
        for o, arr in layout:
            host[o:o + arr.nbytes] = arr.view(numpy.uint8).reshape(-1)
        self._dev[:off].copy_(self._pinned[:off], non_blocking=True)
        for i in range(2):
            if self._scratch[i] is None or self._scratch[i].numel() < 4 * scratch_px:
                self._scratch[i] = torch.empty(4 * scratch_px, dtype=torch.uint8, device="cuda")
        if self._sums is None:
            self._sums = torch.zeros(4, dtype=torch.int64, device="cuda")
        if out is None:
            out = torch.empty(B, self.crop, self.crop, self.cp, dtype=torch.float32, device="cuda")
        stream = dops.stream_ptr()
        base = self._dev.data_ptr()
        for b, (p, (img_off, ops_, tabs)) in enumerate(zip(plans, progs)):
            cur_ptr, cur_is_src, flip = base + img_off, True, 0
            for o, tab in zip(ops_, tabs):
                dst = self._scratch[flip].data_ptr()
                if o[0] == "crop":
                    _, sw, sh, px, py, x0, y0, w, h = o
                    dops.check(L.denet_image_crop(cur_ptr, dst, sw, sh, 3 if cur_is_src else 4, px, py, x0, y0, w, h, stream),
                               "image_crop")
                else:
                    in_w, in_h = (o[1], o[2]) if o[0] == "reduce" else (o[2], o[3])
                    if cur_is_src:      # the first step is a resampling of the whole image: bring it to RGBX first
                        dops.check(L.denet_image_crop(cur_ptr, dst, in_w, in_h, 3, 0, 0, 0, 0, in_w, in_h, stream), "image_crop")
                        cur_ptr, cur_is_src, flip = dst, False, flip ^ 1
                        dst = self._scratch[flip].data_ptr()
                if o[0] == "reduce":
                    dops.check(L.denet_image_reduce(cur_ptr, dst, o[1], o[2], o[3], o[4], stream), "image_reduce")
                elif o[0] == "pass":
                    _, horizontal, in_w, in_h, out_n, (bounds, kk, ks) = o
                    dops.check(L.denet_image_resample_pass(cur_ptr, dst, in_w, in_h, out_n, horizontal, base + tab[0],
                                                           base + tab[1], ks, stream), "image_resample_pass")
                cur_ptr, cur_is_src, flip = dst, False, flip ^ 1
            if cur_is_src:          # a plan without geometric steps: the decoded image is the view
                a = images[b]
                dst = self._scratch[flip].data_ptr()
                dops.check(L.denet_image_crop(cur_ptr, dst, a.shape[1], a.shape[0], 3, 0, 0, 0, 0, a.shape[1], a.shape[0], stream),
                           "image_crop")
                cur_ptr = dst
            n_ops = len(p["photo"])
            ops_arr = (ctypes.c_int * 3)(*([op for op, _ in p["photo"]] + [0] * (3 - n_ops)))
            alphas = (ctypes.c_double * 3)(*([al for _, al in p["photo"]] + [0.0] * (3 - n_ops)))
            noise = (ctypes.c_double * 3)(*[float(v) for v in p["noise"]]) if p["noise"] is not None else None
            ms = (ctypes.c_float * 6)(*p["mean_std"]) if p["mean_std"] is not None else None
            dops.check(L.denet_image_finish(cur_ptr, out[b].data_ptr(), self.crop, self.crop, self.cp, n_ops, ops_arr, alphas,
                                            noise, ms, int(p["mirror"]), self._sums.data_ptr(), stream), "image_finish")
        return out


class DeviceImageLoader:
    """ImageLoader whose pixels are rendered on the GPU: same format_params, same per-image seeds drawn from the parent's
    random stream, same metas; `load_batch` returns the fp32 NHWC batch already in HBM instead of host arrays.
    Planning is serial (it owns the global random streams, microseconds per image); JPEG decoding runs on `thread_num`
    threads (Pillow releases the GIL while decoding)."""

    def __init__(self, thread_num, is_training, format_params={}, cp=4):
        from concurrent.futures import ThreadPoolExecutor
        from .image_loader import ImageLoader
        self.params = ImageLoader(1, is_training, format_params)       # parameter parsing and make_args only
        self.renderer = DeviceRenderer(self.params.crop, cp)
        self.pool = ThreadPoolExecutor(max(1, int(thread_num)))

    def load_batch(self, images, out=None):
        """-> (torch fp32 [len(images), crop, crop, cp] on the device, list of metas)"""
        import random
        args_list = [self.params.make_args(image) for image in images]
        decoded = self.pool.map(DeviceRenderer.decode, [image["fname"] for image in images])
        state, np_state = random.getstate(), numpy.random.get_state()
        plans = [planmod.plan_sample(a) for a in args_list]
        random.setstate(state)
        numpy.random.set_state(np_state)
        x = self.renderer.render_batch(plans, images=list(decoded), out=out)
        return x, [p["meta"] for p in plans]
