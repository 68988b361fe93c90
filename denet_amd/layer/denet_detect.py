"""`DND` — DeNet detection layer. Mirrors denet/layer/denet_detect.py (DeNetDetectLayer :25-424): final 1x1
convolution -> class (+joint fitness) logits, box regressors (:60-107); IoU based target assignment
(get_target :147-235); detection / box costs (get_errors :238-301, cost :304-313). Pass-through layer.
The shipped recipes write `DND[0.5,1,1]`, i.e. a scalar overlap threshold that get_target then indexes as a
pair (:172,:198) — a latent bug of the reference; scalar or [t_class, t_bbox] are both accepted here."""
import os

import numpy

from . import AbstractLayer, InitialLayer
from .convolution import ConvLayer
from .. import common
from .. import ops


class DeNetDetectLayer(AbstractLayer):
    type_name = "denet-detect"

    def __init__(self, layers, class_num=10, overlap_threshold=0.5, cost_factor=1.0, bbox_factor=0.0, indfit_factor=0.0,
                 use_jointfit=False, use_bounded_iou=False, json_param={}):
        super().__init__(layer_index=len(layers))
        self.input = layers[-1].output
        self.input_shape = layers[-1].output_shape
        self.output = layers[-1].output
        self.output_shape = layers[-1].output_shape

        self.cost_factor = json_param.get("costFactor", cost_factor)
        self.bbox_factor = json_param.get("bboxFactor", bbox_factor)
        self.class_num = json_param.get("classNum", class_num)
        self.overlap_threshold = json_param.get("overlapThreshold", overlap_threshold)
        self.use_jointfit = json_param.get("useJointFitness", use_jointfit)
        self.use_bounded_iou = json_param.get("useBoundedIoU", use_bounded_iou)
        self.indfit_factor = json_param.get("fitnessFactor", indfit_factor)
        self.use_indfit = (self.indfit_factor > 0.0)
        assert not (self.use_indfit and self.use_jointfit), "Cannot enable both fitness methods at once!"

        self.sparse_layer = common.find_layers(layers, "denet-sparse", False)
        assert self.sparse_layer is not None, "Error: Requires denet-sparse layer to be specified before denet-detect layer!"
        self.use_bbox_reg = (self.bbox_factor > 0.0)
        self.batch_size = self.sparse_layer.batch_size
        self.sample_num = self.sparse_layer.sample_num
        if self.use_jointfit:
            self.fitness_num = 5
            self.null_class = self.class_num * self.fitness_num
            s0 = self.class_num * self.fitness_num + 1
        else:
            self.fitness_num = 6
            self.null_class = self.class_num
            s0 = self.class_num + 1
        s1 = 4 if self.use_bbox_reg else 0
        s2 = self.fitness_num if self.use_indfit else 0       # independent fitness distribution (:103-108)
        self.s0, self.s1, self.s2 = s0, s1, s2
        self.layers = [ConvLayer([InitialLayer(self.input, self.input_shape)], (s0 + s1 + s2, self.input_shape[1], 1, 1),
                                 (1, 1), True, "valid", 0.0)]
        if self.use_indfit:
            self.indfit_shape = (self.batch_size, s2, self.sample_num, self.sample_num)
        self.det_shape = (self.batch_size, s0, self.sample_num, self.sample_num)
        if self.use_bbox_reg:
            self.bbox_shape = (self.batch_size, s1, self.sample_num, self.sample_num)
        self._targets = None

    def _thresholds(self):
        t = self.overlap_threshold
        if isinstance(t, (list, tuple)):
            return float(t[0]), float(t[1] if len(t) > 1 else t[0])
        return float(t), float(t)

    @staticmethod
    def parse_desc(layers, name, tags, params):
        if name != "DND":
            return False
        layers.append(DeNetDetectLayer(layers, params.get("classNum"), params.get(0, 0.5), params.get(1, 1.0),
                                       params.get(2, 0.0), params.get(3, 0.0), "J" in tags, "B" in tags))
        return True

    def import_json(self, json_param):
        super().import_json(json_param)
        if "conv" in json_param:
            self.layers[0].import_json(json_param["conv"])

    def export_json(self):
        json = super().export_json()
        json.update({"costFactor": self.cost_factor, "bboxFactor": self.bbox_factor, "fitnessFactor": self.indfit_factor,
                     "useJointFitness": self.use_jointfit, "useBoundedIoU": self.use_bounded_iou,
                     "classNum": self.class_num, "overlapThreshold": self.overlap_threshold})
        return json

    def _buf(self, name, shape):
        """reusable pinned host staging buffers (RoI-major target arrays)"""
        import torch
        bufs = self.__dict__.setdefault("_bufs", {})
        t = bufs.get(name)
        if t is None or tuple(t.shape) != tuple(shape):
            t = torch.empty(shape, dtype=torch.float32)
            try:
                t = t.pin_memory()
            except RuntimeError:      # no HIP runtime (CPU-only host logic tests)
                pass
            bufs[name] = t
        return t.numpy()

    def build_targets(self, metas):
        """IoU based target assignment of denet_detect.py:147-235 in RoI-major layout (row m = b*sn*sn + index,
        index = j*sn + i, :174-175), one native host call for the batch (denet_host_detect_targets) writing the
        pinned staging buffers. returns det [M,s0], bbox_valid [M] or None, bbox_reg [M,8] or None (float32)"""
        from .. import lib as _lib
        t0, t1 = self._thresholds()
        sn, B = self.sample_num, self.batch_size
        S = sn * sn
        sp = self.sparse_layer
        boxes = sp.sample_boxes
        if any(len(bx) != S for bx in boxes):
            return self.build_targets_numpy(metas)        # short RoI lists (inference-style calls): general path
        roi = numpy.ascontiguousarray(numpy.stack(boxes, axis=0), dtype=numpy.float64)
        gts = [numpy.asarray(m["bbox"], dtype=numpy.float64).reshape(-1, 4) for m in metas]
        off = numpy.zeros(B + 1, dtype=numpy.int32)
        numpy.cumsum([len(g) for g in gts], out=off[1:])
        gt = numpy.ascontiguousarray(numpy.concatenate(gts, axis=0)) if off[-1] > 0 else numpy.zeros((1, 4))
        cls = numpy.ascontiguousarray(numpy.concatenate([numpy.asarray(m["class"], dtype=numpy.int32).reshape(-1)
                                                         for m in metas]) if off[-1] > 0 else numpy.zeros(1), dtype=numpy.int32)
        det = self._buf("det", (B * S, self.s0))
        valid = reg = None
        if self.use_bbox_reg:
            valid = self._buf("valid", (B * S,))
            reg = self._buf("reg", (B * S, 8))
        fit = self._buf("indfit", (B * S, self.s2)) if self.use_indfit else None
        _lib.check(_lib.load().denet_host_detect_targets(
            gt.ctypes.data, off.ctypes.data, cls.ctypes.data, roi.ctypes.data, B, S, self.s0, self.null_class,
            self.fitness_num, int(bool(self.use_jointfit)), float(t0), float(t1), det.ctypes.data,
            valid.ctypes.data if valid is not None else None, reg.ctypes.data if reg is not None else None,
            fit.ctypes.data if fit is not None else None), "detect_targets")
        self._fit_target = fit
        return det, valid, reg

    def build_targets_numpy(self, metas):
        """the same assignment with numpy (general RoI counts; the checker of the native path in tests/test_host.py):
        IoU based target assignment of denet_detect.py:147-235 in RoI-major layout: row m = b*sn*sn + index,
        index = j*sn + i (:174-175). The per-match Python loops of the reference become fancy indexing — every
        assignment writes the constants 1.0 / 0.0, so the order of the matches is immaterial.
        returns det [M,s0], bbox_valid [M] or None, bbox_reg [M,8] or None (float32)"""
        t0, t1 = self._thresholds()
        sn, B = self.sample_num, self.batch_size
        S = sn * sn
        sp = self.sparse_layer
        det = self._buf("det", (B * S, self.s0))
        det.fill(0.0)
        det[:, self.null_class] = 1.0
        valid = reg = None
        if self.use_bbox_reg:
            valid = self._buf("valid", (B * S,))
            reg = self._buf("reg", (B * S, 8))
            valid.fill(0.0)
            reg.fill(0.0)
            reg[:, [2, 3, 6, 7]] = 1.0
        fit = None
        if self.use_indfit:
            fit = self._buf("indfit", (B * S, self.s2))
            fit.fill(0.0)
            fit[:, 0] = 1.0

        for b, meta in enumerate(metas):
            boxes = sp.sample_boxes[b]
            if len(meta["bbox"]) == 0 or len(boxes) == 0:
                continue
            overlap = common.get_overlap_iou(meta["bbox"], boxes)
            bbox_indexs, sample_indexs = numpy.where(overlap > t0)
            if len(bbox_indexs) > 0:
                rows = b * S + sample_indexs
                cls = numpy.asarray(meta["class"], dtype=numpy.int64)[bbox_indexs]
                if self.use_jointfit:
                    # float64 arithmetic on float32 IoUs, as the Python loop does (:176,:180-183)
                    sample_f = (overlap[bbox_indexs, sample_indexs].astype(numpy.float64) - t0) / (1.0 - t0)
                    f = numpy.clip((self.fitness_num * sample_f).astype(numpy.int64), 0, self.fitness_num - 1)
                    det[rows, cls * self.fitness_num + f] = 1.0
                else:
                    det[rows, cls] = 1.0
                det[rows, self.null_class] = 0.0
                if self.use_indfit:
                    # f = clip(1 + floor((n-1) * sample_f), 1, n-1) on the float64 value of the float32 IoU (:188-192)
                    sample_f = (overlap[bbox_indexs, sample_indexs].astype(numpy.float64) - t0) / (1.0 - t0)
                    f = numpy.clip(1 + numpy.floor((self.fitness_num - 1) * sample_f).astype(numpy.int64), 1,
                                   self.fitness_num - 1)
                    fit[rows, 0] = 0.0
                    fit[rows, f] = 1.0
            if self.use_bbox_reg:
                overlap_max = overlap.argmax(axis=0)
                idx = numpy.arange(len(boxes))
                idx = idx[overlap[overlap_max, idx] > t1]
                if len(idx) > 0:
                    tgt = numpy.asarray(meta["bbox"], dtype=numpy.float64)[overlap_max[idx]]
                    smp = boxes[idx]
                    rows = b * S + idx
                    valid[rows] = 1.0
                    reg[rows, 0] = 0.5 * (tgt[:, 0] + tgt[:, 2])
                    reg[rows, 1] = 0.5 * (tgt[:, 1] + tgt[:, 3])
                    reg[rows, 2] = tgt[:, 2] - tgt[:, 0]
                    reg[rows, 3] = tgt[:, 3] - tgt[:, 1]
                    reg[rows, 4] = 0.5 * (smp[:, 0] + smp[:, 2])
                    reg[rows, 5] = 0.5 * (smp[:, 1] + smp[:, 3])
                    reg[rows, 6] = smp[:, 2] - smp[:, 0]
                    reg[rows, 7] = smp[:, 3] - smp[:, 1]

        det /= det.sum(axis=1, keepdims=True)
        det /= S
        if self.use_bbox_reg:
            valid /= S
        if self.use_indfit:
            fit /= fit.sum(axis=1, keepdims=True)
            fit /= S
        self._fit_target = fit
        return det, valid, reg

    def get_target(self, model, samples, metas):
        """the reference's packed target vector: det (B,s0,sn,sn) || bbox_valid (B,sn,sn) || bbox_reg (B,8,sn,sn)"""
        sn, B = self.sample_num, self.batch_size
        det, valid, reg = self.build_targets(metas)
        yt_value = det.reshape(B, sn, sn, self.s0).transpose(0, 3, 1, 2).flatten()
        if self.use_bbox_reg:
            yt_value = numpy.concatenate((yt_value, valid.flatten(),
                                          reg.reshape(B, sn, sn, 8).transpose(0, 3, 1, 2).flatten()))
        if self.use_indfit:
            yt_value = numpy.concatenate((yt_value,
                                          self._fit_target.reshape(B, sn, sn, self.s2).transpose(0, 3, 1, 2).flatten()))
        return numpy.array([], dtype=numpy.int64), yt_value

    def cost(self, yt_index, yt_value):
        return True

    # ---- execution ----
    @property
    def conv(self):
        return self.layers[0]

    def prepare_target(self, ctx, model, data_x, metas):
        import torch
        det, valid, reg = self.build_targets(metas)
        b = self._bufs
        t = {"valid": None, "reg": None, "fit": None}
        t["det"], ev = ops.upload_async(b["det"])
        if self.use_bbox_reg:
            t["valid"], _ = ops.upload_async(b["valid"])
            t["reg"], ev = ops.upload_async(b["reg"])
        if self.use_indfit:
            t["fit"], ev = ops.upload_async(b["indfit"])
        t["event"] = ev           # the copies are ordered on the copy stream: the last event covers all of them
        self._targets = t

    def set_target(self, ctx, yt_index, yt_value):
        """unpack the reference's packed target vector (:241-254) into RoI-major device arrays"""
        import torch
        sn, B = self.sample_num, self.batch_size
        shapes = [self.det_shape]
        if self.use_bbox_reg:
            shapes += [(B, sn, sn), (B, 8, sn, sn)]
        if self.use_indfit:
            shapes += [self.indfit_shape]
        v = common.ndarray_unpack(numpy.asarray(yt_value, dtype=numpy.float32), shapes)
        det = numpy.ascontiguousarray(v[0].transpose(0, 2, 3, 1).reshape(B * sn * sn, self.s0))
        t = {"det": torch.from_numpy(det).cuda(), "valid": None, "reg": None, "fit": None}
        if self.use_bbox_reg:
            t["valid"] = torch.from_numpy(numpy.ascontiguousarray(v[1].reshape(-1))).cuda()
            t["reg"] = torch.from_numpy(numpy.ascontiguousarray(v[2].transpose(0, 2, 3, 1).reshape(-1, 8))).cuda()
        if self.use_indfit:
            t["fit"] = torch.from_numpy(numpy.ascontiguousarray(v[-1].transpose(0, 2, 3, 1).reshape(-1, self.s2))).cuda()
        self._targets = t

    def forward(self, ctx):
        self.conv.forward(ctx)

    def loss_backward(self, ctx, cost_out, want_grad=True):
        """cost_out: device float[2] receiving (DET cost, BBOX cost)"""
        logits = self.conv.output.data
        M = self.batch_size * self.sample_num * self.sample_num
        lg = logits.view(M, self.conv.kp)
        dl = ops.empty(M, self.conv.kp) if want_grad else None
        t = self._targets
        ops.wait_upload(t.get("event"))
        ops.detect_loss(lg, t["det"], t["valid"], t["reg"], self.sparse_layer.sample_bbox, dl, cost_out,
                        self.batch_size, self.s0, self.s1, float(self.cost_factor), float(self.bbox_factor),
                        bool(self.use_bounded_iou), fit_target=t.get("fit"), nfit=self.s2,
                        fit_factor=float(self.indfit_factor))
        if want_grad:
            self.conv.output.grad = dl.view(logits.shape)

    def backward(self, ctx):
        if self.conv.output.grad is not None:
            self.conv.backward(ctx)

    # ---- inference ------------------------------------------------------------------------------
    def get_detections(self, model, data_x, data_m, params):
        """list of {"detections": [(pr, cls, (x0, y0, x1, y1)), ...], "meta": meta} per image — the return value of
        the reference's get_detections (denet_detect.py:316-424): corner detector -> RoIs -> head in test mode ->
        per-class threshold + NMS (build_detections_nms, denet_detect.cc:99-173)"""
        import torch
        pr_threshold = params.get("prThreshold", 0.01)
        nms_threshold = params.get("nmsThreshold", 0.5)
        sp = self.sparse_layer
        saved = (sp.corner_threshold, sp.corner_max)
        sp.corner_threshold = params.get("cornerThreshold", sp.corner_threshold)
        sp.corner_max = params.get("cornerMax", 1024)
        use_soft_nms = params.get("useSoftNMS", 0) == 1
        timer = common.Timer()
        try:
            model.forward(data_x, None, train=False)
        finally:
            sp.corner_threshold, sp.corner_max = saved
        B, S = self.batch_size, self.sample_num * self.sample_num
        logits = self.conv.output.data.view(B * S, self.conv.kp)
        t0, _ = self._thresholds()
        det_pr, fitness, bbox = ops.detect_decode(logits, sp.sample_bbox, self.class_num, self.use_jointfit, self.s1, t0, nfit=self.s2)
        counts = numpy.array([len(bx) for bx in sp.sample_boxes], dtype=numpy.int32)
        self.last_outputs = (det_pr, fitness, bbox, counts)
        results = []
        fit_h = fitness.cpu().numpy()
        box_h = bbox.cpu().numpy()
        if not use_soft_nms:
            keep = ops.detect_nms(det_pr, fitness, bbox, torch.from_numpy(counts).cuda(), B, S, self.class_num,
                                  pr_threshold, nms_threshold).cpu().numpy()
            # one pass over the whole batch: the flat index walks (image, class, RoI), the reference's output order
            flat = numpy.flatnonzero(keep.reshape(-1).view(numpy.bool_))
            b_idx, rem = numpy.divmod(flat, self.class_num * S)
            cls_idx, roi_idx = numpy.divmod(rem, S)
            rows = b_idx * S + roi_idx
            # tolist() turns the fp32 values into the same Python floats float() would, in one C loop
            prs = numpy.exp(fit_h[rows, cls_idx]).tolist()
            cls_l = cls_idx.tolist()
            bx = box_h[rows]
            boxes = list(zip(bx[:, 0].tolist(), bx[:, 1].tolist(), bx[:, 2].tolist(), bx[:, 3].tolist()))
            ends = numpy.cumsum(numpy.bincount(b_idx, minlength=B)).tolist()
            lo = 0
            for b in range(B):
                hi = ends[b]
                results.append({"detections": list(zip(prs[lo:hi], cls_l[lo:hi], boxes[lo:hi])),
                                "meta": data_m[b] if data_m is not None else None})
                lo = hi
        else:
            # one wave per (class, image) on the device; a single image is quicker on host copies (one native call instead of three
            # launches and two dependent copies: 2.21 against 2.44 ms at B = 1, 21.7 against 14.4 ms at B = 32).
            # DENET_SOFT_NMS_HOST=1 / 0 forces the host / device form
            on_host = os.environ.get("DENET_SOFT_NMS_HOST")
            # (the device form keeps a (class, image) pair's candidates in LDS: S <= 4096 RoIs per image; beyond that - sample_num
            # > 64 - the host call, which has no such limit)
            if S > 4096 or ((on_host == "1") if on_host in ("0", "1") else B < 4):
                det_h = numpy.ascontiguousarray(det_pr.cpu().numpy())
                scores, cls_idx, rows, per = ops.soft_nms_batch_host(det_h, numpy.ascontiguousarray(fit_h), numpy.ascontiguousarray(box_h),
                                                                     counts, B, S, self.class_num, pr_threshold, nms_threshold)
            else:
                scores, cls_idx, rows, per = ops.soft_nms_batch(det_pr, fitness, bbox, torch.from_numpy(counts).cuda(), B, S,
                                                                self.class_num, pr_threshold, nms_threshold)
            prs, cls_l = numpy.exp(scores).tolist(), cls_idx.tolist()
            bx = box_h[rows]
            boxes = list(zip(bx[:, 0].tolist(), bx[:, 1].tolist(), bx[:, 2].tolist(), bx[:, 3].tolist()))
            lo = 0
            for b in range(B):
                hi = lo + int(per[b])
                results.append({"detections": list(zip(prs[lo:hi], cls_l[lo:hi], boxes[lo:hi])),
                                "meta": data_m[b] if data_m is not None else None})
                lo = hi
        self.detect_ms = timer.current_ms()
        return results
