"""`DND` — DeNet detection layer. Mirrors denet/layer/denet_detect.py (DeNetDetectLayer :25-424): final 1x1
convolution -> class (+joint fitness) logits, box regressors (:60-107); IoU based target assignment
(get_target :147-235); detection / box costs (get_errors :238-301, cost :304-313). Pass-through layer.
The shipped recipes write `DND[0.5,1,1]`, i.e. a scalar overlap threshold that get_target then indexes as a
pair (:172,:198) — a latent bug of the reference; scalar or [t_class, t_bbox] are both accepted here."""
import math

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
        if self.use_indfit:
            raise NotImplementedError("independent fitness head (indfit_factor > 0) is outside the hot path")

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
        self.s0, self.s1 = s0, s1
        self.layers = [ConvLayer([InitialLayer(self.input, self.input_shape)], (s0 + s1, self.input_shape[1], 1, 1),
                                 (1, 1), True, "valid", 0.0)]
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

    def get_target(self, model, samples, metas):
        """vectorised restatement of denet_detect.py:147-235 (the per-match Python loops become fancy indexing;
        assignments only ever write the constant 1.0 / 0.0 so the order of matches is immaterial)"""
        t0, t1 = self._thresholds()
        sn = self.sample_num
        det_pr = numpy.zeros(self.det_shape, dtype=numpy.float32)
        det_pr[:, self.null_class, ...] = 1.0
        if self.use_bbox_reg:
            bbox_valid = numpy.zeros((self.batch_size, sn, sn), dtype=numpy.float32)
            bbox_reg = numpy.zeros((self.batch_size, 8, sn, sn), dtype=numpy.float32)
            bbox_reg[:, 2, ...] = 1.0
            bbox_reg[:, 3, ...] = 1.0
            bbox_reg[:, 6, ...] = 1.0
            bbox_reg[:, 7, ...] = 1.0

        for b, meta in enumerate(metas):
            samples = [bbox for _, bbox in self.sparse_layer.sample_bbox_list[b]]
            if len(meta["bbox"]) > 0 and len(samples) > 0:
                overlap = common.get_overlap_iou(meta["bbox"], samples)
                bbox_indexs, sample_indexs = numpy.where(overlap > t0)
                if len(bbox_indexs) > 0:
                    si = sample_indexs % sn
                    sj = sample_indexs // sn
                    cls = numpy.asarray(meta["class"], dtype=numpy.int64)[bbox_indexs]
                    if self.use_jointfit:
                        # float64 arithmetic on float32 IoUs, as the Python loop does (:176,:180-183)
                        sample_f = (overlap[bbox_indexs, sample_indexs].astype(numpy.float64) - t0) / (1.0 - t0)
                        f = numpy.clip((self.fitness_num * sample_f).astype(numpy.int64), 0, self.fitness_num - 1)
                        det_pr[b, cls * self.fitness_num + f, sj, si] = 1.0
                    else:
                        det_pr[b, cls, sj, si] = 1.0
                    det_pr[b, self.null_class, sj, si] = 0.0

                if self.use_bbox_reg:
                    overlap_max = overlap.argmax(axis=0)
                    idx = numpy.arange(len(samples))
                    keep = overlap[overlap_max, idx] > t1
                    idx = idx[keep]
                    if len(idx) > 0:
                        obj = overlap_max[idx]
                        tgt = numpy.asarray(meta["bbox"], dtype=numpy.float64)[obj]
                        smp = numpy.asarray(samples, dtype=numpy.float64)[idx]
                        si, sj = idx % sn, idx // sn
                        bbox_valid[b, sj, si] = 1.0
                        bbox_reg[b, 0, sj, si] = 0.5 * (tgt[:, 0] + tgt[:, 2])
                        bbox_reg[b, 1, sj, si] = 0.5 * (tgt[:, 1] + tgt[:, 3])
                        bbox_reg[b, 2, sj, si] = tgt[:, 2] - tgt[:, 0]
                        bbox_reg[b, 3, sj, si] = tgt[:, 3] - tgt[:, 1]
                        bbox_reg[b, 4, sj, si] = 0.5 * (smp[:, 0] + smp[:, 2])
                        bbox_reg[b, 5, sj, si] = 0.5 * (smp[:, 1] + smp[:, 3])
                        bbox_reg[b, 6, sj, si] = smp[:, 2] - smp[:, 0]
                        bbox_reg[b, 7, sj, si] = smp[:, 3] - smp[:, 1]

        det_pr /= det_pr.sum(axis=1)[:, None, ...]
        nfactor = sn * sn
        det_pr /= nfactor
        if self.use_bbox_reg:
            bbox_valid /= nfactor
        yt_value = det_pr.flatten()
        if self.use_bbox_reg:
            yt_value = numpy.concatenate((yt_value, bbox_valid.flatten(), bbox_reg.flatten()))
        return numpy.array([], dtype=numpy.int64), yt_value

    def cost(self, yt_index, yt_value):
        return True

    # ---- execution ----
    @property
    def conv(self):
        return self.layers[0]

    def set_target(self, ctx, yt_index, yt_value):
        """unpack the reference's packed target vector (:241-254) into RoI-major device arrays"""
        import torch
        sn, B = self.sample_num, self.batch_size
        shapes = [self.det_shape]
        if self.use_bbox_reg:
            shapes += [(B, sn, sn), (B, 8, sn, sn)]
        v = common.ndarray_unpack(numpy.asarray(yt_value, dtype=numpy.float32), shapes)
        det = numpy.ascontiguousarray(v[0].transpose(0, 2, 3, 1).reshape(B * sn * sn, self.s0))
        t = {"det": torch.from_numpy(det).cuda(non_blocking=True), "valid": None, "reg": None}
        if self.use_bbox_reg:
            t["valid"] = torch.from_numpy(numpy.ascontiguousarray(v[1].reshape(-1))).cuda(non_blocking=True)
            t["reg"] = torch.from_numpy(numpy.ascontiguousarray(v[2].transpose(0, 2, 3, 1).reshape(-1, 8))).cuda(non_blocking=True)
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
        ops.detect_loss(lg, t["det"], t["valid"], t["reg"], self.sparse_layer.sample_bbox, dl, cost_out,
                        self.batch_size, self.s0, self.s1, float(self.cost_factor), float(self.bbox_factor),
                        bool(self.use_bounded_iou))
        if want_grad:
            self.conv.output.grad = dl.view(logits.shape)

    def backward(self, ctx):
        if self.conv.output.grad is not None:
            self.conv.backward(ctx)
