"""`DNS` — DeNet sparse sampling layer. Mirrors denet/layer/denet_sparse.py (DeNetSparseLayer :26-218):
get_samples (:117-145) = corner detector -> build_samples; get_target (:164-206) = training-time RoI list
editing with the stdlib `random` module (random trim, random boxes, ground-truth injection) -> set_samples
(:155-161); the graph node is the sparse RoI gather (DeNetSparseOp, denet_sparse_op.py).

Differences by construction: the corner map never leaves the device — build_samples runs as HIP kernels
(csrc/samples.hip) and only the <= sample_count boxes per image come back for the Python-side editing; the
backbone is not run a second time (the reference compiles a separate `corner_func`)."""
import math
import random

import numpy

from . import AbstractLayer, Act, get_train
from .. import common
from .. import ops

TAP_THEANO = 0   # denet_sparse.py:72-84 (Theano CPU path, canonical per north star)
TAP_CUDA = 1     # denet_sparse_op.py:65-71


class DeNetSparseLayer(AbstractLayer):
    type_name = "denet-sparse"

    def __init__(self, layers, grid_size=3, sample_num=16, corner_threshold=0.01, random_sample=0.0, local_max=0,
                 nms_threshold=0.7, sample_gt=True, version="v2", json_param={}):
        super().__init__(layer_index=len(layers))
        self.input = layers[-1].output
        self.input_shape = layers[-1].output_shape
        self.batch_size = self.input_shape[0]

        self.grid_size = json_param.get("gridSize", grid_size)
        self.sample_num = json_param.get("sampleNum", sample_num)
        self.sample_gt = json_param.get("sampleGT", sample_gt)
        self.corner_threshold = json_param.get("cornerThreshold", corner_threshold)
        self.nms_threshold = json_param.get("nmsThreshold", nms_threshold)
        self.random_sample = json_param.get("randomSample", random_sample)
        self.local_max = json_param.get("localMax", local_max)
        self.version = json_param.get("version", version)
        self.tap_rule = json_param.get("tapRule", TAP_THEANO)

        self.corner_max = 1024
        self.thread_num = self.batch_size
        self.sample_count = self.sample_num * self.sample_num

        self.corner_layer = common.find_layers(layers, "denet-corner", True)
        assert self.corner_layer is not None, "denet-corner layer required before spare layer!"
        if self.nms_threshold < 1.0:
            raise NotImplementedError("RoI clustering (nms_threshold < 1, apply_cluster) is outside the hot path")

        self.sample_bbox_list = []
        self.sample_bbox = None   # device [B*sn*sn, 4]
        self.output_feat = self.grid_size * self.grid_size * self.corner_layer.sample_shape[1] + 2
        self.output_shape = (self.batch_size, self.output_feat, self.sample_num, self.sample_num)
        self.output = Act(self.output_shape, None, "sparse")
        self._taps = None

    @staticmethod
    def parse_desc(layers, name, tags, params):
        if name != "DNS":
            return False
        layers.append(DeNetSparseLayer(layers, params.get(0, 3), params.get(1, 4), params.get(2, 0.01),
                                       params.get(3, 0.1), params.get(4, 0), params.get(5, 1.0), not "G" in tags))
        return True

    # ---- corner detector -> sample boxes ----
    def get_samples(self, data_x, train=False, store_shared=False):
        """list[B] of list[(pr, (x0, y0, x1, y1))], the return shape of c_code.build_samples
        (denet_sparse.cc:587-592). Uses the corner map of the forward pass in flight."""
        cl = self.corner_layer
        assert cl.corner_pr is not None, "run the model forward up to the corner layer first"
        timer = common.Timer()
        box, absd, count = ops.build_samples(cl.corner_pr, float(self.corner_threshold), self.sample_count,
                                             self.corner_max, int(self.local_max))
        if store_shared:
            cl.sample_shared = cl.conv.output.data
        box, absd, count = box.cpu(), absd.cpu(), count.cpu()     # one small D2H, implicit sync
        timer.mark()
        samples = ops.samples_finish_host(box, absd, count, cl.height, cl.width).numpy()
        counts = count.tolist()
        result = []
        for b in range(self.batch_size):
            rows = samples[b, :counts[b]].tolist()
            result.append([(r[0], (r[1], r[2], r[3], r[4])) for r in rows])
        timer.mark()
        self.last_timing_ms = (timer.delta_ms(0), timer.delta_ms(1))
        return result

    def get_bbox_array(self, sample_bboxs):
        """build_bbox_array (denet_sparse.cc:670-699): bbox[b, i//sn, i%sn] = box i; the rest stays 0"""
        bboxs = numpy.zeros((self.batch_size, self.sample_num, self.sample_num, 4), dtype=numpy.float32)
        flat = bboxs.reshape(self.batch_size, self.sample_count, 4)
        for b, samples in enumerate(sample_bboxs):
            if len(samples) > 0:
                flat[b, :len(samples)] = numpy.array([s[1] for s in samples], dtype=numpy.float64).astype(numpy.float32)
        return bboxs

    def set_samples(self, sample_bboxs):
        import torch
        bboxs = self.get_bbox_array(sample_bboxs)
        self.sample_bbox = torch.from_numpy(bboxs.reshape(-1, 4)).cuda(non_blocking=True)
        self.sample_bbox_list = sample_bboxs
        return bboxs

    def get_target(self, model, data_x, metas):
        sample_bboxs = self.get_samples(data_x, train=True)
        total_cover = 0
        total_bbox = 0
        for b, meta in enumerate(metas):
            if self.log_coverage:
                cover = 0
                for meta_bbox in meta["bbox"]:
                    for _, sample_bbox in sample_bboxs[b]:
                        if common.overlap_iou(meta_bbox, sample_bbox) > 0.5:
                            cover += 1
                            break
                total_cover += cover
                total_bbox += len(meta["bbox"])

            n = self.sample_count - math.floor(self.random_sample * self.sample_count)
            if len(sample_bboxs[b]) > n:
                sample_bboxs[b] = random.sample(sample_bboxs[b], n)

            while len(sample_bboxs[b]) < self.sample_count:
                x0 = random.uniform(0.0, 1.0)
                y0 = random.uniform(0.0, 1.0)
                x1 = random.uniform(x0, 1.0)
                y1 = random.uniform(y0, 1.0)
                sample_bboxs[b].append((0.0, (x0, y0, x1, y1)))

            if self.sample_gt:
                for index, bbox in enumerate(meta["bbox"]):
                    sample_bboxs[b][-(index + 1)] = (1.0, bbox)
        self.coverage = (total_cover, total_bbox)
        self.set_samples(sample_bboxs)
        return None

    # coverage statistics are log output only in the reference (denet_sparse.py:172-182); off by default because
    # the O(GT x RoI) Python loop is pure host overhead
    log_coverage = False

    def export_json(self):
        json = super().export_json()
        json.update({"gridSize": self.grid_size, "sampleNum": self.sample_num, "sampleGT": self.sample_gt,
                     "localMax": self.local_max, "cornerThreshold": self.corner_threshold,
                     "randomSample": self.random_sample, "nmsThreshold": self.nms_threshold, "version": self.version})
        return json

    # ---- execution ----
    def forward(self, ctx):
        cl = self.corner_layer
        if get_train() or cl.sample_shared is None:
            fmap, coff, F = cl.sample_map()
        else:
            fmap, coff, F = cl.sample_shared, cl.corner_num, cl.sample_feat
        assert self.sample_bbox is not None, "set_samples() must run before the sparse layer"
        out, self._taps = ops.sparse_fwd(fmap, self.sample_bbox, coff, F, self.sample_count, self.grid_size,
                                         self.output.cp, self.tap_rule)
        self.output.data = out.view(self.batch_size, self.sample_num, self.sample_num, self.output.cp)

    def backward(self, ctx):
        cl = self.corner_layer
        dconv = cl.alloc_dconv(zero=False)
        dy = self.output.grad.view(-1, self.output.cp)
        ops.sparse_bwd(dy, self._taps, dconv, cl.corner_num, cl.sample_feat, self.sample_count, self.grid_size,
                       cl.corner_num + cl.sample_feat)
