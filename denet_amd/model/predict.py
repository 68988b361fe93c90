"""`model-predict` driver: evaluate a trained model on a dataset. Mirrors denet/model/predict.py — `test_single`
(:18-50), `test_multicrop` (:53-86), `test_detector` (:164-235) and `main` (:292-333) with the reference's flags
(`--model --input --extension --results --batch-size --predict-mode --thread-num --params`). Detection runs
`DeNetDetectLayer.get_detections` batch by batch (GPU forward + decode + NMS) and hands the per-image records to the
dataset's writer: Pascal VOC result files + 11-point AP, MSCOCO results JSON, or the ImageNet localisation error."""
import argparse
import os
import sys

import numpy

from .. import common
from .. import dataset
from . import model_cnn


def _log(log):
    if log is not None:
        return log
    from ..common import logging
    return logging.info


def _top_errors(y, yy, yt):
    y, yy, yt = numpy.array(y, numpy.int64), numpy.array(yy, numpy.int64), numpy.array(yt, numpy.int64)
    error1 = float(numpy.sum(yt != y) / yt.shape[0])
    error5 = float(1.0 - numpy.sum(numpy.any(yy == yt[:, None], axis=1)) / yt.shape[0])
    return error1, error5


def _top5(pr_i):
    k = min(5, pr_i.shape[0])
    return numpy.argpartition(-pr_i, k - 1)[:k] if k < pr_i.shape[0] else numpy.arange(pr_i.shape[0])


def test_single(mode, model, data, log=None):
    """top-1 / top-5 error of a classifier, one centre crop per image"""
    log = _log(log)
    y, yy, yt = [], [], []
    for subset in range(data.subset_num):
        data.load_from_subset(subset)
        labels = data.get_labels()
        pr = model.predict_output(data)
        for i in range(pr.shape[0]):
            y.append(int(numpy.argmax(pr[i])))
            yy.append(_top5(pr[i]))
            yt.append(labels[i])
    error1, error5 = _top_errors(y, yy, yt)
    log("Top1 - Error Rate: %.3f%%" % (100.0 * error1))
    log("Top5 - Error Rate: %.3f%%" % (100.0 * error5))
    return error1, error5


def test_multicrop(mode, model, data, log=None):
    """10-crop testing: the probabilities of the ten views of an image are summed"""
    log = _log(log)
    y, yy, yt = [], [], []
    for subset in range(data.subset_num):
        data.load_from_subset(subset)
        labels = data.get_labels()
        pr = model.predict_output(data)
        for i in range(pr.shape[0] // 10):
            pr_i = numpy.sum(pr[i * 10:(i + 1) * 10, :], axis=0)
            y.append(int(numpy.argmax(pr_i)))
            yy.append(_top5(pr_i))
            yt.append(labels[i * 10])
    error1, error5 = _top_errors(y, yy, yt)
    log("Top1 - Error Rate: %.3f%%" % (100.0 * error1))
    log("Top5 - Error Rate: %.3f%%" % (100.0 * error5))
    return error1, error5


def test_detector(mode, model, data, output_fname, params, log=None, device_render=False, thread_num=1):
    log = _log(log)
    detect_params = common.get_params_dict(params)
    detect_layer = model.layers[-1]
    class_labels_inv = {v: k for k, v in model.class_labels.items()} if model.class_labels else {}
    detections = []
    loader = None
    if device_render and hasattr(data, "images") and hasattr(data, "image_loader"):
        # scale + centre crop rendered on the GPU straight into the batch (denet_amd/dataset/device_render.py)
        from ..dataset.device_render import DeviceImageLoader
        loader = DeviceImageLoader(max(1, thread_num), False, cp=model.input.cp if model.input is not None else 4,
                                   decode="process", params=data.image_loader)
    for subset in range(data.subset_num):
        subset_det = []
        if loader is not None:
            lo = subset * data.subset_size
            hi = min((subset + 1) * data.subset_size, data.subset_total_size)
            data_size = hi - lo
            for dx, dm in loader.iterate(data.images[lo:hi], model.batch_size):
                subset_det += detect_layer.get_detections(model, dx, dm, detect_params)
        else:
            data.load_from_subset(subset)
            data_x, data_m, data_size = data.export(model.batch_size)
            for n in range(data_x.shape[0] // model.batch_size):
                dx = data_x[n * model.batch_size:(n + 1) * model.batch_size]
                dm = data_m[n * model.batch_size:(n + 1) * model.batch_size]
                subset_det += detect_layer.get_detections(model, dx, dm, detect_params)
        detections += subset_det[:data_size]          # drop the padding of the last batch
    if loader is not None:
        loader.close()
    log("Found %i detections for %i samples" % (sum(len(d["detections"]) for d in detections), len(detections)))

    out_dir = os.path.dirname(output_fname)
    if out_dir and not os.path.isdir(out_dir):
        os.makedirs(out_dir)
    raw = [{"detections": d["detections"], "meta": {k: v for k, v in d["meta"].items()}} for d in detections]
    common.json_to_file(os.path.join(out_dir, "detections.json"),
                        {"dets": raw, "classLabels": model.class_labels, "detectParams": detect_params})
    result = {"detections": detections}
    if "voc" in mode:
        from ..dataset.pascal_voc import DatasetPascalVOC
        _, _, height, width = model.get_input_shape()
        DatasetPascalVOC.export_detections(out_dir, detections, width, height, class_labels_inv)
        mean_ap, aps = DatasetPascalVOC.get_precision(detections, detect_params.get("matchIOU", 0.5))
        for cls, ap in enumerate(aps):
            log("%s - AP: %.4f" % (class_labels_inv.get(cls, cls), ap))
        log("Mean AP: %.4f" % mean_ap)
        result.update(mean_ap=mean_ap, ap=aps)
    elif "mscoco" in mode:
        data.export_detections(output_fname + ".json", detections)
        result["results_file"] = output_fname + ".json"
    elif "imagenet" in mode:
        from ..dataset.imagenet import DatasetImagenet
        result["localization_error"] = DatasetImagenet.get_localization_error(detections)
        log("Imagenet localization error: %.2f" % result["localization_error"])
    return result


def build_parser():
    parser = argparse.ArgumentParser(description="Predict labels / detections using a trained model")
    from ..common import logging
    logging.add_arguments(parser)
    parser.add_argument("--model", required=True, help="the model file")
    parser.add_argument("--input", required=True, help="The folder with data")
    parser.add_argument("--results", default="./results", type=str, help="Results folder / filename")
    parser.add_argument("--extension", default="png", help="Image file extension / dataset format string")
    parser.add_argument("--batch-size", type=int, default=100, help="Size of processing batchs")
    parser.add_argument("--predict-mode", default="single", help="single, multicrop, detect[,voc|,mscoco|,imagenet]")
    parser.add_argument("--thread-num", default=1, type=int, help="Number of threads for dataset loading")
    parser.add_argument("--params", default="", type=str, help="Additional detection params")
    parser.add_argument("--device-render", default=False, action="store_true",
                        help="detect mode: scale / crop the images on the GPU instead of in the loader processes")
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    from ..common import logging
    logging.init(args)
    model = model_cnn.load_from_file(args.model, args.batch_size)
    data = dataset.load(args.input, args.extension, class_labels=model.class_labels, thread_num=args.thread_num)
    if "single" in args.predict_mode:
        test_single(args.predict_mode, model, data)
    elif "multicrop" in args.predict_mode:
        assert "multicrop" in args.extension
        test_multicrop(args.predict_mode, model, data)
    elif "detect" in args.predict_mode:
        test_detector(args.predict_mode, model, data, args.results, args.params, device_render=args.device_render,
                      thread_num=args.thread_num)
    else:
        raise NotImplementedError("predict mode '%s' (segmentation is outside the detection hot path)" % args.predict_mode)
    return 0


if __name__ == "__main__":
    sys.exit(main())
