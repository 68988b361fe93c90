"""model-train — single-GPU training driver with the CLI flag surface of the reference's denet/model/train.py
(:46-156): --model / --model-desc, --solver, --learn-rate, --learn-momentum, --learn-decay, --learn-anneal,
--learn-anneal-epochs, --batch-size, --seed, --border-mode, --activation, --weight-init, --cost-factors, --epochs,
--output-prefix. The reference's dataset loaders are outside the hot path; `--train synthetic[,k=v]`
feeds MSCOCO-shaped synthetic batches with the same meta-dict contract (image_loader.py:134-136), and any object
with `export(batch_size) -> (x, metas, n)` (dataset/__init__.py:349-366) can be passed to train() directly."""
import argparse
import math
import random
import sys

import numpy

from . import model_cnn
from . import zoo


class ArrayDataset:
    """In-memory dataset of (C,H,W) float arrays + meta dictionaries with the surface ModelCNN.train_epoch and the
    training driver use from the reference's DatasetAbstract (denet/dataset/__init__.py:43-366): len, get_data_shape,
    get_class_num, shuffle, load_from_subset, export. export() pads the last batch with samples drawn by
    random.randint like the reference (:349-366) - the draws are part of the stdlib stream the RoI editing shares."""

    def __init__(self, data_x, metas, class_num):
        self._x = numpy.ascontiguousarray(data_x, dtype=numpy.float32)
        self._m = list(metas)
        self.class_num = class_num
        self.class_labels = {"class%i" % i: i for i in range(class_num)}
        self.subset_num = 1

    def __len__(self):
        return len(self._m)

    def get_data_shape(self):
        return tuple(self._x.shape[1:])

    def get_class_num(self):
        return self.class_num

    def get_metas(self):
        return list(self._m)

    def shuffle(self, mode="random"):
        if mode != "random":
            raise Exception("Unknown shuffle mode:", mode)
        order = list(range(len(self)))
        random.shuffle(order)       # same draws and same permutation as random.shuffle(self.data)
        self._x = self._x[order]
        self._m = [self._m[i] for i in order]

    def load_from_subset(self, subset):
        pass

    def export(self, batch_size=1, dtype=numpy.float32):
        n = len(self)
        size = batch_size * int(math.ceil(n / batch_size))
        index = list(range(n)) + [random.randint(0, n - 1) for _ in range(size - n)]
        x = self._x if size == n else self._x[index]
        return x.astype(dtype, copy=False), [self._m[i] for i in index], n


class SyntheticDataset(ArrayDataset):
    """`samples` synthetic images with boxes/classes (there is no network for the real datasets)"""

    def __init__(self, samples=64, image=512, class_num=80, channels=3, seed=1):
        self.samples, self.image, self.channels, self.seed = samples, image, channels, seed
        x, m = zoo.synthetic_batch(samples, image, class_num, seed)
        super().__init__(x, m, class_num)


def load_dataset(spec, seed, extension="ppm", is_training=True, thread_num=1, class_labels=None):
    """`synthetic[,samples=N,image=S,classes=C]` (there is no network for the real datasets), or a directory read by
    denet_amd.dataset.load with the reference's format string (`--extension mscoco,2014-train,crop=512,...`)"""
    parts = spec.split(",")
    if parts[0] != "synthetic":
        from .. import dataset
        return dataset.load(spec, extension, is_training=is_training, thread_num=thread_num, class_labels=class_labels)
    kw = {}
    for p in parts[1:]:
        k, v = p.split("=")
        kw[{"samples": "samples", "image": "image", "classes": "class_num"}[k]] = int(v)
    return SyntheticDataset(seed=seed, **kw)


def build_parser():
    parser = argparse.ArgumentParser(description="Train a convolutional network (MI355X hot path of lachlants/denet)")
    from ..common import logging
    logging.add_arguments(parser)
    parser.add_argument("--model", required=False, default=None, help="Model (.mdl.gz) to continue training.")
    parser.add_argument("--cost-factors", default=[], nargs="+", help="Multiplicative factors for model costs")
    parser.add_argument("--train", default="synthetic",
                        help="training data folder, or synthetic[,samples=N,image=S,classes=C]")
    parser.add_argument("--test", default=None, help="The folder with testing data (optional)")
    parser.add_argument("--test-epochs", type=int, default=1, help="Epochs between each test evaluation")
    parser.add_argument("--extension", default="ppm", help="Image file extension / dataset format string")
    parser.add_argument("--thread-num", type=int, default=1, help="Worker processes for loading / augmenting data")
    parser.add_argument("--max-samples", type=int, default=None, help="Maximum samples to load from training set")
    parser.add_argument("--augment-mirror", default=False, action="store_true")
    parser.add_argument("--device-render", default=False, action="store_true",
                        help="render the augmentation (crop, resampling, colour jitter) on the GPU instead of in the loader "
                             "processes: same batches, one host process per GPU keeps up with training")
    parser.add_argument("--border-mode", default="valid")
    parser.add_argument("--output-prefix", default="./model")
    parser.add_argument("--activation", default="relu")
    parser.add_argument("--solver", type=str, default="nesterov")
    parser.add_argument("--weight-init", nargs="+", default=["he-backward"])
    parser.add_argument("--learn-rate", type=float, default=0.1)
    parser.add_argument("--learn-momentum", type=float, default=[0.0, 0.0], nargs="+")
    parser.add_argument("--learn-anneal", type=float, default=1)
    parser.add_argument("--learn-anneal-epochs", nargs="+", type=int, default=[])
    parser.add_argument("--learn-decay", type=float, default=0.0)
    parser.add_argument("--epochs", type=int, default=30)
    parser.add_argument("--batch-size", type=int, default=32)
    parser.add_argument("--seed", type=int, default=23455)
    parser.add_argument("--disable-intermediate", default=False, action="store_true")
    parser.add_argument("--skip-layer-updates", type=int, nargs="+", default=[])
    parser.add_argument("--model-desc", default=["C[100,7]", "P[2]", "C[150,4]", "P[2]", "C[250,4]", "P[2]", "C[300,1]", "R"],
                        nargs="+", type=str)
    return parser


def compute_error(data, model):
    """per-class top-1 error of a classifier over all subsets (train.py:18-39) -> (error %, [(class, error %, samples)])"""
    class_errors = [0] * model.class_num
    class_samples = [0] * model.class_num
    for subset in range(data.subset_num):
        data.load_from_subset(subset)
        predicted = model.predict_label(data)
        labels = data.get_labels()
        for i in range(len(data)):
            class_samples[labels[i]] += 1
            if predicted[i] != labels[i]:
                class_errors[labels[i]] += 1
    error = 100.0 * sum(class_errors) / sum(class_samples)
    return error, [(i, 100.0 * class_errors[i] / class_samples[i] if class_samples[i] else 0.0, class_samples[i])
                   for i in range(model.class_num)]


def save_results(fname, error, class_errors):
    with open(fname, "w") as f:
        print("Overall Error=%.2f%%" % error, file=f)
        for d in class_errors:
            print("Class %i=%.2f%% (%i samples)" % (d[0], d[1], d[2] * d[1] / 100), file=f)


def _default_log():
    from ..common import logging
    return logging.info


def train(args, train_data, log=None, test_data=None):
    """the epoch loop of train.py:117-151 (shuffle, train_epoch, learning-rate annealing, checkpoint per epoch)"""
    log = _default_log() if log is None else log
    model = model_cnn.initialize(args, train_data.get_data_shape(), train_data.class_labels, train_data.get_class_num())
    model.build_train_func(args.solver, args.cost_factors)
    learn_rate = args.learn_rate
    costs = []
    device_loader = None
    if getattr(args, "device_render", False) and hasattr(train_data, "images") and hasattr(train_data, "image_loader"):
        from ..dataset.device_render import DeviceImageLoader
        device_loader = DeviceImageLoader(max(1, getattr(args, "thread_num", 1)), True, cp=model.input.cp, decode="process",
                                          params=train_data.image_loader)
    for epoch in range(args.epochs):
        train_data.shuffle()
        for subset in range(train_data.subset_num):
            if device_loader is not None:
                lo = subset * train_data.subset_size
                hi = min((subset + 1) * train_data.subset_size, train_data.subset_total_size)
                cost = model.train_epoch_device(device_loader, train_data.images[lo:hi], epoch, learn_rate,
                                                args.learn_momentum, args.learn_decay)
            else:
                train_data.load_from_subset(subset)
                cost = model.train_epoch(train_data, epoch, learn_rate, args.learn_momentum, args.learn_decay)
            costs.append(cost)
            log("epoch %i subset %i - cost: %.4f (lr %g)" % (epoch, subset, cost, learn_rate))
        if len(args.learn_anneal_epochs) == 0 or (epoch + 1) in args.learn_anneal_epochs:
            learn_rate *= args.learn_anneal
        if test_data is not None and ((epoch % getattr(args, "test_epochs", 1)) == 0 or epoch == (args.epochs - 1)):
            test_error, test_class_errors = compute_error(test_data, model)
            log("epoch %i test error: %.2f%%" % (epoch, test_error))
            save_results(args.output_prefix + "_epoch%03i.test" % epoch, test_error, test_class_errors)
        if not args.disable_intermediate:
            model_cnn.save_to_file(model, args.output_prefix + "_epoch%03i.mdl.gz" % epoch)
    if device_loader is not None:
        device_loader.close()
    model_cnn.save_to_file(model, args.output_prefix + "_epoch%03i_final.mdl.gz" % (args.epochs - 1))
    return model, costs


def main(argv=None):
    args = build_parser().parse_args(argv)
    from ..common import logging
    logging.init(args)
    random.seed(args.seed)
    numpy.random.seed(args.seed)
    train_data = load_dataset(args.train, args.seed, args.extension, True, args.thread_num)
    if args.max_samples is not None:
        train_data.data = random.sample(train_data.data, args.max_samples)
    if args.augment_mirror:
        train_data.augment_mirror()
    test_data = None
    if args.test:
        test_data = load_dataset(args.test, args.seed, args.extension, False, args.thread_num, train_data.class_labels)
    train(args, train_data, test_data=test_data)
    return 0


if __name__ == "__main__":
    sys.exit(main())
