"""`model-train-multi`: data-parallel training over the GPUs of one node. Replaces denet/model/train_multi.py + denet/multi
(one Python process with a worker thread per GPU, parameters averaged through host shared memory, :96-145) by one process
per GPU under `torch.distributed.run`, gradients all-reduced over RCCL while the backward pass runs
(denet_amd/multi.DataParallel; DESIGN.md §7 shows the equivalence for sgd / nesterov at batch_size_factor 1):

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        -m denet_amd.model.train_multi --train DIR --extension mscoco,2014-train,crop=512,crop_mode=denet ... --device-render

Same flags as model-train. Like the reference (train_multi.py:44-46) every rank shuffles the image list with the seed
`seed + epoch`, so all ranks see the same order; a global batch is `world x batch_size` consecutive samples of which rank r
takes the r-th slice. BN statistics stay per GPU; rank 0 writes the checkpoints."""
import math
import os
import random
import sys

import numpy

from . import model_cnn
from . import train as train_mod


class _Shard:
    """the rank's slices of a dataset's image list, with the loader / attributes the training loop needs"""

    def __init__(self, data, rank, world, batch_size):
        self.data, self.rank, self.world, self.batch_size = data, rank, world, batch_size

    def images_of_subset(self, subset):
        d = self.data
        lo = subset * d.subset_size
        hi = min((subset + 1) * d.subset_size, d.subset_total_size)
        images = d.images[lo:hi]
        g = self.world * self.batch_size
        n_glob = math.floor(len(images) / g)          # whole global batches only: every rank runs the same step count
        mine = []
        for k in range(n_glob):
            mine += images[k * g + self.rank * self.batch_size:k * g + (self.rank + 1) * self.batch_size]
        return mine


def train(args, train_data, dp, log=print):
    model = model_cnn.initialize(args, train_data.get_data_shape(), train_data.class_labels, train_data.get_class_num())
    model.build_train_func(args.solver, args.cost_factors)
    if dp is not None:
        model.dist = dp
        dp.broadcast_state(model)
    rank = dp.rank if dp is not None else 0
    world = dp.world_size if dp is not None else 1
    shard = _Shard(train_data, rank, world, model.batch_size)
    loader = None
    if getattr(args, "device_render", False):
        from ..dataset.device_render import DeviceImageLoader
        loader = DeviceImageLoader(max(1, args.thread_num), True, cp=model.input.cp, decode="process",
                                   params=train_data.image_loader)
    learn_rate = args.learn_rate
    costs = []
    for epoch in range(args.epochs):
        random.seed(args.seed + epoch)                 # same order on every rank (train_multi.py:44-46)
        train_data.shuffle()
        for subset in range(train_data.subset_num):
            images = shard.images_of_subset(subset)
            if len(images) == 0:
                continue
            if loader is not None:
                cost = model.train_epoch_device(loader, images, epoch, learn_rate, args.learn_momentum, args.learn_decay)
            else:
                train_data.data = train_data.image_loader.load(images)
                cost = model.train_epoch(train_data, epoch, learn_rate, args.learn_momentum, args.learn_decay)
            costs.append(cost)
            if rank == 0:
                log("epoch %i subset %i - cost (rank 0): %.4f (lr %g, %i GPUs)" % (epoch, subset, cost, learn_rate, world))
        if len(args.learn_anneal_epochs) == 0 or (epoch + 1) in args.learn_anneal_epochs:
            learn_rate *= args.learn_anneal
        if rank == 0 and not args.disable_intermediate:
            model_cnn.save_to_file(model, args.output_prefix + "_epoch%03i.mdl.gz" % epoch)
    if loader is not None:
        loader.close()
    if rank == 0:
        model_cnn.save_to_file(model, args.output_prefix + "_epoch%03i_final.mdl.gz" % (args.epochs - 1))
    return model, costs


def main(argv=None):
    import torch
    args = train_mod.build_parser().parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dp = None
    if world > 1 or os.environ.get("DENET_FORCE_DP") == "1":
        from ..multi import DataParallel
        dp = DataParallel(backend="nccl")
        dp.force_collectives = os.environ.get("DENET_FORCE_DP") == "1"
    random.seed(args.seed)
    numpy.random.seed(args.seed)                       # identical initial weights on every rank (+ broadcast)
    data = train_mod.load_dataset(args.train, args.seed, args.extension, True, args.thread_num)
    if not hasattr(data, "images"):
        raise SystemExit("model-train-multi shards the image list of an MSCOCO / Pascal VOC / ImageNet dataset")
    train(args, data, dp)
    if hasattr(data, "image_loader"):
        data.image_loader.close()
    if dp is not None:
        dp.barrier()
        dp.dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
