"""`model-train-multi`: data-parallel training over the GPUs of one node. Replaces denet/model/train_multi.py + denet/multi
(one Python process with a worker thread per GPU, parameters averaged through host shared memory, :96-145) by one process
per GPU under `torch.distributed.run`, gradients all-reduced over RCCL while the backward pass runs
(denet_amd/multi.DataParallel; DESIGN.md §7 shows the equivalence for sgd / nesterov at batch_size_factor 1):

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        -m denet_amd.model.train_multi --train DIR --extension mscoco,2014-train,crop=512,crop_mode=denet ... --device-render

Same flags as model-train plus `--batch-size-factor F`. Like the reference (train_multi.py:44-46) every rank shuffles the
image list with the seed `seed + epoch`, so all ranks see the same order. One global iteration consumes `F x world` batches
in the reference's order (train_multi.py:113-119: `for worker: for _ in range(F): index += 1`), i.e. worker r takes the F
CONSECUTIVE batches `r * F ... r * F + F - 1` of the iteration. Like the reference's export(batch_size = world x B x F)
(train_multi.py:53-55, dataset/__init__.py:349-354) the last partial iteration of a subset is padded with samples drawn by
`random.randint(0, len - 1)` - every rank draws all of them, so the random streams stay in step.
  F = 1: gradients are all-reduced inside every step (equal to the reference's parameter averaging for sgd / nesterov);
  F > 1: what the shipped recipes run (papers/dss/denet34.sh:43 `--batch-size-factor 2` without --use-acc-mode): every rank
         takes F full local steps (with `--use-acc-mode`: F accumulated steps from the same parameters, one averaged update,
         model_cnn.py:374-392 / worker.py:60-116), then parameters, momentum and BN statistics are averaged over the ranks
         (DataParallel.average_state) - the reference's scheme itself, over RCCL instead of host shared memory.
Rank 0 writes the checkpoints with the reference's names: `<prefix>_epochNNN_final.mdl.gz` after every epoch
(train_multi.py:166) - the reference's timed `_epochNNN_subsetMMM` files are written with `--save-subsets` after every subset.
`--epoch-start E` skips the first E epochs (the learning rate is annealed for them, train_multi.py:406-410; the shuffle seed is
seed + epoch, so the data order of the remaining epochs is unchanged); `--restart` finds the newest `<prefix>_epoch*.mdl.gz`
like load_restart_args (train_multi.py:242-268) and continues behind it. The momentum buffer is not part of a .mdl.gz (as in
the reference): a restarted run begins with zero momentum."""
import math
import os
import random
import sys

import numpy

from . import model_cnn
from . import train as train_mod


class _Shard:
    """the rank's slices of a dataset's image list, with the loader / attributes the training loop needs"""

    def __init__(self, data, rank, world, batch_size, factor=1):
        self.data, self.rank, self.world, self.batch_size, self.factor = data, rank, world, batch_size, factor

    def images_of_subset(self, subset):
        d = self.data
        lo = subset * d.subset_size
        hi = min((subset + 1) * d.subset_size, d.subset_total_size)
        images = list(d.images[lo:hi])
        B = self.batch_size
        per_it = self.world * self.factor             # batches of one global iteration
        n = len(images)
        if n == 0:
            return []
        n_it = math.ceil(n / (per_it * B))
        # export() pads to a multiple of world x B x F with random samples of the subset (dataset/__init__.py:350-354)
        images += [images[random.randint(0, n - 1)] for _ in range(n_it * per_it * B - n)]
        mine = []
        for it in range(n_it):
            for f in range(self.factor):
                k = it * per_it + self.rank * self.factor + f
                mine += images[k * B:(k + 1) * B]
        return mine


def train(args, train_data, dp, log=None):
    if log is None:
        from ..common import logging
        log = logging.info
    model = model_cnn.initialize(args, train_data.get_data_shape(), train_data.class_labels, train_data.get_class_num())
    factor = max(1, int(getattr(args, "batch_size_factor", 1)))
    use_acc_mode = factor > 1 and bool(getattr(args, "use_acc_mode", False))          # worker.py:60
    model.build_train_func(args.solver, args.cost_factors, use_acc_mode=use_acc_mode)
    if dp is not None:
        if factor == 1:
            model.dist = dp                 # gradient all-reduce inside every step
        dp.broadcast_state(model)
    rank = dp.rank if dp is not None else 0
    world = dp.world_size if dp is not None else 1
    shard = _Shard(train_data, rank, world, model.batch_size, factor)
    loader = None
    if getattr(args, "device_render", False):
        from ..dataset.device_render import DeviceImageLoader
        loader = DeviceImageLoader(max(1, args.thread_num), True, cp=model.input.cp, decode="process",
                                   params=train_data.image_loader)
    learn_rate = args.learn_rate
    costs = []
    epoch_start = int(getattr(args, "epoch_start", 0))
    for epoch in range(0, epoch_start):                 # train_multi.py:406-410
        if len(args.learn_anneal_epochs) == 0 or (epoch + 1) in args.learn_anneal_epochs:
            learn_rate *= args.learn_anneal
    for epoch in range(epoch_start, args.epochs):
        random.seed(args.seed + epoch)                 # same order on every rank (train_multi.py:44-46)
        train_data.shuffle()
        # a restarted run continues the first epoch behind the subsets the checkpoint has seen (UpdateClient(epoch_start,
        # subset_start, ...), train_multi.py:396-398); the shuffle above is the whole epoch's, so the data order is unchanged
        first = int(getattr(args, "subset_start", 0)) if epoch == epoch_start else 0
        for subset in range(first, train_data.subset_num):
            images = shard.images_of_subset(subset)
            if len(images) == 0:
                continue
            if loader is not None:
                batches = loader.iterate(images, model.batch_size)
            else:
                train_data.data = train_data.image_loader.load(images)
                dx, dm, n = train_data.export(model.batch_size)
                batches = ((dx[i:i + model.batch_size], dm[i:i + model.batch_size]) for i in range(0, n, model.batch_size))
            cost = 0.0
            for step, (data_x, data_m) in enumerate(batches):
                if use_acc_mode and step % factor == 0:
                    model.train_begin()                                  # worker.py:95-97
                c, _ = model.train_step(data_x, data_m, epoch, model.iteration, learn_rate, args.learn_momentum, args.learn_decay)
                if math.isnan(c):
                    raise Exception("ERROR: Cost is NaN")
                cost += c
                model.iteration += 1
                if use_acc_mode and (step + 1) % factor == 0:
                    model.train_end()                                    # worker.py:114-116
                if factor > 1 and dp is not None and (step + 1) % factor == 0:
                    dp.average_state(model)
            costs.append(cost)
            if rank == 0:
                log("epoch %i subset %i - cost (rank 0): %.4f (lr %g, %i GPUs)" % (epoch, subset, cost, learn_rate, world))
                if getattr(args, "save_subsets", False) and not args.disable_intermediate:
                    model_cnn.save_to_file(model, args.output_prefix + "_epoch%03i_subset%03i.mdl.gz" % (epoch, subset + 1))
        if len(args.learn_anneal_epochs) == 0 or (epoch + 1) in args.learn_anneal_epochs:
            learn_rate *= args.learn_anneal
        if rank == 0 and (not args.disable_intermediate or epoch == args.epochs - 1):
            model_cnn.save_to_file(model, args.output_prefix + "_epoch%03i_final.mdl.gz" % epoch)     # train_multi.py:166
    if loader is not None:
        loader.close()
    return model, costs


def find_restart(output_prefix):
    """(model file, epoch_start, subset_start) of the newest checkpoint of a run, with the arithmetic of load_restart_args
    (train_multi.py:242-268): a `_final` file continues with the next epoch at subset 0; a `_subsetMMM` file continues ITS epoch
    at subset MMM + 1. (The file number is already the finished subset + 1, train_multi.py:158, so the reference steps over one
    subset when it restarts; mirrored as it is.)"""
    import glob
    files = sorted(glob.glob(output_prefix + "_epoch*.mdl.gz"))
    if not files:
        raise Exception("Could not find any intermediate models to continue training from!")
    v = os.path.basename(files[-1])
    v = v[:v.find(".")].split("_")
    if v[-1] == "final":
        return files[-1], int(v[-2][5:]) + 1, 0
    return files[-1], int(v[-2][5:]), int(v[-1][6:]) + 1


def main(argv=None):
    import torch
    parser = train_mod.build_parser()
    parser.add_argument("--batch-size-factor", type=int, default=1,
                        help="local training steps per rank between two parameter averagings (1: gradient all-reduce every step)")
    parser.add_argument("--use-acc-mode", default=False, action="store_true",
                        help="Use model accumulation over multiple batches (with --batch-size-factor F > 1: the F local steps "
                             "of an iteration all start from the same parameters and their updates are averaged)")
    parser.add_argument("--epoch-start", type=int, default=0, help="Epoch to start from")
    parser.add_argument("--subset-start", type=int, default=0, help="Subset to start from")
    parser.add_argument("--restart", default=False, action="store_true", help="Restart training of model")
    parser.add_argument("--save-subsets", default=False, action="store_true", help="checkpoint after every subset")
    args = parser.parse_args(argv)
    if args.restart:
        args.model, args.epoch_start, args.subset_start = find_restart(args.output_prefix)
    from ..common import logging
    logging.init(args)
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
