"""Data-parallel training with TWO REAL RANKS of the real model on one GPU (both processes share cuda:0, torch.distributed
`gloo` carries the CUDA tensors), against an in-process emulation of the same exchange.

Reference: denet/model/train_multi.py:96-145 + denet/multi/shared.py:105-119 (every worker trains on its own batches with
its own batch-norm statistics, then all update targets are averaged). The build all-reduces gradients inside the step
(batch_size_factor 1; equal to parameter averaging for nesterov) or averages the state every F steps
(batch_size_factor F > 1, the reference's own scheme).

Emulation (one process, two replicas with identical initial state): per step replica 1 runs forward + backward on its shard
and its gradient / running statistics are captured before the solver; replica 0 runs its step with a `dist` object that adds
the captured tensors where the all-reduce would (sum of two ranks: the same fp32 additions as the collective) and lets the
solver apply 1/2; replica 1 then adopts replica 0's state. Per-rank batch-norm batch statistics by construction. The two-
process result must be IDENTICAL on both ranks and equal to the emulation (<= 1e-6; bit-identical in practice).
"""
import os
import random
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B, IMG, STEPS = 2, 128, 2

WORKER = r'''
import os, random, sys
sys.path.insert(0, %(root)r)
import numpy, torch
torch.cuda.set_device(0)
from denet_amd.model import zoo
from denet_amd.multi import DataParallel
mode, out = sys.argv[1], sys.argv[2]
rank = int(os.environ["RANK"])
dp = DataParallel(backend="gloo")
model = zoo.warm_corner_head(zoo.denet34(%(B)d, "skip", %(IMG)d, class_num=80, seed=1 + 5 * rank), 4.0, 0.3)   # ranks start DIFFERENT:
model.build_train_func("nesterov")                                                                  # broadcast_state must fix it
dp.broadcast_state(model)
x, metas = zoo.synthetic_batch(%(B)d, %(IMG)d, seed=11 + rank)
random.seed(100 + rank)
costs = []
if mode == "grad":
    model.dist = dp
    for it in range(%(STEPS)d):
        c, _ = model.train_step(x, metas, 0, it, 0.02, [0.9], 1e-4)
        costs.append(c)
else:
    for it in range(%(STEPS)d):
        c, _ = model.train_step(x, metas, 0, it, 0.02, [0.9], 1e-4)
        costs.append(c)
    dp.average_state(model)
torch.cuda.synchronize()
torch.save({"P": model.P.cpu(), "M": model.M.cpu(), "S": model.S.cpu(), "costs": costs}, out)
dp.barrier()
'''


class _Abort(Exception):
    pass


class _Capture:
    """dist of replica 1: keeps the gradient and the running statistics at the point where the collectives would run, then
    aborts the step before the solver"""
    world_size = 2

    def begin_step(self, model):
        pass

    def layer_done(self, model, layer):
        pass

    def finish_step(self, model):
        torch.cuda.synchronize()
        self.G, self.S = model.G.clone(), model.S.clone()
        raise _Abort()


class _Merge:
    """dist of replica 0: the sum of the two ranks where the all-reduce would put it; 1/2 of the running statistics"""
    world_size = 2

    def __init__(self, other):
        self.other = other

    def begin_step(self, model):
        pass

    def layer_done(self, model, layer):
        pass

    def finish_step(self, model):
        model.G[:model.n_trainable] += self.other.G[:model.n_trainable]
        model.S += self.other.S
        model.S *= 0.5


def _run_two_ranks(tmp_path, mode):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT, "B": B, "IMG": IMG, "STEPS": STEPS})
    port = 29700 + (os.getpid() % 200) + (7 if mode == "grad" else 0)
    procs, outs = [], []
    for r in range(2):
        out = str(tmp_path / ("rank%d_%s.pt" % (r, mode)))
        outs.append(out)
        # the product default: a geometry outside denet_amd/tuned/gfx950.json (128x128, B = 2 here) runs ops.static_policy and is
        # never measured, so both ranks and the emulation in this process run the same kernels - bit-identical results
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.pop("DENET_AUTOTUNE", None)
        env.pop("DENET_TUNE", None)
        procs.append(subprocess.Popen([sys.executable, str(script), mode, out], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    for p in procs:
        o, _ = p.communicate(timeout=600)
        assert p.returncode == 0, o.decode()[-3000:]
    return [torch.load(o) for o in outs]


@pytest.fixture
def heuristic_kernels():
    """(name kept) the PRODUCT DEFAULT in this process too: committed decisions where the file has the geometry, ops.static_policy
    elsewhere, nothing measured - what the two worker processes run; the decision table is restored afterwards"""
    from denet_amd import ops
    ops._load_tuned_once()
    saved = (ops.AUTOTUNE, dict(ops._WINO), set(ops._TUNED), ops.MEASURE, ops.POLICY)
    ops.AUTOTUNE, ops.MEASURE, ops.POLICY = True, False, None
    yield
    ops.AUTOTUNE, ops.MEASURE, ops.POLICY = saved[0], saved[3], saved[4]
    ops._WINO.clear()
    ops._WINO.update(saved[1])


def _replicas():
    from denet_amd.model import zoo
    reps = []
    for r in range(2):
        m = zoo.warm_corner_head(zoo.denet34(B, "skip", IMG, class_num=80, seed=1), 4.0, 0.3)    # rank 0's initial state on both
        m.build_train_func("nesterov")
        x, metas = zoo.synthetic_batch(B, IMG, seed=11 + r)
        random.seed(100 + r)
        reps.append({"m": m, "x": x, "metas": metas, "rng": random.getstate()})
    reps[1]["m"].P.copy_(reps[0]["m"].P)
    return reps


def _step(rep, it):
    random.setstate(rep["rng"])
    try:
        return rep["m"].train_step(rep["x"], rep["metas"], 0, it, 0.02, [0.9], 1e-4)[0]
    finally:
        rep["rng"] = random.getstate()


def test_two_rank_gradient_allreduce_equals_emulation(hip, tmp_path, heuristic_kernels):
    got = _run_two_ranks(tmp_path, "grad")
    for k in ("P", "M", "S"):
        assert torch.equal(got[0][k], got[1][k]), "ranks diverged in " + k
    reps = _replicas()
    cap = _Capture()
    reps[1]["m"].dist = cap
    reps[0]["m"].dist = _Merge(cap)
    costs = [[], []]
    for it in range(STEPS):
        with pytest.raises(_Abort):
            _step(reps[1], it)
        costs[0].append(_step(reps[0], it))
        torch.cuda.synchronize()
        for name in ("P", "M", "S"):
            getattr(reps[1]["m"], name).copy_(getattr(reps[0]["m"], name))
        from denet_amd import ops
        ops.bump_weights_version()
    m0 = reps[0]["m"]
    for k in ("P", "M", "S"):
        ref = getattr(m0, k).cpu()
        err = float((got[0][k] - ref).abs().max() / (ref.abs().max() + 1e-30))
        assert err <= 1e-6, "%s differs from the emulated two-rank step: %.2e" % (k, err)
    assert abs(got[0]["costs"][-1] - costs[0][-1]) <= 1e-5 * abs(costs[0][-1])
    # and the exchange mattered: a single rank training alone ends elsewhere
    solo = _replicas()[0]
    for it in range(STEPS):
        _step(solo, it)
    assert float((solo["m"].P.cpu() - got[0]["P"]).abs().max()) > 1e-5


def test_two_rank_state_averaging_equals_emulation(hip, tmp_path, heuristic_kernels):
    """--batch-size-factor F > 1: F local steps per rank, then parameters / momentum / running statistics averaged"""
    got = _run_two_ranks(tmp_path, "avg")
    for k in ("P", "M", "S"):
        assert torch.equal(got[0][k], got[1][k]), "ranks diverged in " + k
    reps = _replicas()
    for it in range(STEPS):
        for r in range(2):
            _step(reps[r], it)
    torch.cuda.synchronize()
    for k in ("P", "M", "S"):
        ref = ((getattr(reps[0]["m"], k) + getattr(reps[1]["m"], k)) * 0.5).cpu()
        err = float((got[0][k] - ref).abs().max() / (ref.abs().max() + 1e-30))
        assert err <= 1e-6, "%s differs from the emulated averaging: %.2e" % (k, err)


def test_bench_self_launches_two_ranks_on_a_shared_gpu():
    """LAUNCH-PATH test, never a scaling measurement. `python bench.py --gpus 2` WITHOUT a launcher must start its own two
    ranks (the reference's driver spawns its workers itself, denet/model/train_multi.py:96-145, denet/multi/worker.py:138-243),
    rendezvous on a free local port, run the bucketed gradient exchange inside the step and have rank 0 print ONE JSON line
    whose n_gpus is the process group's world size. A one-GPU box cannot host two RCCL ranks, so `--share-gpu` puts both
    ranks on cuda:0 and lets gloo carry the tensors; everything else (full-size DeNet-34 skip, B = 32 per rank, buckets,
    event-timed exposed collective time, max over ranks) is the code path of an 8-GPU run."""
    import json
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--share-gpu"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=840)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 64 and d["config"]["parallelism"] == "dp2"
    assert d["value"] > 0 and d["ms_per_step"] > 0
    dp = d["data_parallel"]
    assert dp["world_size"] == 2 and dp["backend"] == "gloo" and "launch_path_test_only" in dp
    assert len(dp["ms_per_step_per_rank"]) == 2 and len(dp["exposed_collective_ms_per_step_per_rank"]) == 2
    # the whole flat gradient (~137 MB for DeNet-34 skip) + the BN running statistics cross the exchange every step
    # exactly one collective per bucket (five for DeNet-34 skip): the bias / BN-affine gradients and the BN running statistics
    # ride in the last bucket's packed collective, none is issued behind the backward sweep
    assert dp["allreduce_bytes_per_step"] > 100e6 and dp["collectives_per_step"] == len(dp["bucket_bytes"]) <= 5, dp
    assert abs(sum(dp["bucket_bytes"]) + 0 - dp["allreduce_bytes_per_step"]) < 0.05 * dp["allreduce_bytes_per_step"]
    # per bucket: issued from inside the backward sweep (the head's bucket first, with the longest window before the sweep ends; the
    # last one when the sweep has passed the first layer), and what the compute stream still waited for it after the sweep
    ov = dp["bucket_overlap"]
    assert [r["bytes"] for r in ov] == dp["bucket_bytes"], ov
    win = [r["issued_before_sweep_end_ms"] for r in ov]
    assert all(a >= b for a, b in zip(win, win[1:])) and win[0] > 5.0 and win[-1] >= 0.0, ov
    assert all(r["waited_after_sweep_end_ms"] >= 0.0 for r in ov)
    for k in ("roofline", "cpu_baseline", "warm_regime"):          # rank 0 at N = 1 only
        assert k not in d
