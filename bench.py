#!/usr/bin/env python
"""bench.py — images/sec of one DeNet-34 skip training step at 512x512 (BASELINE.json metric) on N MI355X.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
         bench.py --gpus N --steps K --warmup W

A "step" = one full training step of the hot path on one synthetic MSCOCO-shaped batch of 32 images per GPU:
host targets -> forward (conv/BN stack, corner map, GPU RoI proposal, RoI editing, sparse gather, RoI head) ->
costs -> backward -> (N>1: RCCL gradient all-reduce overlapped with backward) -> fused nesterov update.
Inputs are resident in HBM before the timed region. Rank 0 prints ONE JSON line.

Extra legs (rank 0, N=1 only; outside the timed region):
  roofline      the dominant kernel (implicit-GEMM convolution instantiation with the largest total time) is timed
                live with HIP events on the launch stream over `steps` further instrumented steps;
                achieved = algorithmic FLOPs per launch / average launch duration; peak = fp32 MFMA 157.3 TFLOP/s.
  cpu_baseline  the numpy/C++ oracle (oracle/, a CPU restatement of the reference path, kind "port") runs ONE
                training step of the same model at batch 1 on the host cores.
"""
import argparse
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_IMAGE_STEP = 164.3e9     # 3 x 2 x 27.38 GMAC (conv/GEMM only, no recompute)
PEAK_FP32_MFMA_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
BATCH_PER_GPU = 32                # papers/dss/denet34.sh:43 --batch-size 32


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE need
    separate passes, so they cannot be collected inside this run): profiles/rNN_*_pmc_traffic.json, written by
    tools/pmc_traffic.py from the same bench command. None if no such summary is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None
    with open(files[-1]) as f:
        k = json.load(f)["kernels"].get(kernel)
    if not k or k.get("write_bytes_per_launch") is None:
        return None
    return round(k["fetch_bytes_per_launch"] + k["write_bytes_per_launch"])


def host_cores():
    """cores the CPU baseline can really use: the affinity mask clipped by the cgroup CPU quota"""
    n = len(os.sched_getaffinity(0))
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-warm", action="store_true", help="skip the warm-corner-regime leg (rank 0, N=1)")
    ap.add_argument("--no-split-bf16", action="store_true",
                    help="skip the leg of the OPT-IN variant (head GEMMs as 3-term bf16 splits; own key, never the headline)")
    ap.add_argument("--regime", default="cold", choices=["cold", "warm"],
                    help="cold = weights as initialised (corner bias +5: no detector RoIs, SURVEY §8d); warm = corner "
                         "head re-biased so that ~1%% of the cells fire")
    args = ap.parse_args()

    import numpy
    import torch
    from denet_amd import ops
    from denet_amd.model import zoo

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs the torch.distributed.run launcher (one rank per GPU)" % args.gpus)
    torch.cuda.set_device(local_rank)
    ops.HEAD_BF16X3 = False      # the headline is fp32 MFMA whatever the environment says; the opt-in variant has its own leg below
    dp = None
    if world > 1 or os.environ.get("DENET_FORCE_DP") == "1":      # DENET_FORCE_DP: exercise the RCCL path on 1 GPU
        from denet_amd.multi import DataParallel
        dp = DataParallel(backend="nccl")
        dp.force_collectives = os.environ.get("DENET_FORCE_DP") == "1"

    # identical initial weights on every rank (seed), per-rank data shard (seed + rank)
    model = zoo.denet34(BATCH_PER_GPU, "skip", 512, class_num=80, seed=1)
    if args.regime == "warm":
        zoo.warm_corner_head(model)
    model.build_train_func("nesterov")
    if dp is not None:
        model.dist = dp
        dp.broadcast_state(model)
    x, metas = zoo.synthetic_batch(BATCH_PER_GPU, 512, 80, seed=1 + rank)
    xd = torch.from_numpy(x).cuda()
    random.seed(1 + rank)
    lr, mom, decay = 0.1, [0.9], 1e-4     # papers/dss/denet34.sh:43

    def sync():
        if dp is not None:
            dp.barrier()
        torch.cuda.synchronize()

    it = 0
    # launch configurations are measured during the first two steps (one-off setup, like kernel compilation): they are
    # taken out of the timed region even when fewer warm-up steps were asked for
    for _ in range(max(args.warmup, 2)):
        model.train_step(xd, metas, 0, it, lr, mom, decay)
        it += 1
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        cost, _ = model.train_step(xd, metas, 0, it, lr, mom, decay)
        it += 1
    sync()
    dt = time.perf_counter() - t0
    if dp is not None:
        dt = dp.max_over_ranks(dt)
    if not numpy.isfinite(cost):
        raise SystemExit("non-finite cost %r" % cost)
    images = BATCH_PER_GPU * world * args.steps
    value = images / dt

    out = {
        "metric": "images/sec train step, DeNet-34 skip 512x512",
        "value": round(value, 2),
        "unit": "images/sec",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(1e3 * dt / args.steps, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "DeNet-34 skip full DSS head (corner + sparse RoI + classify) 512x512 synthetic "
                               "MSCOCO, full train step (targets, fwd, bwd, nesterov), %s corner regime" % args.regime,
                   "global_batch": BATCH_PER_GPU * world, "batch_per_gpu": BATCH_PER_GPU, "classes": 80,
                   "rois_per_image": 576, "parallelism": "dp%d" % world, "solver": "nesterov",
                   "conv_algorithms": "fp32 throughout; per layer and pass the fastest of the direct implicit GEMM, "
                                      "Winograd F(2x2,3x3)/F(4x4,3x3) and (64 input channels) F(2x2,3x3) fused into one "
                                      "kernel, as measured once by tools/tune.py and stored in "
                                      "denet_amd/tuned/gfx950.json (every process runs the same kernels; geometries not "
                                      "in the file are measured on the first step; DENET_WINOGRAD=0: direct kernels only)",
                   "input": "fp32 NCHW batch resident in HBM before the timed region",
                   "final_cost": round(float(cost), 5)},
        # rate in FLOPs of the reference's direct algorithm (164.3 GFLOP per image and step); layers that run Winograd
        # execute fewer, so this is an effective rate - the MFMA utilisation of the kernels is in `roofline`
        "step_tflops_algorithmic": round(value * FLOP_PER_IMAGE_STEP / 1e12, 2),
        "step_frac_of_fp32_mfma_peak": round(value * FLOP_PER_IMAGE_STEP / 1e12 / (PEAK_FP32_MFMA_TFLOPS * world), 4),
    }

    if rank == 0 and world == 1 and not args.no_roofline:
        prof = ops.KernelProfile()
        ops.PROFILE = prof
        nprof = max(1, min(args.steps, 5))
        for _ in range(nprof):
            model.train_step(xd, metas, 0, it, lr, mom, decay)
            it += 1
        ops.PROFILE = None
        agg = prof.summary()
        name, a = max(agg.items(), key=lambda kv: kv[1]["ms"])
        avg_ms = a["ms"] / a["launches"]
        achieved = (a["flops"] / a["launches"]) / (avg_ms * 1e-3) / 1e12
        out["roofline"] = {"bound": "mfma", "kernel": name, "achieved": round(achieved, 2),
                           "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                           "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": pmc_traffic(name),
                           "launches_per_step": a["launches"] // nprof, "avg_launch_ms": round(avg_ms, 4),
                           "flop_per_launch": round(a["flops"] / a["launches"]),
                           "all_igemm": {k: {"launches_per_step": v["launches"] // nprof,
                                             "ms_per_step": round(v["ms"] / nprof, 3),
                                             "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2)}
                                         for k, v in sorted(agg.items())}}

    if rank == 0 and world == 1 and not args.no_warm and args.regime == "cold":
        # the headline regime has no detector RoIs (cold corner head); the same step with a firing corner head
        # exercises the RoI proposal for real (pair enumeration of a few hundred corners per type, top-576 selection)
        del model
        torch.cuda.empty_cache()
        mw = zoo.warm_corner_head(zoo.denet34(BATCH_PER_GPU, "skip", 512, class_num=80, seed=1))
        mw.build_train_func("nesterov")
        random.seed(1)
        wit = 0
        for _ in range(max(args.warmup, 2)):
            mw.train_step(xd, metas, 0, wit, lr, mom, decay)
            wit += 1
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rois = []
        for _ in range(args.steps):
            wcost, _ = mw.train_step(xd, metas, 0, wit, lr, mom, decay)
            wit += 1
            dns = [l for l in mw.layers if l.type_name == "denet-sparse"][0]
            raw = getattr(dns, "_raw_samples", None)
            rois.append(float(raw[1].mean()) if raw is not None else 0.0)
        torch.cuda.synchronize()
        wdt = time.perf_counter() - t0
        # the RoI proposal alone (corner_select + pair_* kernels) on the last corner map, event-timed on its stream
        dnc = [l for l in mw.layers if l.type_name == "denet-corner"][0]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ops.build_samples(dnc.corner_pr, dns.corner_threshold, dns.proposal_count, 1024, dns.local_max)
        e0.record()
        for _ in range(10):
            ops.build_samples(dnc.corner_pr, dns.corner_threshold, dns.proposal_count, 1024, dns.local_max)
        e1.record()
        torch.cuda.synchronize()
        out["warm_regime"] = {"value": round(BATCH_PER_GPU * args.steps / wdt, 2), "unit": "images/sec",
                              "ms_per_step": round(1e3 * wdt / args.steps, 3),
                              "detector_rois_per_image": round(sum(rois) / len(rois), 1),
                              "roi_proposal_kernels_ms": round(e0.elapsed_time(e1) / 10, 4),
                              "final_cost": round(float(wcost), 5),
                              "note": "same step, DNC corner head re-biased (zoo.warm_corner_head) so that ~1 % of the cells "
                                      "fire; the proposal replaces the reference's host pair search (denet_sparse.cc:337-373)"}

    if rank == 0 and world == 1 and not args.no_split_bf16:
        try:
            # OPT-IN variant (ops.HEAD_BF16X3, csrc/gemm3b.hip): the same step with the 1x1 convolutions of the detection head (fwd,
            # data and filter gradient; 4736 -> 1536 -> 1024 -> 768 -> 512) as 3-term bf16-split GEMMs on the bf16 matrix cores. Its
            # products are NOT the exact fp32 FMA chain (relative error ~1e-6 of a sum): own key, own number.
            torch.cuda.empty_cache()
            costs = {}
            sdt = None
            for flag in (False, True):
                ops.HEAD_BF16X3 = flag
                ms = zoo.denet34(BATCH_PER_GPU, "skip", 512, class_num=80, seed=1)
                ms.build_train_func("nesterov")
                random.seed(1)
                sit = 0
                c1, _ = ms.train_step(xd, metas, 0, sit, lr, mom, decay)       # first step from identical weights: cost fp32 vs split
                costs[flag] = float(c1)
                sit += 1
                if flag:
                    for _ in range(max(args.warmup, 2)):
                        ms.train_step(xd, metas, 0, sit, lr, mom, decay)
                        sit += 1
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(args.steps):
                        scost, _ = ms.train_step(xd, metas, 0, sit, lr, mom, decay)
                        sit += 1
                    torch.cuda.synchronize()
                    sdt = time.perf_counter() - t0
                del ms
                torch.cuda.empty_cache()
            ops.HEAD_BF16X3 = False
            out["split_bf16"] = {"value": round(BATCH_PER_GPU * args.steps / sdt, 2), "unit": "images/sec",
                                 "ms_per_step": round(1e3 * sdt / args.steps, 3), "dtype": "f32 storage; head GEMM products bf16 x 3",
                                 "first_step_cost_fp32": round(costs[False], 6), "first_step_cost_split": round(costs[True], 6),
                                 "final_cost": round(float(scost), 5),
                                 "note": "opt-in (DENET_HEAD_BF16X3=1), NOT the headline: a b ~= a_hi b_hi + a_hi b_lo + a_lo b_hi on "
                                         "v_mfma_f32_32x32x16_bf16, fp32 accumulation; 2-4e-6 max-norm error per GEMM against fp64 "
                                         "(the exact fp32 kernels: 1-2.5e-6); everything else as in `value`"}
        except Exception as exc:          # an optional leg must never cost the headline line
            ops.HEAD_BF16X3 = False
            out["split_bf16"] = {"error": repr(exc)[:300]}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import model as OM
        cores = host_cores()
        m1 = zoo.denet34(1, "skip", 512, class_num=80, seed=1)
        om = OM.OracleModel(m1.export_json(), 1)
        x1, metas1 = zoo.synthetic_batch(1, 512, 80, seed=1)
        random.seed(1)
        from threadpoolctl import threadpool_limits
        with threadpool_limits(limits=cores):      # BLAS threads = the cores this process may really use
            t0 = time.perf_counter()
            nsteps = 0
            while nsteps < 8 and (nsteps == 0 or time.perf_counter() - t0 < 12.0):
                om.train_step(x1, metas1, nsteps, lr, mom[0], decay, "nesterov")
                nsteps += 1
            cdt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(nsteps / cdt, 4), "unit": "images/sec", "cores": cores, "kind": "port",
                               "sample": "%d full train steps at batch 1 of the same model (numpy im2col+BLAS conv, "
                                         "C++ RoI proposal), %.1f s" % (nsteps, cdt)}

    if rank == 0:
        print(json.dumps(out), flush=True)
    if dp is not None:
        dp.dist.destroy_process_group()


if __name__ == "__main__":
    main()
