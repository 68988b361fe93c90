#!/usr/bin/env python
"""bench.py — images/sec of one DeNet-34 skip training step at 512x512 (BASELINE.json metric) on N MI355X.

  python bench.py --gpus N --steps K --warmup W        (N > 1 without a launcher: starts its own N ranks, see self_launch)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
         bench.py --gpus N --steps K --warmup W

A "step" = one full training step of the hot path on one synthetic MSCOCO-shaped batch of 32 images per GPU:
host targets -> forward (conv/BN stack, corner map, GPU RoI proposal, RoI editing, sparse gather, RoI head) ->
costs -> backward -> (N>1: RCCL gradient all-reduce overlapped with backward) -> fused nesterov update.
Inputs are resident in HBM before the timed region. Rank 0 prints ONE JSON line. N > 1: key `data_parallel` (backend as the
process group reports it, per-rank ms/step, all-reduce bytes and buckets per step, event-timed exposed collective time).

Extra legs (rank 0, N=1 only; outside the timed region):
  roofline      the dominant kernel (the matrix-kernel SYMBOL with the largest total time, named as rocprofv3 names it) is timed
                live with HIP events on the launch stream over `steps` further instrumented steps;
                achieved = algorithmic FLOPs per launch / average launch duration; peak = fp32 MFMA 157.3 TFLOP/s.
  warm_regime / stress_regime
                the same step with the corner head held in a firing regime (40 / 1400 cells per corner type and image above
                the threshold; the second takes the max_corners truncation branch): RoI proposal on the GPU, host hand-off
                phases under the reference's names.
  split_bf16    OPT-IN variant, never the headline: the head GEMMs as 3-term bf16 splits (own key, own number).
  data_parallel_selftest
                the RCCL code path on this one GPU: a world-size-1 "nccl" process group, the collectives forced on
                (DataParallel.force_collectives): communicator creation, bucket -> stream ordering, collectives per step,
                bytes, event-timed exposed collective time, img/s with the exchange inside the step. NOT a scaling number.
  config2 / config5
                BASELINE.json's secondary configurations as legs of this line (a few seconds each): ResNet-34 224x224 batch
                64 (examples/resnet34-imagenet.sh:7) and DeNet-101 wide 512x512 batch 16 with joint fitness + bounded-IoU
                loss, `DND.JB` (papers/dss/denet101.sh:19): images/sec of full training steps and the executed-FLOP matrix
                utilisation.
  cpu_baseline  the numpy/C++ oracle (oracle/, a CPU restatement of the reference path, kind "port") runs full training
                steps of the same model at batch 4 on the host cores (CPU model and core count stated).
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
        kernels = json.load(f)["kernels"]
    k = kernels.get(kernel)
    if k is None:
        # a profile id that covers several instantiations (wino4f_kernel -> wino4f_kernel_32<1>, _32x2<2>, ...): launch-weighted
        fam = [v for n, v in kernels.items() if n.startswith(kernel + "_") or n.startswith(kernel + "<")]
        fam = [v for v in fam if v.get("write_bytes_per_launch") is not None and v.get("launches")]
        if not fam:
            return None
        n = sum(v["launches"] for v in fam)
        return round(sum(v["launches"] * (v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"]) for v in fam) / n)
    if k.get("write_bytes_per_launch") is None:
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


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or "unknown"


def config_leg(name, model, x, metas, steps, warm):
    """images/sec of full training steps of a secondary configuration + the FLOPs its matrix kernels execute (KernelProfile:
    every implicit-GEMM / fused Winograd launch of one extra step, Winograd layers with their reduced products)"""
    import torch
    from denet_amd import ops
    model.build_train_func("nesterov")
    xd = torch.from_numpy(x).cuda()
    random.seed(1)
    it = 0
    for _ in range(warm):
        model.train_step(xd, metas, 0, it, 0.05, [0.9], 1e-4)
        it += 1
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        cost, _ = model.train_step(xd, metas, 0, it, 0.05, [0.9], 1e-4)
        it += 1
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    prof = ops.KernelProfile()
    ops.PROFILE = prof
    model.train_step(xd, metas, 0, it, 0.05, [0.9], 1e-4)
    ops.PROFILE = None
    executed = sum(a["flops"] for a in prof.summary().values())
    B = x.shape[0]
    return {"workload": name, "batch": B, "value": round(B / dt, 1), "unit": "images/sec", "ms_per_step": round(1e3 * dt, 2),
            "steps": steps, "warmup": warm, "dtype": "f32", "step_gflop_executed_per_image": round(executed / B / 1e9, 2),
            "step_mfma_util_executed": round(executed / dt / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
            # (executed FLOPs / time / peak: it FALLS when a layer moves to an algorithm with fewer products - F(4x4) on the 14x14 and
            # 7x7 maps of config 2 executes 7.5 instead of 11.6 GFLOP per image and the step is 20 % faster; images/sec is the figure)
            "final_cost": round(float(cost), 5)}


def inference_leg(steps):
    """SURVEY section 8 f-1, the reference's only PUBLISHED rate (README.md:118-128: DeNet-34 skip 82 Hz on a Titan X, other hardware,
    context only): DeNetDetectLayer.get_detections (denet_detect.py:316-424, denet_detect.cc:35-173) on DeNet-34 skip 512x512 -
    test-mode forward with folded batch norms, GPU RoI proposal, head, decode, per-class NMS, host lists - at B = 1 (Hz) and B = 32
    (img/s), hard NMS and Gaussian soft-NMS. Before timing, the B = 1 detections are CHECKED against the oracle (the checker, not
    the thing measured): RoI lists on the product's corner map, test-mode corner map of oracle/model.py, threshold + NMS on the
    product's decoded arrays."""
    import ctypes
    import numpy
    import torch
    from denet_amd import ops
    from denet_amd.model import audit, zoo
    res = {"reference_published": {"value": 82, "unit": "Hz", "hardware": "Titan X (Pascal), cuDNN", "source": "README.md:122",
                                   "note": "other hardware: context, not a baseline for vs_baseline"},
           "model": "DeNet-34 skip 512x512, 80 classes, 576 RoIs per image, warm corner head (random corner filters, bias 4), "
                    "random detection filters; prThreshold 0.05, nmsThreshold 0.5"}
    for B in (1, 32):
        _inference_batch(B, steps, res)
        torch.cuda.empty_cache()
    return res


def _inference_batch(B, steps, res):
    """one batch size of inference_leg (its model lives and dies in this frame)"""
    import numpy
    import torch
    from denet_amd import ops
    from denet_amd.model import audit, zoo
    model = zoo.denet34(B, "skip", 512, class_num=80, seed=1)
    zoo.warm_corner_head(model, 4.0, 0.3)
    rng = numpy.random.RandomState(3)
    by_type = lambda t, layers=model.layers: [l for l in layers if l.type_name == t][0]
    dnd, dns, dnc = by_type("denet-detect"), by_type("denet-sparse"), by_type("denet-corner")
    dnd.layers[0].omega.set_value(rng.normal(0, 0.02, dnd.layers[0].omega.value.shape))
    x, metas = zoo.synthetic_batch(B, 512, 80, seed=1)
    xd = torch.from_numpy(x).cuda()
    for soft in (0, 1):
        params = {"prThreshold": 0.05, "nmsThreshold": 0.5, "useSoftNMS": soft}
        key = "b%d_%s" % (B, "soft_nms" if soft else "nms")
        for _ in range(3):
            r = dnd.get_detections(model, xd, metas, params)
        ent = {}
        if B == 1:
            ent["oracle_check"] = _inference_check(model, dnd, dns, dnc, x, r, params)
            # the checker kept the host busy and the device idle for a second or two (power state, BLAS threads winding down):
            # a few untimed calls before the clock starts (without them the first leg read 6.5 ms per call against 2.4 alone)
            for _ in range(10):
                dnd.get_detections(model, xd, metas, params)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            r = dnd.get_detections(model, xd, metas, params)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        ent.update({"value": round(B / dt, 2), "unit": "Hz" if B == 1 else "images/sec", "ms_per_batch": round(1e3 * dt, 3),
                    "batch": B, "calls_timed": steps, "detections_last_batch": sum(len(i["detections"]) for i in r),
                    "rois_last_batch": int(dnd.last_outputs[3].sum())})
        if not soft:
            prof = ops.KernelProfile()
            ops.PROFILE = prof
            try:
                for _ in range(3):
                    dnd.get_detections(model, xd, metas, params)
            finally:
                ops.PROFILE = None
            agg = prof.summary()
            name, a = max(agg.items(), key=lambda kv: kv[1]["ms"])
            tf = a["flops"] / (a["ms"] * 1e-3) / 1e12
            ent["dominant_kernel"] = {"kernel": name, "launches_per_call": a["launches"] // 3, "ms_per_call": round(a["ms"] / 3, 4),
                                      "tflops": round(tf, 2), "frac": round(tf / PEAK_FP32_MFMA_TFLOPS, 4)}
            ent["matrix_kernel_ms_per_call"] = round(sum(v["ms"] for v in agg.values()) / 3, 3)
            with audit.KernelAudit(model) as ka:
                dnd.get_detections(model, xd, metas, params)
                torch.cuda.synchronize()
            ent["kernels_used"] = {g: e["fwd"] for g, e in ka.summary().items()}
        res[key] = ent


def _inference_check(model, dnd, dns, dnc, x, results, params):
    """the oracle as CHECKER of one B = 1 get_detections call (outside every timed region); raises on a mismatch"""
    import ctypes
    import numpy
    from oracle import layers as OL
    from oracle import model as OM
    corner = dnc.corner_pr.cpu().numpy()
    thr = params.get("cornerThreshold", dns.corner_threshold)
    lists = OM.oracle_build_samples(corner, thr, dns.sample_num, 1024, 0)
    got = dns.sample_bbox_list
    for g, r in zip(got, lists):
        if [p for p, _ in g] != [p for p, _ in r]:
            raise AssertionError("inference check: RoI scores differ from the oracle's proposal on the same corner map")
        if len({p for p, _ in r}) == len(r) and g != r:
            raise AssertionError("inference check: RoI lists differ (tie-free)")
    om = OM.OracleModel(model.export_json(), 1)
    om.forward(x, None, train=False, sample_override=got)
    cerr = float(numpy.abs(corner - om.corner_pr).max() / (numpy.abs(om.corner_pr).max() + 1e-12))
    det_pr, fitness, bbox, counts = dnd.last_outputs
    sn, C = dns.sample_num, dnd.class_num
    t0 = dnd._thresholds()[0]
    o_det, o_fit, o_box = OL.detect_outputs(om.detect_out.v, om.sample_bbox, C, bool(dnd.use_jointfit), t0)
    n = int(counts[0])
    d = det_pr.cpu().numpy().reshape(1, sn, sn, C + 1).transpose(0, 3, 1, 2).reshape(C + 1, -1)[:, :n]
    # (the tolerance of tests/test_inference_gpu.py: |product - oracle| <= 1e-3 + 1e-3 |oracle| on the class log-probabilities)
    o_d = o_det.reshape(C + 1, -1)[:, :n]
    derr = float((numpy.abs(d - o_d) / (1e-3 + 1e-3 * numpy.abs(o_d))).max()) if n else 0.0
    # threshold + NMS: exact on the product's decoded arrays (oracle/build_samples.cc restates denet_detect.cc:99-173)
    S, C1 = sn * sn, C + 1
    det = numpy.ascontiguousarray(det_pr.cpu().numpy().reshape(1, sn, sn, C1).transpose(0, 3, 1, 2), dtype=numpy.float32)
    fit = numpy.ascontiguousarray(fitness.cpu().numpy().reshape(1, sn, sn, C1).transpose(0, 3, 1, 2), dtype=numpy.float32)
    bx = numpy.ascontiguousarray(bbox.cpu().numpy().reshape(1, sn, sn, 4), dtype=numpy.float32)
    num = numpy.ascontiguousarray(counts, dtype=numpy.int32)
    out = numpy.zeros((1, S * C, 6), numpy.float32)
    cnt = numpy.zeros(1, numpy.int32)
    f = OM.oracle_lib().oracle_build_detections_nms
    f.argtypes = [ctypes.c_float, ctypes.c_float, ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_int] * 4 + [ctypes.c_void_p] * 2
    f(params["prThreshold"], params["nmsThreshold"], int(params.get("useSoftNMS", 0)), det.ctypes.data, fit.ctypes.data,
      bx.ctypes.data, num.ctypes.data, 1, C1, sn, S * C, out.ctypes.data, cnt.ctypes.data)
    ref = out[0, :cnt[0]]
    dets = results[0]["detections"]
    if len(dets) != len(ref):
        raise AssertionError("inference check: %d detections, the oracle keeps %d" % (len(dets), len(ref)))
    for (pr, cls, box), rr in zip(dets, ref):
        if cls != int(rr[1]) or not numpy.array_equal(numpy.array(box, numpy.float32), rr[2:]) or abs(pr - rr[0]) > 2e-6 * rr[0]:
            raise AssertionError("inference check: a detection differs from the oracle's NMS on the same decoded arrays")
    if cerr > 1e-3 or derr > 1.0:
        raise AssertionError("inference check: corner map %.2e (bound 1e-3) / class log-probabilities %.2f of their tolerance off the "
                             "oracle's test-mode forward" % (cerr, derr))
    return {"rois": n, "detections": len(dets), "roi_lists_equal_oracle_proposal": True, "nms_equal_oracle": True,
            "corner_map_max_rel_err_vs_oracle": float("%.2e" % cerr), "class_logprob_max_err_in_units_of_tolerance_1e-3_abs_plus_1e-3_rel": float("%.3f" % derr)}


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-run this command line as N ranks (one per GPU) under
    torch.distributed.run on a free local port, pass the ranks' output through and exit with the job's status. The reference's
    driver spawns its own workers too (denet/model/train_multi.py:96-145, denet/multi/worker.py:138-243)."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: the only mode the host driver supports
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


def regime_leg(args, xd, metas, target_cells, lr, mom, decay):
    """The headline step with the corner head of DNC held in a firing regime: `target_cells` cells per corner type and image
    above the corner threshold. The corner rows of the DNC convolution (random weights, zoo.warm_corner_head) are FROZEN: after every step
    their weights are restored, their momentum zeroed and their bias re-set by a controller (hold() below; a few tiny device
    ops, no host synchronisation) - the corner cost would otherwise train the head cold within the warm-up steps (lr 0.1 x
    cost factor) and the leg would time a second cold regime. Everything else trains as in `value`: same kernels, same
    host work."""
    import math
    import torch
    from denet_amd import ops
    from denet_amd.model import zoo
    m = zoo.warm_corner_head(zoo.denet34(BATCH_PER_GPU, "skip", 512, class_num=80, seed=1))
    m.build_train_func("nesterov")
    dnc = [l for l in m.layers if l.type_name == "denet-corner"][0]
    dns = [l for l in m.layers if l.type_name == "denet-sparse"][0]
    conv, cn = dnc.layers[-1], dnc.corner_num
    log_thr = math.log(dns.corner_threshold)
    # The synthetic batch has no image content a corner head could learn, so its cost can only push every logit towards "no
    # corner" - through the frozen rows into the backbone, 10 logit units within 20 steps (measured: the bias controller below
    # ran away to -39 and the regime collapsed). The corner cost factor (DNC costFactor, denet_corner.py:24) is therefore 0 in
    # these legs: its loss and gradient kernels run as always and write a zero gradient; the backbone trains on the detection
    # cost alone.
    dnc.cost_factor = 0.0
    random.seed(1)
    it = [0]

    def step(rate):
        c, _ = m.train_step(xd, metas, 0, it[0], rate, mom, decay)
        it[0] += 1
        return c

    def cells():
        # corner_pr [B, 2, types, H, W] log-probabilities, plane 1 = "corner" (denet_sparse.cc:503-511)
        return (dnc.corner_pr[:, 1] > log_thr).sum(dim=(2, 3)).float()

    rows = conv.omega.dev_shape[0]             # filters are [Kp][R][S][Cp] inside the flat parameter buffer: row = output channel
    w_dev, w_mom = conv.omega.dev.view(rows, -1), conv.omega.mom.view(rows, -1)
    keep_w = w_dev[:cn].clone()
    bias = conv.beta.dev[:cn].clone()
    kth = max(1, int(round(target_cells * BATCH_PER_GPU)))
    x_thr = 0.5 * math.log(math.expm1(-log_thr))

    def hold():
        """frozen corner rows + a one-step bias controller, all on the device (no host synchronisation): the backbone keeps
        training (detection cost), its features drift, so the bias of each corner type follows - it is set to the value that
        puts the target number of cells of THIS step's corner map above the threshold (plane 1 = log sigmoid(-2x): the k-th largest log-probability v of a type corresponds to the
        logit x_v = log(expm1(-v)) / 2, and the bias moves by x_thr - x_v)"""
        w_dev[:cn].copy_(keep_w)
        w_mom[:cn].zero_()
        conv.beta.mom[:cn].zero_()
        plane = dnc.corner_pr[:, 1].transpose(0, 1).reshape(cn, -1)                 # [types, B*H*W]
        v = torch.sort(plane, dim=1, descending=True).values[:, kth - 1].clamp(max=-1e-6)
        x_v = (0.5 * torch.log(torch.expm1(-v))).clamp(min=-12.0, max=12.0)
        bias.add_(x_thr - x_v)
        conv.beta.dev[:cn].copy_(bias)

    for _ in range(4):             # calibration on this batch: no learning, the controller alone
        step(0.0)
        hold()
    for _ in range(max(args.warmup, 2)):
        step(lr)
        hold()
    torch.cuda.synchronize()
    rois, ncell, nkept, ncand, phases = [], [], [], [], {}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        cost = step(lr)
        hold()
        raw = getattr(dns, "_raw_samples", None)
        rois.append(float(raw[1].mean()) if raw is not None else 0.0)
        for k, v in dns.phase_ms.items():
            phases[k] = phases.get(k, 0.0) + v
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # diagnostics of the regime, outside the timed region: three further steps
    for _ in range(3):
        step(lr)
        hold()
        c = cells()
        kept, cand = ops.build_samples_stats(dnc.corner_pr, dns.proposal_count, dns.corner_max)
        ncell.append((float(c.mean()), float(c.min()), float(c.max())))
        nkept.append(float(kept.float().mean()))
        ncand.append(float(cand.float().mean()))
    # the RoI proposal alone (corner_select + pair_* kernels) on the last corner map, event-timed on its stream
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ops.build_samples(dnc.corner_pr, dns.corner_threshold, dns.proposal_count, dns.corner_max, dns.local_max)
    e0.record()
    for _ in range(10):
        ops.build_samples(dnc.corner_pr, dns.corner_threshold, dns.proposal_count, dns.corner_max, dns.local_max)
    e1.record()
    torch.cuda.synchronize()
    mean_rois = sum(rois) / len(rois)
    if mean_rois < 50:
        raise RuntimeError("corner head left the regime: %.1f detector RoIs per image (%.0f cells per type)"
                           % (mean_rois, ncell[-1][0]))
    if not math.isfinite(cost):
        raise RuntimeError("non-finite cost %r" % cost)
    return {"value": round(BATCH_PER_GPU * args.steps / dt, 2), "unit": "images/sec",
            "ms_per_step": round(1e3 * dt / args.steps, 3),
            "target_cells_above_threshold_per_type": target_cells,
            "cells_above_threshold_per_type": {"mean": round(sum(c[0] for c in ncell) / len(ncell), 1),
                                               "min": min(c[1] for c in ncell), "max": max(c[2] for c in ncell)},
            "corners_kept_per_type": round(sum(nkept) / len(nkept), 1), "max_corners": dns.corner_max,
            "truncation_branch_taken": bool(max(c[2] for c in ncell) > dns.corner_max),
            "pair_candidates_per_image": round(sum(ncand) / len(ncand), 1),
            "detector_rois_per_image": round(mean_rois, 1), "rois_per_image_cap": dns.proposal_count,
            "roi_proposal_kernels_ms": round(e0.elapsed_time(e1) / 10, 4),
            # host phases of the RoI hand-off per step, the reference's names (denet_sparse.py:127-161): model = queue the proposal
            # + wait for the device (forward pass up to the corner map included), build = host epilogue of build_samples
            "host_phase_ms_per_step": {k: round(v / args.steps, 3) for k, v in sorted(phases.items())},
            "corner_bias_per_type": [round(float(b), 3) for b in bias.tolist()], "final_cost": round(float(cost), 5),
            "note": "same step and kernels as `value`; corner rows of the DNC convolution frozen, their bias follows a one-step "
                    "on-device controller that keeps the target number of cells above the threshold, corner cost factor 0 "
                    "(regime_leg in bench.py says why)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-warm", action="store_true", help="skip the warm / stress corner-regime legs (rank 0, N=1)")
    ap.add_argument("--no-dp-selftest", action="store_true", help="skip the single-GPU RCCL self-test leg (rank 0, N=1)")
    ap.add_argument("--no-configs", action="store_true", help="skip the config2 / config5 legs (rank 0, N=1)")
    ap.add_argument("--no-instep", action="store_true", help="skip the in-step kernel timing leg (roofline.dominant_by_time_in_step)")
    ap.add_argument("--no-h2d", action="store_true", help="skip the input_h2d leg (host batch uploaded every step; rank 0, N=1)")
    ap.add_argument("--no-audit", action="store_true", help="skip the per-layer kernel audit (config.kernels_used; rank 0)")
    ap.add_argument("--no-inference", action="store_true", help="skip the inference leg (get_detections Hz / img/s; rank 0, N=1)")
    ap.add_argument("--no-split-bf16", action="store_true",
                    help="skip the leg of the OPT-IN variant (head GEMMs as 3-term bf16 splits; own key, never the headline)")
    ap.add_argument("--regime", default="cold", choices=["cold", "warm"],
                    help="cold = weights as initialised (corner bias +5: no detector RoIs, SURVEY §8d); warm = corner "
                         "head re-biased so that ~1%% of the cells fire")
    ap.add_argument("--share-gpu", action="store_true", default=os.environ.get("DENET_BENCH_SHARE_GPU") == "1",
                    help="LAUNCH-PATH DEBUG MODE, never a scaling measurement: all N ranks run on cuda:0 and torch.distributed "
                         "uses gloo (one GPU cannot host two RCCL ranks); exercises launcher, rendezvous, bucketed "
                         "all-reduce and the rank-0 JSON line on a one-GPU box")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus)          # does not return

    import numpy
    import torch
    from denet_amd import ops
    from denet_amd.model import zoo

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but the launcher started %d ranks" % (args.gpus, world))
    share = args.share_gpu
    if not share and world > torch.cuda.device_count():
        raise SystemExit("--gpus %d but %d HIP devices are visible (one rank per GPU; --share-gpu is the launch-path "
                         "debug mode)" % (world, torch.cuda.device_count()))
    torch.cuda.set_device(0 if share else local_rank)
    ops.HEAD_BF16X3 = False      # the headline is fp32 MFMA whatever the environment says; the opt-in variant has its own leg below
    dp = None
    if world > 1 or os.environ.get("DENET_FORCE_DP") == "1":      # DENET_FORCE_DP: exercise the RCCL path on 1 GPU
        from denet_amd.multi import DataParallel
        dp = DataParallel(backend="gloo" if share else "nccl")      # "nccl" is RCCL on ROCm
        dp.force_collectives = os.environ.get("DENET_FORCE_DP") == "1"
        world = dp.world_size          # what the process group itself reports
        # the first multi-GPU run must not be able to fall back silently: the group has as many ranks as --gpus asked for, and
        # between GPUs it is RCCL ("nccl" on ROCm) - gloo only in the one-GPU launch-path debug mode
        import torch.distributed as tdist
        if tdist.get_world_size() != args.gpus:
            raise SystemExit("--gpus %d but the process group has %d ranks" % (args.gpus, tdist.get_world_size()))
        backend = str(tdist.get_backend()).lower()
        if args.gpus > 1 and not share and backend != "nccl":
            raise SystemExit("--gpus %d without --share-gpu must run on RCCL (backend 'nccl'), the process group reports %r" % (
                args.gpus, backend))
        if dp.backend != backend:
            raise SystemExit("DataParallel says backend %r, torch.distributed %r" % (dp.backend, backend))

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
        torch.cuda.synchronize()
        if dp is not None:
            dp.barrier()
            torch.cuda.synchronize()

    from denet_amd.model import audit
    # 3x3 passes the committed tuned file does not decide: with ops.MEASURE (DENET_TUNE=1) they are timed in the warm-up; in the product
    # default they run ops.static_policy's algorithm and nothing is measured
    uncovered = len(audit.decisions_cover(model))
    undecided = uncovered if ops.MEASURE and ops.POLICY is None else 0
    it = 0
    # one-off setup (buffers, filter transforms; with DENET_TUNE=1 also measuring) happens in the first two steps: they are taken out
    # of the timed region even when fewer warm-up steps were asked for
    for _ in range(max(args.warmup, 2)):
        model.train_step(xd, metas, 0, it, lr, mom, decay)
        it += 1
    sync()
    if dp is not None:
        dp.start_timing()
    dns_layer = [l for l in model.layers if l.type_name == "denet-sparse"][0]
    dns_layer.proposed_total = dns_layer.proposed_steps = 0
    dns_layer.handoff_modes = {k: 0 for k in dns_layer.handoff_modes}
    import gc
    gc.collect()
    gc.disable()      # no collector pause of the host thread inside the timed region (a full collection is milliseconds): the policy
    try:              # of the shipped epoch loops too (ModelCNN.train_epoch / train_epoch_device: _collector_paused)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            cost, _ = model.train_step(xd, metas, 0, it, lr, mom, decay)
            it += 1
        sync()
        dt_rank = dt = time.perf_counter() - t0
    finally:
        gc.enable()
    # what the corner detector did during the timed steps: as initialised it is silent (bias +5), but the corner cost (factor 100,
    # lr 0.1) has it firing within the warm-up steps and cooling down over the next ~30 - the timed steps see a detector that
    # proposes, and the RoI lists are trimmed by random.sample
    det_rois = dns_layer.proposed_total / max(1, dns_layer.proposed_steps) / BATCH_PER_GPU
    dp_info = None
    if dp is not None:
        dt = dp.max_over_ranks(dt_rank)
        per_rank = dp.gather_floats(1e3 * dt_rank / args.steps)
        exposed = dp.gather_floats(dp.exposed_ms_per_step())
        dp_info = {"backend": dp.backend + (" (RCCL)" if dp.backend == "nccl" else ""),
                   "world_size": dp.world_size, "ms_per_step_per_rank": [round(v, 3) for v in per_rank],
                   "allreduce_bytes_per_step": dp.bytes_per_step(), "collectives_per_step": dp.collectives_per_step(),
                   "bucket_bytes": [4 * (hi - lo) for lo, hi, _ in (dp._buckets or [])],
                   # time the compute stream stood waiting for collectives after its last backward kernel (HIP events
                   # around the wait in DataParallel.finish_step): the part of the exchange the backward pass did not hide
                   "exposed_collective_ms_per_step_per_rank": [round(v, 3) for v in exposed],
                   # rank 0: per bucket, the window between its collective's issue and the end of the backward sweep, and what the
                   # compute stream still waited for it afterwards
                   "bucket_overlap": dp.bucket_overlap_table()}
        if share:
            dp_info["launch_path_test_only"] = ("all %d ranks share cuda:0 and gloo carries the tensors through the host: this "
                                                "line proves the launch / rendezvous / exchange path, it is NOT a scaling "
                                                "measurement" % dp.world_size)
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
                               "MSCOCO, full train step (targets, fwd, bwd, nesterov), corner head %s" % (
                                   "as initialised (silent at step 0; see detector_rois_per_image_in_the_timed_steps)"
                                   if args.regime == "cold" else "warm"),
                   "global_batch": BATCH_PER_GPU * world, "batch_per_gpu": BATCH_PER_GPU, "classes": 80,
                   "rois_per_image": 576, "detector_rois_per_image_in_the_timed_steps": round(det_rois, 1),
                   # which form of the RoI hand-off each timed step took (denet_amd/layer/roi_handoff.py)
                   "handoff_modes": dict(dns_layer.handoff_modes),
                   "parallelism": "dp%d" % world, "solver": "nesterov",
                   "conv_algorithms": "fp32 throughout; per layer and pass the fastest of the direct implicit GEMM, "
                                      "Winograd F(2x2,3x3)/F(4x4,3x3) and (64 input channels) F(2x2,3x3) fused into one "
                                      "kernel, as measured once by tools/tune.py and stored in "
                                      "denet_amd/tuned/gfx950.json (every process runs the same kernels; a geometry not "
                                      "in the file runs ops.static_policy's algorithm and is NEVER measured unless DENET_TUNE=1; "
                                      "DENET_WINOGRAD=0: direct kernels only)",
                   "input": "fp32 NCHW batch resident in HBM before the timed region",
                   "gc_disabled_in_timed_region": True,       # as in ModelCNN.train_epoch (cyclic collector paused per epoch)
                   "final_cost": round(float(cost), 5)},
        # rate in FLOPs of the reference's direct algorithm (164.3 GFLOP per image and step); layers that run Winograd
        # execute fewer, so this is an effective rate - the MFMA utilisation of the kernels is in `roofline`
        "step_tflops_algorithmic": round(value * FLOP_PER_IMAGE_STEP / 1e12, 2),
        # NOT a utilisation: FLOPs of the reference's direct algorithm / time / peak; exceeds what the matrix cores execute by
        # the Winograd saving and can pass 1.0. The utilisation of the step is `step_mfma_util_executed` (roofline leg).
        "step_effective_frac_algorithmic": round(value * FLOP_PER_IMAGE_STEP / 1e12 / (PEAK_FP32_MFMA_TFLOPS * world), 4),
    }

    if dp_info is not None:
        out["data_parallel"] = dp_info

    # ---- self-audit: WHICH kernels the timed steps ran, layer by layer, and under which switches (one more step outside the
    # timed region; the C side notes the instantiation of every matrix-kernel launch, no events, no effect on streams) ----
    out["config"]["denet_switches"] = audit.active_switches()          # DENET_* environment switches: {} is the product default
    from denet_amd import switches
    out["config"]["product_default_switches"] = not switches.changes_kernels(out["config"]["denet_switches"])
    out["config"]["tuned_file"] = os.path.relpath(ops.TUNE_CACHE, os.path.dirname(os.path.abspath(__file__))) if ops._TUNE_LOADED else None
    if not args.no_audit:       # (every rank: the step holds collectives when the job is data parallel)
        out["config"]["passes_measured_in_the_warmup"] = undecided       # 0: nothing was timed to choose a kernel
        out["config"]["passes_not_in_the_tuned_file"] = uncovered        # 0: every implementation came from the committed file
        out["config"]["untuned_geometry_policy"] = "measure (DENET_TUNE=1)" if ops.MEASURE else "static_policy (nothing measured)"
        with audit.KernelAudit(model) as ka:
            model.train_step(xd, metas, 0, it, lr, mom, decay)
            it += 1
            torch.cuda.synchronize()
        out["config"]["kernels_used"] = {g: {k: v for k, v in e.items()} for g, e in ka.summary().items()}

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
        # FLOPs the matrix cores really execute in one step (every implicit-GEMM launch and the fused F(2x2) kernels, Winograd
        # layers counted with their reduced products) / the timed step / peak: the step-level MFMA utilisation
        executed = sum(v["flops"] for v in agg.values()) / nprof
        out["step_gflop_executed"] = round(executed / 1e9, 1)
        out["step_mfma_util_executed"] = round(executed / (dt / args.steps) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)
        out["roofline"] = {"bound": "mfma", "kernel": name, "achieved": round(achieved, 2),
                           "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                           "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": pmc_traffic(name),
                           "launches_per_step": a["launches"] // nprof, "avg_launch_ms": round(avg_ms, 4),
                           "flop_per_launch": round(a["flops"] / a["launches"]),
                           "all_igemm": {k: {"launches_per_step": v["launches"] // nprof,
                                             "ms_per_step": round(v["ms"] / nprof, 3),
                                             "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2)}
                                         for k, v in sorted(agg.items())}}
        # the same records grouped by source kernel (all instantiations of a template together): the fused F(4x4) kernel's
        # tile-block / epilogue variants add up to more step time than any single symbol
        fam = {}
        for k, v in agg.items():
            f = fam.setdefault(k.split("<")[0].replace("_32x2", "").replace("_32k", "").replace("_32", "").replace("_64", "")
                               if k.startswith("wino4f") else k.split("<")[0], {"launches": 0, "ms": 0.0, "flops": 0.0})
            for q in ("launches", "ms", "flops"):
                f[q] += v[q]
        out["roofline"]["by_family"] = {k: {"launches_per_step": v["launches"] // nprof, "ms_per_step": round(v["ms"] / nprof, 3),
                                            "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2),
                                            "frac": round(v["flops"] / (v["ms"] * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)}
                                        for k, v in sorted(fam.items())}

    if rank == 0 and world == 1 and not args.no_roofline and not args.no_instep:
        # the same event pairs with the step's two kernel chains LEFT ON: what every matrix kernel costs inside the step (beside the
        # other chain's kernels, the wait for CU slots included) next to its alone figure above
        try:
            prof2 = ops.KernelProfile(alone=False)
            ops.PROFILE = prof2
            for _ in range(nprof):
                model.train_step(xd, metas, 0, it, lr, mom, decay)
                it += 1
            ops.PROFILE = None
            agg2 = prof2.summary()
            name2, a2 = max(agg2.items(), key=lambda kv: kv[1]["ms"])
            tf2 = a2["flops"] / (a2["ms"] * 1e-3) / 1e12
            out["roofline"]["dominant_by_time_in_step"] = {
                "kernel": name2, "launches_per_step": a2["launches"] // nprof, "ms_per_step_in_step": round(a2["ms"] / nprof, 3),
                "ms_per_step_alone": round(agg[name2]["ms"] / nprof, 3) if name2 in agg else None,
                "tflops_in_step": round(tf2, 2), "frac_in_step": round(tf2 / PEAK_FP32_MFMA_TFLOPS, 4),
                "note": "event pair per launch on its own stream while BOTH backward chains run (the events themselves cost the "
                        "step a few per cent): a kernel's in-step duration includes what it waits for CU slots beside the other chain"}
            out["roofline"]["in_step_ms_per_step"] = {k: round(v["ms"] / nprof, 3) for k, v in sorted(agg2.items())}
        except Exception as exc:          # an extra leg must never cost the headline line
            ops.PROFILE = None
            out["roofline"]["dominant_by_time_in_step"] = {"error": repr(exc)[:300]}

    if rank == 0 and world == 1 and not args.no_h2d:
        # the reference's train_step takes HOST arrays (model_cnn.py:407): the same steps with the batch coming from pinned host
        # memory every step - uploaded into one of two device buffers on a copy stream while the previous step trains (the buffer is
        # free again when the step that read it has recorded model.input_consumed)
        try:
            xh = torch.from_numpy(x).pin_memory()
            bufs = [torch.empty_like(xd), torch.empty_like(xd)]
            copy_stream = ops.side_stream(2)
            ready = [None, None]

            def upload(i, after):
                with torch.cuda.stream(copy_stream):
                    if after is not None:
                        copy_stream.wait_event(after)
                    bufs[i].copy_(xh, non_blocking=True)
                    ready[i] = torch.cuda.Event()
                    ready[i].record(copy_stream)

            upload(0, None)
            nh = max(1, min(args.steps, 20))
            for s_ in range(2 + nh):
                if s_ == 2:
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                i_ = s_ & 1
                upload(1 - i_, getattr(model, "input_consumed", None) if s_ > 0 else None)      # the NEXT batch, beside this step
                torch.cuda.current_stream().wait_event(ready[i_])
                hcost, _ = model.train_step(bufs[i_], metas, 0, it, lr, mom, decay)
                it += 1
            torch.cuda.synchronize()
            hdt = time.perf_counter() - t0
            out["input_h2d"] = {"value": round(BATCH_PER_GPU * nh / hdt, 2), "unit": "images/sec", "ms_per_step": round(1e3 * hdt / nh, 3),
                                "steps": nh, "h2d_bytes_per_step": int(xh.numel() * 4), "final_cost": round(float(hcost), 5),
                                "note": "the batch is uploaded from pinned host memory EVERY step (double-buffered on a copy stream, "
                                        "hidden behind the previous step); `value` above keeps it resident in HBM as the contract asks"}
            del bufs, xh
        except Exception as exc:          # an extra leg must never cost the headline line
            out["input_h2d"] = {"error": repr(exc)[:300]}

    if rank == 0 and world == 1 and not args.no_warm and args.regime == "cold":
        # the headline regime has no detector RoIs (cold corner head). The same step with a firing corner head
        # exercises the RoI proposal for real: "warm" = a few hundred corners per type and image (pair search of
        # denet_sparse.cc:337-373, top-576 selection, random.sample trim), "stress" = more than max_corners = 1024 cells per
        # type above the threshold (the truncation branch, denet_sparse.cc:526-530, then up to 2 x 1024^2 pairs per image)
        try:
            del model
        except NameError:
            pass
        torch.cuda.empty_cache()
        for key, target in (("warm_regime", 40.0), ("stress_regime", 1400.0)):
            try:
                out[key] = regime_leg(args, xd, metas, target, lr, mom, decay)
            except Exception as exc:          # an extra leg must never cost the headline line
                out[key] = {"error": repr(exc)[:300]}
            torch.cuda.empty_cache()

    if rank == 0 and world == 1 and dp is None and not args.no_dp_selftest:
        # the data-parallel code path under the REAL backend on the one GPU this run has: RCCL communicator (world size 1),
        # every bucket's all-reduce launched from the backward sweep on the filter-gradient stream, the packed tail collective,
        # the wait in finish_step - driver-exercised every round although no round has had two GPUs
        try:
            try:
                del model
            except NameError:
                pass
            torch.cuda.empty_cache()
            import socket
            from denet_amd.multi import DataParallel
            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                port = sock.getsockname()[1]
            os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            sdp = DataParallel(backend="nccl")
            sdp.force_collectives = True
            sm = zoo.denet34(BATCH_PER_GPU, "skip", 512, class_num=80, seed=1)
            sm.build_train_func("nesterov")
            sm.dist = sdp
            sdp.broadcast_state(sm)
            random.seed(1)
            sit = 0
            for _ in range(max(args.warmup, 2)):
                sm.train_step(xd, metas, 0, sit, lr, mom, decay)
                sit += 1
            torch.cuda.synchronize()
            sdp.start_timing()
            nst = max(1, min(args.steps, 10))
            t0 = time.perf_counter()
            for _ in range(nst):
                scost, _ = sm.train_step(xd, metas, 0, sit, lr, mom, decay)
                sit += 1
            torch.cuda.synchronize()
            sdt = time.perf_counter() - t0
            out["data_parallel_selftest"] = {
                "backend": sdp.backend + (" (RCCL)" if sdp.backend == "nccl" else ""), "world_size": sdp.world_size,
                "collectives_per_step": sdp.collectives_per_step(), "allreduce_bytes_per_step": sdp.bytes_per_step(),
                "bucket_bytes": [4 * (hi - lo) for lo, hi, _ in (sdp._buckets or [])],
                "exposed_collective_ms_per_step": round(sdp.exposed_ms_per_step(), 3),
                # per bucket: bytes, how long before the end of the backward sweep the collective was issued (the window a real
                # exchange has to hide in: at ~100 GB/s per direction a ring moves 2 (N-1)/N x bytes), how long the compute stream
                # still waited for it after the sweep
                "bucket_overlap": sdp.bucket_overlap_table(),
                "ms_per_step": round(1e3 * sdt / nst, 3), "value": round(BATCH_PER_GPU * nst / sdt, 2), "unit": "images/sec",
                "final_cost": round(float(scost), 5),
                "note": "single GPU, world size 1, collectives forced on: exercises the RCCL launch path inside the step; the "
                        "exposed time is a self-copy's, not an xGMI exchange - never a scaling measurement"}
            # the recipe's actual multi-GPU mode (papers/dss/denet34.sh:43 `--batch-size-factor 2`, train_multi.py:104-119): F = 2
            # full LOCAL steps per rank, then parameters, momentum and BN statistics averaged over the ranks (three all-reduces:
            # DataParallel.average_state) - no gradient exchange inside the steps
            sm.dist = None
            nit = max(1, min(args.steps, 10) // 2)
            for k in range(1 + nit):
                if k == 1:
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                for _ in range(2):
                    scost, _ = sm.train_step(xd, metas, 0, sit, lr, mom, decay)
                    sit += 1
                sdp.average_state(sm)
            torch.cuda.synchronize()
            bdt = time.perf_counter() - t0
            out["data_parallel_selftest"]["local_sgd_bsf2"] = {
                "ms_per_iteration_of_2_steps": round(1e3 * bdt / nit, 3), "value": round(2 * BATCH_PER_GPU * nit / bdt, 2),
                "unit": "images/sec", "allreduce_bytes_per_iteration": int(sum(t.numel() * 4 for t in sdp._state(sm))),
                "final_cost": round(float(scost), 5)}
            del sm
            sdp.dist.destroy_process_group()
        except Exception as exc:          # an extra leg must never cost the headline line
            out["data_parallel_selftest"] = {"error": repr(exc)[:300]}
        torch.cuda.empty_cache()

    if rank == 0 and world == 1 and not args.no_configs:
        # BASELINE.json configs 2 and 5 (parity-test cases; their throughput rides along so that a driver-run record carries it)
        try:
            del model
        except NameError:
            pass
        torch.cuda.empty_cache()
        for key in ("config2", "config5"):
            try:
                if key == "config2":
                    mc = zoo.resnet34(64, 224, 1000)
                    xc, mtc = zoo.synthetic_batch(64, 224, 1000, seed=1, image_class=True)
                    out[key] = config_leg("ResNet-34 backbone + classifier 224x224 (examples/resnet34-imagenet.sh:7), full train step, "
                                          "algorithmic 22.0 GFLOP per image and step", mc, xc, mtc, 8, 3)
                else:
                    mc = zoo.denet101(16, "wide", 512, 80, head_desc=zoo.DENET101_WIDE_DESC.replace("DND[0.5,1,1]", "DND.JB[0.5,1,1]"))
                    xc, mtc = zoo.synthetic_batch(16, 512, 80, seed=1)
                    out[key] = config_leg("DeNet-101 wide 512x512, 2304 RoIs per image, joint fitness + bounded-IoU loss DND.JB "
                                          "(papers/dss/denet101.sh:19), full train step", mc, xc, mtc, 5, 2)
                del mc
            except Exception as exc:          # an extra leg must never cost the headline line
                out[key] = {"error": repr(exc)[:300]}
            torch.cuda.empty_cache()

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

    if rank == 0 and world == 1 and not args.no_inference:
        try:
            model = None                      # (another leg may have dropped it already: not `del`)
            torch.cuda.empty_cache()
            out["inference"] = inference_leg(max(5, min(args.steps, 20)))
        except Exception as exc:          # an extra leg must never cost the headline line
            ops.PROFILE = None
            out["inference"] = {"error": repr(exc)[:400]}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import model as OM
        cores = host_cores()
        CB = 4                                      # the CPU leg runs batch 4, scaled to images/sec
        m1 = zoo.denet34(CB, "skip", 512, class_num=80, seed=1)
        om = OM.OracleModel(m1.export_json(), CB)
        x1, metas1 = zoo.synthetic_batch(CB, 512, 80, seed=1)
        random.seed(1)
        from threadpoolctl import threadpool_limits
        with threadpool_limits(limits=cores):      # BLAS threads = the cores this process may really use
            t0 = time.perf_counter()
            nsteps = 0
            while nsteps < 4 and (nsteps == 0 or time.perf_counter() - t0 < 14.0):
                om.train_step(x1, metas1, nsteps, lr, mom[0], decay, "nesterov")
                nsteps += 1
            cdt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(CB * nsteps / cdt, 4), "unit": "images/sec", "cores": cores, "cpu": cpu_model(),
                               "kind": "port",
                               "sample": "%d full train steps at batch %d of the same model (numpy im2col+BLAS conv, "
                                         "C++ RoI proposal), %.1f s; a proxy for the reference's Theano CPU path, which "
                                         "cannot run here (Theano is not installed)" % (nsteps, CB, cdt)}

    if rank == 0:
        print(json.dumps(out), flush=True)
    if dp is not None:
        dp.dist.destroy_process_group()


if __name__ == "__main__":
    main()
