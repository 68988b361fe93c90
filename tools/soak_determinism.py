"""N training steps of DeNet-34 skip from the same seed, twice: parameters, momentum and running statistics must agree bit for bit
(no float atomics; which kernel a pass runs is never a timing race: committed decisions at the benchmark geometries,
ops.static_policy everywhere else).

    python tools/soak_determinism.py [steps] [--img 512] [--batch 32]      two runs in ONE process; with PRESS=1 (default) a side
                                                                           stream keeps the memory system saturated during the second
                                                                           (the condition under which the store hazard of
                                                                           csrc/wino4f.hip showed, EXPERIMENTS.md)
    python tools/soak_determinism.py [steps] --img 768 --batch 8 --two-processes
                                                                           two separate PROCESSES at a geometry the tuned file does
                                                                           not hold (round-5 verdict, item 1: before round 6 each
                                                                           process measured its own kernels there); the digests of
                                                                           their states are compared"""
import argparse
import hashlib
import json
import os
import random
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run_steps(n, img, batch, press_side=None):
    import torch
    from denet_amd.model import zoo
    x, metas = zoo.synthetic_batch(batch, img, 80, seed=1)
    xd = torch.from_numpy(x).cuda()
    model = zoo.warm_corner_head(zoo.denet34(batch, "skip", img, class_num=80, seed=1))
    model.build_train_func("nesterov")
    random.seed(1)
    t0 = time.perf_counter()
    for it in range(n):
        if press_side is not None:
            side, a, b = press_side
            with torch.cuda.stream(side):
                for _ in range(12):
                    b.copy_(a, non_blocking=True)
        c, _ = model.train_step(xd, metas, 0, it, 0.02, [0.9], 1e-4)
    torch.cuda.synchronize()
    rate = n * batch / (time.perf_counter() - t0)
    return model, c, rate


def digest(model):
    h = hashlib.sha256()
    for t in (model.P, model.M, model.S):
        h.update(t.detach().cpu().numpy().tobytes())
    return h.hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("steps", nargs="?", type=int, default=60)
    ap.add_argument("--img", type=int, default=512)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--two-processes", action="store_true")
    ap.add_argument("--child", default=None, help="(internal) run once and write {digest, cost, kernels} to this file")
    args = ap.parse_args()
    if args.child:
        from denet_amd import ops
        from denet_amd.model import audit
        model, c, rate = run_steps(args.steps, args.img, args.batch)
        import torch
        from denet_amd.model import zoo
        x, metas = zoo.synthetic_batch(args.batch, args.img, 80, seed=1)
        with audit.KernelAudit(model) as ka:          # WHICH kernels ran, layer by layer (one more step, learning rate 0)
            model.train_step(torch.from_numpy(x).cuda(), metas, 0, args.steps, 0.0, [0.9], 0.0)
            torch.cuda.synchronize()
        json.dump({"digest": digest(model), "cost": c, "rate": rate, "measure": ops.MEASURE,
                   "kernels": ka.summary()}, open(args.child, "w"))
        return 0
    if args.two_processes:
        outs = []
        for r in range(2):
            out = os.path.join(os.environ.get("TMPDIR", "/tmp"), "soak_%d_%d.json" % (os.getpid(), r))
            env = dict(os.environ)
            env.pop("DENET_TUNE", None)
            subprocess.run([sys.executable, os.path.abspath(__file__), str(args.steps), "--img", str(args.img), "--batch",
                            str(args.batch), "--child", out], check=True, env=env)
            outs.append(json.load(open(out)))
            os.remove(out)
            print("process %d: %d steps %dx%d B=%d, %.1f img/s, final cost %.6f, digest %s" % (
                r, args.steps, args.img, args.img, args.batch, outs[-1]["rate"], outs[-1]["cost"], outs[-1]["digest"][:16]), flush=True)
        same = outs[0]["digest"] == outs[1]["digest"] and outs[0]["kernels"] == outs[1]["kernels"]
        print("two processes, %dx%d B=%d, %d steps: same kernels %s, bit-identical state %s" % (
            args.img, args.img, args.batch, args.steps, outs[0]["kernels"] == outs[1]["kernels"], outs[0]["digest"] == outs[1]["digest"]))
        return 0 if same else 1
    import torch
    press = os.environ.get("PRESS", "1") == "1"
    side = (torch.cuda.Stream(), torch.empty(1 << 27, device="cuda"), torch.empty(1 << 27, device="cuda"))
    states = []
    for run in range(2):
        model, c, rate = run_steps(args.steps, args.img, args.batch, side if run == 1 and press else None)
        print("run %d: %d steps, %.1f img/s, final cost %.6f" % (run, args.steps, rate, c), flush=True)
        states.append((model.P.clone(), model.M.clone(), model.S.clone(), c))
        del model
    same = all(torch.equal(a, b) for a, b in zip(states[0][:3], states[1][:3]))
    print("bit-identical state after %d steps (second run %s): %s" % (args.steps, "under memory pressure" if press else "alone", same))
    return 0 if same else 1


if __name__ == "__main__":
    sys.exit(main())
