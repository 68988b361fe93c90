"""Measures the launch configuration of every convolution pass of the benchmark configurations on this GPU and writes them to a
file (default gpurun_out/gfx950.json; copy it to denet_amd/tuned/gfx950.json and commit). With the file in place no
process measures these geometries again: bench.py, the profiles and the PMC passes all run the same kernels.

    python tools/tune.py [--out PATH] [--reps N]

Each configuration is built and trained for two steps with an EMPTY cache (DENET_TUNE_CACHE=0 is forced), `--reps` times;
a geometry keeps the decision of the repetition... there is one decision per repetition and the majority wins (ties: the
first), which removes most of the timing noise of a single measurement."""
import argparse
import collections
import os
import random
import sys

os.environ["DENET_TUNE_CACHE"] = "0"
os.environ["DENET_TUNE"] = "1"                # measuring is an explicit act: the product default never times a candidate (ops.MEASURE)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from denet_amd import ops  # noqa: E402
from denet_amd.model import zoo  # noqa: E402


def run_configs(only=None):
    random.seed(1)
    for name, build, B, img in [c for c in [("denet34-skip", lambda: zoo.denet34(32, "skip", 512), 32, 512),
                                ("resnet34", lambda: zoo.resnet34(64, 224), 64, 224),
                                ("denet101-wide", lambda: zoo.denet101(16, "wide", 512), 16, 512),
                                ("cifar3", lambda: zoo.cifar3(32), 32, 32)] if only in (None, c[0])]:
        model = build()
        x, metas = zoo.synthetic_batch(B, img, class_num=model.class_num if model.class_num <= 80 else 80, image_class=True)
        model.build_train_func("nesterov")
        xd = torch.from_numpy(x).cuda()
        for it in range(2):
            model.train_step(xd, metas, 0, it, 0.01, [0.9], 1e-4)
        torch.cuda.synchronize()
        print("tuned", name, flush=True)
        del model
        torch.cuda.empty_cache()


def snapshot():
    import ctypes
    n = ops._L().denet_tune_export(None, 0)
    buf = (ctypes.c_int * (14 * max(n, 1)))()
    ops._L().denet_tune_export(buf, n)
    kern = {tuple(buf[i * 14:i * 14 + 11]): tuple(buf[i * 14 + 11:i * 14 + 14]) for i in range(n)}
    return kern, dict(ops._WINO)


def merge_records(old_kernels, old_wino, rec, wino):
    """replaces exactly the keys that were re-measured: a convolution record (mode <= 2) of the old file goes only if the new
    measurement holds the same 11-value geometry key, a decision only if the same (mode, geometry) was decided again; everything
    else - incl. the records of another configuration with the same batch size (denet34-skip and cifar3 both run batch 32) -
    stays. Batched-product records (mode > 2) are added where the file has none. Returns (kept, added, decisions kept, added)."""
    new_k = {tuple(r[:11]): list(r[11:]) for r in rec}
    kept = [r for r in old_kernels if not (r[0] <= 2 and tuple(r[:11]) in new_k)]
    have = {tuple(r[:11]) for r in kept}
    add = [list(k) + list(v) for k, v in new_k.items() if k[0] <= 2 or k not in have]
    new_w = {(int(m), tuple(int(v) for v in g)): int(t) for (m, g), t in wino.items()}
    wkept = [w for w in old_wino if (int(w[0]), tuple(int(v) for v in w[1])) not in new_w]
    wadd = [[m, list(g), t] for (m, g), t in new_w.items()]
    return kept, add, wkept, wadd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join("gpurun_out", "gfx950.json"))
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--only", default=None, help="measure ONE configuration (denet34-skip / resnet34 / denet101-wide / cifar3)")
    ap.add_argument("--merge-into", default=None,
                    help="with --only: an existing file; exactly the convolution records and decisions whose FULL geometry key was "
                         "re-measured are replaced (another configuration that shares the batch size keeps all of its records), "
                         "batched-product records are added where the file has none")
    args = ap.parse_args()
    votes_k, votes_w = collections.defaultdict(list), collections.defaultdict(list)
    for rep in range(args.reps):
        ops._L().denet_tune_clear()
        ops._WINO.clear()
        ops._TUNED.clear()
        run_configs(args.only)
        kern, wino = snapshot()
        for k, v in kern.items():
            votes_k[k].append(v)
        for k, v in wino.items():
            votes_w[k].append(v)
    ops._L().denet_tune_clear()
    ops._WINO.clear()
    import ctypes
    rec = []
    for k, vs in votes_k.items():
        best = collections.Counter(vs).most_common(1)[0][0]
        rec.append(list(k) + list(best))
    flat = (ctypes.c_int * (14 * len(rec)))(*[v for r in rec for v in r])
    ops.check(ops._L().denet_tune_import(flat, len(rec)), "tune_import")
    for k, vs in votes_w.items():
        ops._WINO[k] = collections.Counter(vs).most_common(1)[0][0]
    if args.merge_into:
        import json
        old = json.load(open(args.merge_into))
        kept, add, wkept, wadd = merge_records(old["kernels"], old["winograd"], rec, ops._WINO)
        old["kernels"] = sorted(kept + add)
        old["winograd"] = sorted(wkept + wadd)
        old.setdefault("meta", {})["merged"] = old["meta"].get("merged", []) + ["%s re-measured (%d reps)" % (args.only, args.reps)]
        with open(args.out, "w") as f:
            json.dump(old, f, separators=(",", ":"))
        print("merged %s into %s -> %s: %d kernel records, %d decisions" % (args.only, args.merge_into, args.out, len(old["kernels"]),
                                                                            len(old["winograd"])))
        return
    n = ops.save_tuned(args.out, {"device": torch.cuda.get_device_name(0), "reps": args.reps,
                                  "configs": ["denet34-skip b32 512", "resnet34 b64 224", "denet101-wide b16 512", "cifar3 b32"]})
    unstable = sum(1 for vs in votes_k.values() if len(set(vs)) > 1), sum(1 for vs in votes_w.values() if len(set(vs)) > 1)
    print("wrote %s: %d kernel records, %d winograd decisions (%d / %d differed between repetitions)" % (
        args.out, n, len(ops._WINO), unstable[0], unstable[1]))


if __name__ == "__main__":
    main()
