import os, sys, time, random, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy, torch
from denet_amd.model import zoo
from denet_amd.layer import denet_sparse as ds

B = 32
model = zoo.denet34(B, "skip", 512)
random.seed(1)
x, metas = zoo.synthetic_batch(B, 512)
model.build_train_func("nesterov")
xd = torch.from_numpy(x).cuda()
dns = model.layers[31]
orig_dev, orig_edit, orig_up = dns._device_samples, dns.edit_samples, dns._upload_boxes
T = {"dev": [], "edit": [], "up": [], "dnd": [], "step": []}
def timed(name, fn):
    def w(*a, **k):
        t = time.perf_counter(); r = fn(*a, **k); T[name].append((time.perf_counter() - t) * 1e3); return r
    return w
dns._device_samples = timed("dev", orig_dev); dns.edit_samples = timed("edit", orig_edit); dns._upload_boxes = timed("up", orig_up)
dnd = model.layers[40]; dnd.build_targets = timed("dnd", dnd.build_targets)
if len(sys.argv) > 1 and sys.argv[1] == "nogc":
    gc.collect(); gc.freeze(); gc.disable()
import cProfile, pstats
for it in range(25):
    if it == 12:
        prof = cProfile.Profile()
        def pe(*a, **k):
            prof.enable(); r = orig_edit(*a, **k); prof.disable(); return r
        dns.edit_samples = pe
    if it == 13:
        dns.edit_samples = timed("edit", orig_edit)
        pstats.Stats(prof).sort_stats("tottime").print_stats(10)
    torch.cuda.synchronize(); t = time.perf_counter()
    model.train_step(xd, metas, 0, it, 0.1, [0.9], 1e-4)
    torch.cuda.synchronize(); T["step"].append((time.perf_counter() - t) * 1e3)
for k, v in T.items():
    v = v[3:]
    print("%-5s mean %.2f  min %.2f  max %.2f   %s" % (k, sum(v) / len(v), min(v), max(v), " ".join("%.1f" % a for a in v[:22])))
print(open("/sys/fs/cgroup/cpu.stat").read().split("\n")[6:8])
print("gc counts", gc.get_count(), "threads", torch.get_num_threads())
