"""Where the HOST time of a training step goes (cProfile over steady-state steps): the forward pass issues ~250 kernels in
~11 ms of device time, so the Python / ctypes cost per launch decides whether the device ever waits for the host."""
import cProfile, pstats, os, sys, random, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from denet_amd.model import zoo
B = 32
model = zoo.denet34(B, "skip", 512, class_num=80, seed=1)
x, metas = zoo.synthetic_batch(B, 512, 80, seed=1)
xd = torch.from_numpy(x).cuda()
model.build_train_func("nesterov")
random.seed(1)
for it in range(4):
    model.train_step(xd, metas, 0, it, 0.1, [0.9], 1e-4)
torch.cuda.synchronize()
# host issue time of forward / backward without waiting for the device
t0 = time.perf_counter(); ctx = model.forward(xd, metas, train=True); t1 = time.perf_counter(); model.backward(ctx); t2 = time.perf_counter()
torch.cuda.synchronize(); t3 = time.perf_counter()
print("host issue: forward %.2f ms (incl. the RoI round trip), backward %.2f ms; device drained after %.2f ms" % (1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t0)))
pr = cProfile.Profile()
pr.enable()
for it in range(4, 10):
    model.train_step(xd, metas, 0, it, 0.1, [0.9], 1e-4)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(18)
