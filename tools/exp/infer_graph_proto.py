"""prototype: the inference forward up to the sparse layer as ONE captured HIP graph (torch.cuda.CUDAGraph around the ctypes launches)"""
import sys
import time

import numpy
import torch

sys.path.insert(0, ".")
from denet_amd.model import zoo

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
model = zoo.denet34(B, "skip", 512, 80)
x, metas = zoo.synthetic_batch(B, 512, 80, seed=1)
xs = torch.from_numpy(x).cuda()
layers_all = model.layers
idx = [i for i, l in enumerate(layers_all) if l.type_name == "denet-sparse"][0]
print("layers", len(layers_all), "sparse layer at", idx)
model.layers = layers_all[:idx]
for _ in range(4):
    model.forward(xs, None, train=False)
torch.cuda.synchronize()
last = model.layers[-1]
ref = last.output.data.clone() if getattr(last, "output", None) is not None and last.output.data is not None else None


def timeit(fn, n=100):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print("eager backbone + corner: %.3f ms" % timeit(lambda: model.forward(xs, None, train=False)))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    model.forward(xs, None, train=False)      # once on the capture stream (workspaces of this stream)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        model.forward(xs, None, train=False)
torch.cuda.synchronize()
print("captured")
print("graph replay: %.3f ms" % timeit(g.replay))
if ref is not None:
    g.replay()
    torch.cuda.synchronize()
    print("same output:", torch.equal(ref, last.output.data), last.type_name)
