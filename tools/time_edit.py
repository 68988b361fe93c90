import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
from denet_amd.model import zoo
from denet_amd.layer import denet_sparse as ds
import cProfile, pstats
m = zoo.denet34(32); dns = m.layers[31]
x, metas = zoo.synthetic_batch(32)
prs = [numpy.zeros(0)] * 32; boxes = [numpy.zeros((0, 4))] * 32
random.seed(1)
for _ in range(5):
    t = time.perf_counter(); dns.edit_samples(prs, boxes, metas); print("edit %.2f ms" % ((time.perf_counter() - t) * 1e3))
t = time.perf_counter(); mi = ds.PyRandomMirror(); t1 = time.perf_counter(); v = mi.doubles(73000); t2 = time.perf_counter(); mi.push(); t3 = time.perf_counter()
print("pull %.2f doubles %.2f push %.2f ms" % ((t1 - t) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
pr = cProfile.Profile(); pr.enable(); dns.edit_samples(prs, boxes, metas); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(8)
