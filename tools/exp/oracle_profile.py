"""where the CPU oracle spends its time at the benchmark sizes (the slowest tests of the GPU suite wait for it):
python tools/exp/oracle_profile.py [B]  - cProfile of one free-running forward + costs of DeNet-34 skip 512x512"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from denet_amd.model import zoo
from oracle import model as OM

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
model = zoo.denet34(B, "skip", 512, class_num=80, seed=1)
x, metas = zoo.synthetic_batch(B, 512, seed=2)
om = OM.OracleModel(model.export_json(), B)
for rep in range(2):
    t = time.time()
    pr = cProfile.Profile()
    pr.enable()
    om.forward_costs(x, metas)
    pr.disable()
    print("forward_costs pass %d: %.1f s" % (rep, time.time() - t))
    pstats.Stats(pr).sort_stats("tottime").print_stats(12)
