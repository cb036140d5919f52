"""cold-call cost on a local-BA sized graph: a NEW optimiser per call, as ORB-SLAM2 does"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cugo = importlib.import_module("cuda-bundle-adjustment_amd")
d = cugo.synth(30, 3000, 12600, seed=1, n_loop_closures=0, stereo_fraction=0.7)
ti, to, tc, tb = [], [], [], []
for c in range(60):
    t0 = time.perf_counter(); g = cugo.graph_from_arrays(d); t1 = time.perf_counter()
    g.initialize(); t2 = time.perf_counter()
    g.optimize(5); t3 = time.perf_counter()
    g.close(); t4 = time.perf_counter()
    tb.append(t1 - t0); ti.append(t2 - t1); to.append(t3 - t2); tc.append(t4 - t3)
f = lambda v: 1e3 * float(np.median(v[10:]))
print("build graph %.2f ms | initialize %.2f ms | optimize(5) %.2f ms | destroy %.2f ms" % (f(tb), f(ti), f(to), f(tc)))
g = cugo.graph_from_arrays(d); g.initialize(); g.optimize(5)
t0 = time.perf_counter(); g.initialize(); t1 = time.perf_counter(); g.optimize(5); t2 = time.perf_counter()
print("same optimiser again: initialize %.2f ms | optimize(5) %.2f ms" % (1e3 * (t1 - t0), 1e3 * (t2 - t1)))
print(g.time_profile())
