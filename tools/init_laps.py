"""GPU box: host laps of the estimates-only initialize() and of optimize()'s ends (CUGO_INIT_TIMING) on the kitti_00 shape:
what the 0.3 - 0.4 ms between `optimize(10)` alone and the bench's `initialize(); optimize(10)` are made of."""
import importlib, os, sys, time, statistics
os.environ["CUGO_INIT_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cugo = importlib.import_module("cuda-bundle-adjustment_amd")
import numpy as np
P, L, E = 1322, 133383, 561116
d = cugo.synth(P, L, E, seed=0, n_loop_closures=4000, stereo_fraction=0.7)
ids_p, ids_l = np.arange(P, dtype=np.int32), np.arange(L, dtype=np.int32)
g = cugo.graph_from_arrays(d)
g.initialize(); g.optimize(10)
ti, to, ts = [], [], []
for r in range(12):
    t0 = time.perf_counter()
    g.set_poses(ids_p, d["pose"]); g.set_landmarks(ids_l, d["lm"])
    t1 = time.perf_counter()
    g.initialize()
    t2 = time.perf_counter()
    g.optimize(10)
    t3 = time.perf_counter()
    ts.append((t1 - t0) * 1e3); ti.append((t2 - t1) * 1e3); to.append((t3 - t2) * 1e3)
print("set_poses+set_landmarks %.3f ms  initialize %.3f ms  optimize(10) %.3f ms (medians of 12)" % (
    statistics.median(ts), statistics.median(ti), statistics.median(to)), file=sys.stderr)
