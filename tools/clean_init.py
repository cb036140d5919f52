import importlib, sys, time, os
sys.path.insert(0, os.getcwd())
cugo = importlib.import_module("cuda-bundle-adjustment_amd")
import numpy as np
d = cugo.synth(1322, 133383, 561116, seed=0, n_loop_closures=4000, stereo_fraction=0.7)
g = cugo.graph_from_arrays(d)
g.initialize(); g.optimize(2)
ids_p, ids_l = np.arange(1322, dtype=np.int32), np.arange(133383, dtype=np.int32)
for rep in range(4):
    g.set_poses(ids_p, d["pose"]); g.set_landmarks(ids_l, d["lm"])
    if rep == 3: os.environ["CUGO_INIT_TIMING"] = "1"
    t = time.perf_counter(); g.initialize(); t1 = time.perf_counter()
    print("initialize %.3f ms" % ((t1 - t) * 1e3), flush=True)
