"""GPU box: per-section host times (CUGO_INIT_TIMING) of a full re-flatten + structure rebuild on the
kitti_00 shape.   python tools/cold_small.py"""
import importlib, sys, time, os
sys.path.insert(0, os.getcwd())
cugo = importlib.import_module("cuda-bundle-adjustment_amd")
d = cugo.synth(1322, 133383, 561116, seed=0, n_loop_closures=4000, stereo_fraction=0.7)
g = cugo.graph_from_arrays(d)
g.initialize(); g.optimize(1)
g.set_option("structure_reuse", 0); g.set_option("flatten_reuse", 0)
for rep in range(3):
    if rep == 2:
        g.set_option("init_timing", 1)
        os.environ["CUGO_INIT_TIMING"] = "1"  # (the symbolic analysis prints its own laps when this is set)
    t = time.time(); g.initialize(); t1 = time.time(); g.optimize(1); t2 = time.time()
    print("init %.2f ms, optimize(1) incl structure %.2f ms" % ((t1 - t) * 1e3, (t2 - t1) * 1e3), flush=True)
