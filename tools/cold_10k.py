"""GPU box: per-section host times of a new-graph call (full flatten + structure rebuild) on the 10k / 1M / 5M graph."""
import importlib, sys, time, os
sys.path.insert(0, os.getcwd())
cugo = importlib.import_module("cuda-bundle-adjustment_amd")
P,L,E = 10000,1000000,5000000
d = cugo.synth(P, L, E, seed=10000, n_loop_closures=0, stereo_fraction=0.0)
g = cugo.graph_from_arrays(d)
g.initialize(); g.optimize(1)
g.set_option("structure_reuse", 0); g.set_option("flatten_reuse", 0); g.set_option("init_timing", 1)
os.environ["CUGO_INIT_TIMING"] = "1"  # (the symbolic analysis prints its own laps when this is set)
t=time.time(); g.initialize(); t1=time.time(); g.optimize(1); t2=time.time()
print("init %.1f ms, optimize(1) incl structure %.1f ms" % ((t1-t)*1e3,(t2-t1)*1e3))
