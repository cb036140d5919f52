import importlib, sys, time, os
sys.path.insert(0, os.getcwd())
cugo = importlib.import_module("cuda-bundle-adjustment_amd")
P,L,E = 10000,1000000,5000000
d = cugo.synth(P, L, E, seed=10000, n_loop_closures=0, stereo_fraction=0.0)
g = cugo.graph_from_arrays(d)
g.initialize(); g.optimize(1)
os.environ["CUGO_NO_STRUCTURE_REUSE"]="1"; os.environ["CUGO_INIT_TIMING"]="1"
t=time.time(); g.initialize(); t1=time.time(); g.optimize(1); t2=time.time()
print("init %.1f ms, optimize(1) incl structure %.1f ms" % ((t1-t)*1e3,(t2-t1)*1e3))
