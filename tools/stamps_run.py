"""GPU box, stamps build: one kitti00-shaped optimize(2); the library prints the cycle stamps of the
last launch of every Cholesky kernel at its 5th factorisation.
    CUGO_LIB=cuda-bundle-adjustment_amd/libcugo_hip_stamps.so CUGO_DEBUG_STAMPS=1 python tools/stamps_run.py [workload]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cugo = importlib.import_module("cuda-bundle-adjustment_amd")
SHAPES = {"kitti00": (1322, 133383, 561116, 0, 4000, 0.7), "synth10k": (10000, 1000000, 5000000, 10000, 0, 0.0)}
P, L, E, seed, lc, sf = SHAPES[sys.argv[1] if len(sys.argv) > 1 else "kitti00"]
d = cugo.synth(P, L, E, seed=seed, n_loop_closures=lc, stereo_fraction=sf)
g = cugo.graph_from_arrays(d)
g.initialize()
g.optimize(6)
print([round(s["chi2"], 3) for s in g.stats()])
g.close()
