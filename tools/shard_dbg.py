"""GPU box: one sharded-in-threads run (debugging aid).  python tools/shard_dbg.py P L E world niter"""
import importlib, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
cugo = importlib.import_module("cuda-bundle-adjustment_amd")
import test_gpu
P, L, E, world, niter = [int(a) for a in sys.argv[1:6]]
d = cugo.synth(P, L, E, seed=P, n_loop_closures=0, stereo_fraction=0.0)
res = test_gpu.run_sharded_in_threads(d, world, niter, want_sstats=True)
print("ok", [round(s["chi2"], 3) for s in res[0]["stats"]], res[0]["sstats"]["chol_bcasts"], flush=True)
