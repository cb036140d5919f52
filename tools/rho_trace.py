"""GPU box: gain ratio and damping per LM iteration on the bench shapes (does the damping update hit its lower clamp,
the value the speculative build pass predicts?)"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cugo = importlib.import_module("cuda-bundle-adjustment_amd")
SHAPES = {"kitti00": (1322, 133383, 561116, 0, 4000, 0.7), "synth10k": (10000, 1000000, 5000000, 10000, 0, 0.0)}
for wl in sys.argv[1:] or ["kitti00", "synth10k"]:
    P, L, E, seed, lc, sf = SHAPES[wl]
    d = cugo.synth(P, L, E, seed=seed, n_loop_closures=lc, stereo_fraction=sf)
    g = cugo.graph_from_arrays(d)
    g.initialize(); g.optimize(10)
    for s in g.stats():
        a = 1 - (2 * s["rho"] - 1) ** 3
        print(wl, s["iteration"], "chi2 %.6g" % s["chi2"], "lam %.4g" % s["lam"], "rho %.4f" % s["rho"], "trials", s["trials"],
              "clamp" if a <= 1 / 3 else "factor %.3f" % min(a, 2 / 3))
    g.close()
