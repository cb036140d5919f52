"""GPU box: A/B of one environment switch inside ONE process — one optimiser per value of the switch
(the switch is read when the optimiser is initialised / its plan analysed), timed alternately, so
box-to-box and run-to-run drift cancels.  Prints median and minimum of `initialize(); optimize(10)`.

    python tools/ab_env.py CUGO_XCD_AFFINITY 1 0 [--workload synth10k] [--reps 30]
    (a value `unset` removes the variable: some switches only test for presence)
"""
import importlib, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cugo = importlib.import_module("cuda-bundle-adjustment_amd")
import numpy as np

SHAPES = {"kitti00": (1322, 133383, 561116, 0, 4000, 0.7), "synth10k": (10000, 1000000, 5000000, 10000, 0, 0.0)}


def main():
    a = sys.argv[1:]
    wl = "kitti00"
    reps = 30
    if "--workload" in a:
        i = a.index("--workload"); wl = a[i + 1]; del a[i:i + 2]
    dirty = "--dirty" in a   # time the new-graph regime: full flattening + structure rebuild in every call
    if dirty:
        a.remove("--dirty")
    if "--reps" in a:
        i = a.index("--reps"); reps = int(a[i + 1]); del a[i:i + 2]
    var, vals = a[0], a[1:]
    P, L, E, seed, lc, sf = SHAPES[wl]
    d = cugo.synth(P, L, E, seed=seed, n_loop_closures=lc, stereo_fraction=sf)
    ids_p, ids_l = np.arange(P, dtype=np.int32), np.arange(L, dtype=np.int32)
    graphs = []
    def put(v):
        if v == "unset":
            os.environ.pop(var, None)
        else:
            os.environ[var] = v

    for v in vals:
        put(v)
        g = cugo.graph_from_arrays(d)
        g.initialize(); g.optimize(10)
        graphs.append(g)
    times = [[] for _ in vals]
    if dirty:
        for g in graphs:
            g.set_option("structure_reuse", 0); g.set_option("flatten_reuse", 0)
    for r in range(reps):
        for k, g in enumerate(graphs):
            put(vals[k])
            g.set_poses(ids_p, d["pose"]); g.set_landmarks(ids_l, d["lm"])
            t = time.perf_counter()
            g.initialize(); g.optimize(10)
            times[k].append((time.perf_counter() - t) * 1e3)
    for k, v in enumerate(vals):
        t = times[k]
        print("%s=%s  median %.3f ms  min %.3f ms  (n=%d)" % (var, v, statistics.median(t), min(t), len(t)))


if __name__ == "__main__":
    main()
