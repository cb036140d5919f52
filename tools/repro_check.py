"""GPU box: run-to-run reproducibility probe.  For every configuration (environment switches) a fresh graph is
optimised, then re-optimised twice from the same estimates; a second fresh graph repeats it.  Prints whether
the chi2 traces and final poses agree bit for bit.
    python tools/repro_check.py [workload] [VAR=val,VAR=val ...]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cugo = importlib.import_module("cuda-bundle-adjustment_amd")
import numpy as np
SHAPES = {"kitti00": (1322, 133383, 561116, 0, 4000, 0.7), "synth10k": (10000, 1000000, 5000000, 10000, 0, 0.0)}
wl = sys.argv[1] if len(sys.argv) > 1 else "synth10k"
configs = [dict(kv.split("=") for kv in a.split(",") if kv) for a in sys.argv[2:]] or [{}]
P, L, E, seed, lc, sf = SHAPES[wl]
d = cugo.synth(P, L, E, seed=seed, n_loop_closures=lc, stereo_fraction=sf)
ids_p, ids_l = np.arange(P, dtype=np.int32), np.arange(L, dtype=np.int32)
for cfg in configs:
    for k, v in cfg.items():
        os.environ[k] = v
    runs = []
    for fresh in range(2):
        g = cugo.graph_from_arrays(d)
        for rep in range(3):
            if rep and os.environ.get("REPRO_SLEEP"):
                time.sleep(float(os.environ["REPRO_SLEEP"]))  # the device idles (clocks drop) between the runs
            g.set_poses(ids_p, d["pose"]); g.set_landmarks(ids_l, d["lm"])
            g.initialize(); g.optimize(10)
            runs.append(([s["chi2"] for s in g.stats()], [s["trials"] for s in g.stats()], g.poses().copy()))
        g.close()
    ref = runs[0]
    line = []
    for chi, tr, pose in runs[1:]:
        same = chi == ref[0] and np.array_equal(pose, ref[2])
        first = next((i for i, (a, b) in enumerate(zip(chi, ref[0])) if a != b), -1)
        line.append("same" if same else "DIFF@%d(%.1e)" % (first, abs(chi[first] - ref[0][first]) / abs(ref[0][first]) if first >= 0 else 0))
    print(cfg, "chi trace", [repr(c) for c in ref[0]], "trials", ref[1])
    print(cfg, "chi_last %.10g" % ref[0][-1], "nan" if not np.isfinite(ref[0][-1]) else "", " ".join(line), flush=True)
    for k in cfg:
        os.environ.pop(k, None)
