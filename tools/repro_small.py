"""GPU box: does a small graph give the same trajectory every time when optimisers come and go (the allocator
hands every new optimiser blocks that other graphs used before)?  For every configuration (environment
switches, comma separated) N cycles of the soak test's four graph kinds; counts the cycles whose chi2 trace
differs from the first one of their kind.
    python tools/repro_small.py N [VAR=val,VAR=val ...]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cugo = importlib.import_module("cuda-bundle-adjustment_amd")
import numpy as np
N = int(sys.argv[1]) if len(sys.argv) > 1 else 96
configs = [dict(kv.split("=") for kv in a.split(",") if kv) for a in sys.argv[2:]] or [{}]
KINDS = [(30, 3000, 12600, 0), (160, 2500, 10500, 80), (200, 3000, 12500, 100), (400, 8000, 33000, 200)]
data = [cugo.synth(P, L, E, seed=k + 1, n_loop_closures=lc, stereo_fraction=0.7) for k, (P, L, E, lc) in enumerate(KINDS)]
for cfg in configs:
    for k, v in cfg.items():
        os.environ[k] = v
    first, bad = {}, []
    for c in range(N):
        kind, f32 = c % 4, c % 3 == 0
        g = cugo.graph_from_arrays(data[kind])
        if f32:
            g.set_float32(True)
        g.initialize(); g.optimize(3)
        chi = tuple(s["chi2"] for s in g.stats())
        lam = tuple(s["lam"] for s in g.stats())
        key = (kind, f32)
        if key in first and (chi, lam) != first[key]:
            it = next(i for i, (a, b) in enumerate(zip(chi, first[key][0])) if a != b) if chi != first[key][0] else -1
            bad.append((c, key, it, abs(chi[max(it, 0)] - first[key][0][max(it, 0)]) / first[key][0][max(it, 0)]))
        first.setdefault(key, (chi, lam))
        g.initialize(); g.optimize(2)
        g.close()
    print(cfg, "cycles", N, "deviating", len(bad), bad[:6], "nan" if any(not np.isfinite(v[0][-1]) for v in first.values()) else "", flush=True)
    for k in cfg:
        os.environ.pop(k, None)
