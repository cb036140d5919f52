"""GPU box: one medium graph (400 poses / 8 000 landmarks / 33 000 edges, 200 loop closures — the shape on which the
suite's rare run-to-run differences were seen) optimised N times per configuration from the same estimates by NEW
optimisers; counts the runs whose chi2 trace or final poses differ from the first run bit for bit.
    python tools/repro_medium.py N [VAR=val,VAR=val ...]
    python tools/repro_medium.py N --interleave CFG CFG ...   (one run of every configuration in turn, N rounds: the
                                                              deviation comes in bursts, so blocks of runs cannot be compared)"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("CUGO_DEBUG_HASH"):  # in-stream checksums exist in the hooks build only (make HOOKS=1)
    os.environ.setdefault("CUGO_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                   "cuda-bundle-adjustment_amd", "libcugo_hip_hooks.so"))
cugo = importlib.import_module("cuda-bundle-adjustment_amd")
import numpy as np
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
interleave = "--interleave" in sys.argv
big = "--10k" in sys.argv   # the 10 000-pose / 1 M-landmark / 5 M-edge graph of test_synth10k_full_size instead
args = [a for a in sys.argv[2:] if a not in ("--interleave", "--10k")]
configs = [dict(kv.split("=") for kv in a.split(",") if kv) for a in args] or [{}]
d = (cugo.synth(10000, 1000000, 5000000, seed=10000, n_loop_closures=0, stereo_fraction=0.0) if big else
     cugo.synth(400, 8000, 33000, seed=11, n_loop_closures=200))
if interleave:
    refs, bads = [None] * len(configs), [0] * len(configs)
    for c in range(N):
        for q, cfg in enumerate(configs):
            for k, v in cfg.items():
                os.environ[k] = v
            g = cugo.graph_from_arrays(d)
            g.initialize(); g.optimize(10)
            cur = (tuple(s["chi2"] for s in g.stats()), g.poses().copy())
            g.close()
            for k in cfg:
                os.environ.pop(k, None)
            if refs[q] is None:
                refs[q] = cur
            elif cur[0] != refs[q][0] or not np.array_equal(cur[1], refs[q][1]):
                bads[q] += 1
    for q, cfg in enumerate(configs):
        print("interleaved", cfg, "rounds", N, "deviating", bads[q], flush=True)
    sys.exit(0)
for cfg in configs:
    for k, v in cfg.items():
        os.environ[k] = v
    f32 = cfg.get("F32") == "1"
    ref, bad = None, []
    hf = os.environ.get("CUGO_DEBUG_HASH")
    SLOTS = (["Hpp", "b", "Hll", "Hpl", "Hsc|bsc", "T", "invHll", "x_p", "x_l", "poses'", "landmarks'", "fronts after assembly",
              "W after forward", "L21 after forward", "fronts after forward", "x after backward"] +
             ["W after potrf of stage %d" % i for i in range(24)] + ["fronts after tiles of stage %d" % i for i in range(24)])
    # order in which the arrays come into being within an iteration
    ORDER = [0, 1, 2, 3, 4, 5, 6, 11] + [k for i in range(24) for k in (16 + i, 40 + i)] + [12, 13, 14, 15, 7, 8, 9, 10]

    def last_hashes():
        blocks = open(hf).read().split("run\n")
        return [ln.split() for ln in blocks[-1].strip().splitlines()]
    href = None
    for c in range(N):
        if hf and os.path.exists(hf):
            os.remove(hf)
        g = cugo.graph_from_arrays(d)
        if f32:
            g.set_float32(True)
        g.initialize(); g.optimize(10)
        st = g.stats()
        cur = (tuple(s["chi2"] for s in st), g.poses().copy(), [(s["lam"], s["rho"], s["trials"]) for s in st])
        g.close()
        hcur = last_hashes() if hf else None
        if c % 50 == 49:
            print("  ... %d runs, %d deviating" % (c + 1, len(bad)), flush=True)  # (a silent GPU job is taken to be hung)
        if ref is None:
            ref, href = cur, hcur
        elif cur[0] != ref[0] or not np.array_equal(cur[1], ref[1]):
            it = next((i for i, (a, b) in enumerate(zip(cur[0], ref[0])) if a != b), -1)
            bad.append((c, it, abs(cur[0][it] - ref[0][it]) / ref[0][it] if it >= 0 else 0.0))
            lo = max(it - 2, 0)
            print("  run", c, "first chi2 difference at iteration", it)
            if hf:
                first = next(((i, k) for i in range(min(len(href), len(hcur))) for k in ORDER if href[i][k] != hcur[i][k]), None)
                print("    first differing array: iteration %s, %s" % ((first[0], SLOTS[first[1]]) if first else ("-", "none")),
                      " all differing in that iteration:", [SLOTS[k] for k in ORDER if first and href[first[0]][k] != hcur[first[0]][k]][:12])
            for i in range(lo, min(it + 2, len(ref[0]))):
                print("    it %d  chi2 %r / %r   (lam, rho, trials) %r / %r" % (i, cur[0][i], ref[0][i], cur[2][i], ref[2][i]), flush=True)
    print(cfg, "runs", N, "deviating", len(bad), bad[:8], flush=True)
    for k in cfg:
        os.environ.pop(k, None)
