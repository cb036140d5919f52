"""GPU box: k_up_potrf with chosen waves / workgroups put to sleep at chosen points (CUGO_DEBUG_DELAY=1..6, read when a
plan is uploaded): the factorisation of a fixed system and the whole optimisation of the medium graph must give the
same bits whatever runs late.
    python tools/delay_check.py
    CUGO_LIB=.../libcugo_hip_hooks.so python tools/delay_check.py CUGO_DEBUG_ZERO_LDS 0 1 2     (make HOOKS=1; any other diagnosis switch read when a plan is uploaded and
                                                             its values: here the kernels' LDS pre-filled with zeros / NaNs)"""
import ctypes as C, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
cugo = importlib.import_module("cuda-bundle-adjustment_amd")
import numpy as np
import devmem, oracle
L = cugo.lib()
VAR = sys.argv[1] if len(sys.argv) > 1 else "CUGO_DEBUG_DELAY"
VALUES = sys.argv[2:] if len(sys.argv) > 2 else ["0", "1", "2", "3", "4", "5", "6", "0"]
ok = True
for shape in [(400, 8000, 33000, 11, 200), (1322, 133383, 561116, 0, 4000)]:
    P_, L_, E_, seed, lc = shape
    d = cugo.synth(P_, L_, E_, seed=seed, n_loop_closures=lc, stereo_fraction=0.7 if P_ > 1000 else 0.0)
    ref = None
    for delay in VALUES:
        os.environ[VAR] = str(delay)
        g = cugo.graph_from_arrays(d)
        g.initialize(); g.optimize(6)
        cur = (tuple(s["chi2"] for s in g.stats()), g.poses().copy())
        g.close()
        if ref is None:
            ref = cur
        same = cur[0] == ref[0] and np.array_equal(cur[1], ref[1])
        ok = ok and same
        print("graph %d poses  %s=%s  %s  chi2 %r" % (P_, VAR, delay, "same" if same else "DIFFERENT", cur[0][-1]), flush=True)
print(VAR, "check", "ok" if ok else "FAILED")
