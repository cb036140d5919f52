"""GPU box, hooks build (make HOOKS=1; the product library has no delay patterns): the kernels of the factorisation with
chosen waves / workgroups put to sleep at chosen points — CUGO_DEBUG_DELAY=1..7: roles of k_up_potrf, 10..14: chosen
waves behind every barrier of the trsm / syrk / fused-tile / backward kernels; read when a plan is uploaded.  The whole
optimisation of the medium graph (and of the kitti_00 shape, unless --medium) must give the same bits whatever runs late.
    python tools/delay_check.py [--medium]
    python tools/delay_check.py CUGO_DEBUG_ZERO_LDS 0 1 2     (any other diagnosis switch read when a plan is uploaded and
                                                             its values: here the kernels' LDS pre-filled with zeros / NaNs)
    python tools/delay_check.py CUGO_DEBUG_DELAY 0 8          (the negative control: 8 = pattern 7 on the kernel as it was
                                                             before the race fix of round 3 — expected to print DIFFERENT)
Exit code 1 if any value gave other bits than the first."""
import ctypes as C, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("CUGO_LIB", os.path.join(ROOT, "cuda-bundle-adjustment_amd", "libcugo_hip_hooks.so"))
cugo = importlib.import_module("cuda-bundle-adjustment_amd")
import numpy as np
import devmem, oracle
L = cugo.lib()
ARGS = [a for a in sys.argv[1:] if a != "--medium"]
VAR = ARGS[0] if ARGS else "CUGO_DEBUG_DELAY"
VALUES = ARGS[1:] if len(ARGS) > 1 else ["0", "1", "2", "3", "4", "5", "6", "7", "10", "11", "12", "13", "14", "0"]
SHAPES = [(400, 8000, 33000, 11, 200), (1322, 133383, 561116, 0, 4000)]
if "--medium" in sys.argv:
    SHAPES = SHAPES[:1]
ok = True
for shape in SHAPES:
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
sys.exit(0 if ok else 1)
