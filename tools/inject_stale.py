"""GPU box: which 128-byte line, seen ONE LAUNCH LATE, gives the chi2 values the rare run-to-run deviation produces?
(The sibling of tools/inject_skip.py, one level finer: CUGO_DEBUG_STALE=call:kind:line shows the launch that follows a
line's producer the previous factorisation's content of that line — W of a front to the level's tile launch (kind 0), x of
a front to the backward launch of the next level down (kind 1) — and the right content to everything later.)
    python tools/inject_stale.py [seconds] [kind] [call call ...]
--- the text of tools/inject_skip.py follows ---
GPU box: which workgroup's work, left undone ONCE, gives the chi2 values the rare run-to-run deviation produces?
  The deviating runs of tools/repro_medium.py end their first deviating iteration on a handful
of FIXED values (tools/deviation_alternates.txt: one of them in three quarters of the cases).  Here the supposed
failure is made on purpose: with CUGO_DEBUG_SKIP=call:launch:workgroup one workgroup of one launch of one
factorisation returns at once, so whatever it would have written keeps the value of the factorisation before —
for EVERY workgroup of every launch of every factorisation of optimize(10), one run each — and the chi2 of that
iteration is looked up among the alternates.  A match names the kernel, the level and the workgroup.
    CUGO_LIB=.../libcugo_hip.so python tools/inject_skip.py [seconds] [call call ...]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
cugo = importlib.import_module("cuda-bundle-adjustment_amd")
import numpy as np
BUDGET = float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
KIND = int(sys.argv[2]) if len(sys.argv) > 2 else 1
CALLS = [int(a) for a in sys.argv[3:]] or [8, 1, 5, 3, 7, 2, 6, 4, 9, 0]
alts, ref_line = {}, None
for ln in open(os.path.join(ROOT, "tools", "deviation_alternates.txt")):
    if ln.startswith("# reference trace:"):
        ref_line = [float(x) for x in ln.split(":")[1].split()]
    elif not ln.startswith("#") and ln.strip():
        it, v, n = ln.split()
        alts.setdefault(int(it), {})[float(v)] = int(n)
d = cugo.synth(400, 8000, 33000, seed=11, n_loop_closures=200)


def run(niter, skip=None, dump=None):
    for k, v in (("CUGO_DEBUG_STALE", skip), ("CUGO_DEBUG_SKIP_DUMP", dump)):
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    g = cugo.graph_from_arrays(d)
    g.initialize(); g.optimize(niter)
    st = g.stats()
    g.close()
    return [s["chi2"] for s in st], [s["trials"] for s in st]


dump = "/tmp/inject_launches.txt"
ref, trials = run(10, dump=dump)
import ctypes as C
sys.path.insert(0, os.path.join(ROOT, "tests"))
import devmem, oracle
prob = oracle.Problem(d["pose"], d["pose_fixed"], d["lm"], d["lm_fixed"], d["e_pose"], d["e_lm"], d["e_stereo"], d["e_meas"],
                      d["e_omega"], d["e_cam"])
f = devmem.flatten(prob)
rowptr, colind, *_ = devmem.hsc_structure(f)
sv = C.c_void_p()
Lb = cugo.lib()
cugo.check(Lb.cugo_chol_create(None, C.byref(sv)))
cugo.check(Lb.cugo_chol_analyze(sv, f["P"], rowptr.ctypes.data_as(C.POINTER(C.c_int32)), colind.ctypes.data_as(C.POINTER(C.c_int32))))
p = C.POINTER(C.c_int32)()
ncb = np.ctypeslib.as_array(p, shape=(Lb.cugo_chol_plan_array(sv, b"ncb", C.byref(p)),)).copy()
nlines = (int(sum(((6 * int(c) + 15) // 16 * 16) ** 2 for c in ncb)) if KIND == 0 else 6 * f["P"]) // 16
table = [(KIND, "W" if KIND == 0 else "x", nlines, 0)]
print("reference trace equals the one of the alternates file:", ref == ref_line, " lines:", nlines, flush=True)
t0 = time.time()
out = open(os.path.join(ROOT, "gpurun_out", "inject_stale.txt"), "a") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else None
nrun = nmatch = 0
for call in CALLS:
    seen = {}
    for launch, name, grid, first in table:
        for wg in range(grid):
            if time.time() - t0 > BUDGET:
                break
            try:
                chi, _ = run(call + 1, skip="%d:%d:%d" % (call, launch, wg))
                v = chi[call]
            except Exception as e:  # (a factorisation that fails on purpose-made garbage)
                v = float("nan")
            nrun += 1
            kind = name + (" (front)" if first and wg < first else " (extra)" if first else "")
            if out:
                out.write("%d %d %s %d %r\n" % (call, launch, name, wg, v))
            if v != ref[call]:
                seen[v] = seen.get(v, 0) + 1
            if v in alts.get(call, {}):
                nmatch += 1
                print("MATCH  iteration %d  chi2 %r (seen %d times in the wild)  <-  kind %d %s line %d of %d" %
                      (call, v, alts[call][v], launch, kind, wg, grid), flush=True)
    unchanged = sum(t[2] for t in table) - sum(seen.values())
    print("iteration %d done: %d distinct results besides the reference, %d lines whose staleness changes nothing; alternates of this "
          "iteration not reproduced: %s" % (call, len(seen), unchanged, [v for v in alts.get(call, {}) if v not in seen]), flush=True)
    if time.time() - t0 > BUDGET:
        print("time budget used up")
        break
print("runs %d, matches %d" % (nrun, nmatch))
