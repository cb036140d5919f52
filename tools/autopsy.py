"""GPU box: WHERE does a deviating run differ?    The medium graph is optimised N times by fresh
optimisers with CUGO_DEBUG_KEEP=1: the solver keeps device copies of the fronts (factor L, update blocks), W, L21 and
the solution after every factorisation.  The first run's copies are the reference; when a run's chi2 trace deviates,
its copies of the first deviating iteration are written out and compared: which arrays differ, in which fronts, in how
many entries, and whether a differing entry holds the value the PREVIOUS factorisation left there (a stale read or a
lost write) or something new (a wrong computation downstream of one).
    python tools/autopsy.py N [outdir] [call:launch:workgroup of a deviation made on purpose in run 2: needs the HOOKS=1 build, CUGO_LIB=...]"""
import ctypes as C, importlib, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["CUGO_DEBUG_KEEP"] = "1"
cugo = importlib.import_module("cuda-bundle-adjustment_amd")
import numpy as np
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
OUT = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out")
SELFTEST = sys.argv[3] if len(sys.argv) > 3 else None  # e.g. 5:11:3 = call:launch:workgroup left out in run 2
L = cugo.lib()
d = cugo.synth(400, 8000, 33000, seed=11, n_loop_closures=200)
NAMES = ["fronts", "W", "L21", "x (elimination order)", "x"]


def load(dirname, call):
    raw = np.fromfile(os.path.join(dirname, "call%d.bin" % call), dtype=np.uint8)
    hdr = raw[:64].view(np.int64)
    body = raw[64:].view(np.float64)
    out, o = [], 0
    for k in range(5):
        out.append(body[o:o + hdr[k]]); o += int(hdr[k])
    return out


def run(dump_to=None, want=None):
    g = cugo.graph_from_arrays(d)
    g.initialize(); g.optimize(10)
    chi = tuple(s["chi2"] for s in g.stats())
    if dump_to is not None and (want is None or want(chi)):
        n = C.c_int(0)
        cugo.check(L.cugo_debug_dump(dump_to.encode(), C.byref(n)))
    g.close()
    return chi


def plan_fronts():
    """front -> (offset, leading dimension, 6 ncb, 6 nb) from a host-only analysis of the same pattern"""
    import devmem, oracle
    from test_host import plan_arrays
    prob = oracle.Problem(d["pose"], d["pose_fixed"], d["lm"], d["lm_fixed"], d["e_pose"], d["e_lm"], d["e_stereo"],
                          d["e_meas"], d["e_omega"], d["e_cam"])
    f = devmem.flatten(prob)
    rowptr, colind, *_ = devmem.hsc_structure(f)
    s = C.c_void_p()
    cugo.check(L.cugo_chol_create(None, C.byref(s)))
    cugo.check(L.cugo_chol_analyze(s, f["P"], rowptr.ctypes.data_as(C.POINTER(C.c_int32)), colind.ctypes.data_as(C.POINTER(C.c_int32))))
    pl = plan_arrays(L, s)
    p = C.POINTER(C.c_int32)()
    n = L.cugo_chol_plan_array(s, b"alias_of", C.byref(p))
    alias = np.ctypeslib.as_array(p, shape=(n,)).copy()
    ns = len(pl["ncb"])
    off, ld, o = np.zeros(ns, np.int64), np.zeros(ns, np.int64), 0
    for k in range(ns):  # (chol_symbolic.cpp, "5b storage")
        c = alias[k]
        if c >= 0:
            ld[k] = ld[c]; off[k] = off[c] + 6 * pl["ncb"][c] * ld[c] + 6 * pl["ncb"][c]
        else:
            ld[k] = 6 * pl["nb"][k] + 1; off[k] = o; o += ld[k] * 6 * pl["nb"][k]
    stp, tp, tf = pl["stage_task_ptr"], pl["task_ptr"], pl["task_fronts"]
    stage = np.zeros(ns, int)
    for st in range(len(stp) - 1):
        for t in range(stp[st], stp[st + 1]):
            stage[tf[tp[t]]] = st
    return pl, alias, off, ld, stage


refdir = tempfile.mkdtemp(prefix="autopsy_ref_")
ref = run(dump_to=refdir)
print("reference", ref[-1], flush=True)
found = 0
for c in range(1, N):
    devdir = tempfile.mkdtemp(prefix="autopsy_dev_")
    if SELFTEST and c == 2:  # (check of this tool: a deviation made on purpose, see tools/inject_skip.py)
        os.environ["CUGO_DEBUG_SKIP"] = SELFTEST
    chi = run(dump_to=devdir, want=lambda x: x != ref)
    os.environ.pop("CUGO_DEBUG_SKIP", None)
    if chi == ref:
        os.rmdir(devdir)
        continue
    found += 1
    it = next(i for i, (a, b) in enumerate(zip(chi, ref)) if a != b)
    print("run %d deviates from iteration %d on: chi2 %r instead of %r" % (c, it, chi[it], ref[it]), flush=True)
    pl, alias, off, ld, stage = plan_fronts()
    A, B = load(devdir, it), load(refdir, it)
    prevB = load(refdir, it - 1) if it > 0 else None
    for k, name in enumerate(NAMES):
        neq = np.flatnonzero(A[k] != B[k])
        line = "  %-22s %9d entries, %8d differ" % (name, len(A[k]), len(neq))
        if len(neq) and prevB is not None:
            stale = int(np.count_nonzero(A[k][neq] == prevB[k][neq]))
            line += "; %d of them hold the value of iteration %d (the factorisation before)" % (stale, it - 1)
        print(line)
        if len(neq) and name == "fronts":
            # which fronts (storage ranges; a front stored in its child's update block shares the child's range)
            rows = []
            for f_ in range(len(off)):
                if alias[f_] >= 0:
                    continue
                size = ld[f_] * 6 * pl["nb"][f_]
                inside = neq[(neq >= off[f_]) & (neq < off[f_] + size)]
                if len(inside):
                    col = (inside - off[f_]) // ld[f_]; row = (inside - off[f_]) % ld[f_]
                    rows.append((stage[f_], f_, len(inside), int(col.min()), int(col.max()), int(row.min()), int(row.max()),
                                 6 * pl["ncb"][f_], 6 * pl["nb"][f_]))
            rows.sort()
            for r in rows[:12]:
                print("      stage %2d front %3d: %7d entries differ, columns %d..%d rows %d..%d   (pivot columns %d, all %d)" % r)
            f0 = rows[0][1]
            inside = neq[(neq >= off[f0]) & (neq < off[f0] + ld[f0] * 6 * pl["nb"][f0])][:16]
            for i in inside:
                print("        front %d column %d row %d: %r instead of %r%s" % (f0, (i - off[f0]) // ld[f0], (i - off[f0]) % ld[f0], A[k][i], B[k][i],
                      ("  (iteration %d had %r)" % (it - 1, prevB[k][i])) if prevB is not None else ""))
    if found == 1:  # the first deviating run's arrays go home with the call (gpurun_out is capped at 64 MiB)
        keepdir = os.path.join(OUT, "autopsy_run%d" % c)
        os.makedirs(keepdir, exist_ok=True)
        for nm, src in (("dev", devdir), ("ref", refdir)):
            a = np.fromfile(os.path.join(src, "call%d.bin" % it), dtype=np.uint8)
            if a.nbytes < 25e6:
                a.tofile(os.path.join(keepdir, "%s_call%d.bin" % (nm, it)))
    if found >= 6:
        break
print("runs", N, "deviating", found)
