"""GPU box: WHERE does a deviating run differ?    A graph is optimised N times by fresh optimisers
with CUGO_DEBUG_KEEP=1: every solver keeps device copies of the fronts (factor, update blocks), W, L21 and the solution
after each of its factorisations.  The first run is the reference (its graph stays open); when a run's chi2 trace
deviates, its copies of the first deviating iteration are compared with the reference's: which arrays differ, in which
fronts of which level, in how many entries, and whether a differing entry holds the value the PREVIOUS factorisation
left there (a stale read or a lost write) or something new (computed downstream of one).
    python tools/autopsy.py N [--10k] [--inject call:launch:workgroup]
--10k: the 10 000-pose graph of test_synth10k_full_size (deviates in ~1 % of the runs on any box) instead of the
400-pose one;  --inject: run 2 gets a deviation made on purpose (CUGO_DEBUG_SKIP, needs the HOOKS=1 build)"""
import ctypes as C, importlib, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CUGO_DEBUG_KEEP"] = "1"
# the arrays are kept (and cugo_debug_* exported) by the hooks build only: make -C cuda-bundle-adjustment_amd HOOKS=1
os.environ.setdefault("CUGO_LIB", os.path.join(ROOT, "cuda-bundle-adjustment_amd", "libcugo_hip_hooks.so"))
cugo = importlib.import_module("cuda-bundle-adjustment_amd")
import numpy as np
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
BIG = "--10k" in sys.argv
INJECT = sys.argv[sys.argv.index("--inject") + 1] if "--inject" in sys.argv else None
L = cugo.lib()
d = (cugo.synth(10000, 1000000, 5000000, seed=10000, n_loop_closures=0, stereo_fraction=0.0) if BIG else
     cugo.synth(400, 8000, 33000, seed=11, n_loop_closures=200))
NAMES = ["fronts", "W", "L21", "x (elimination order)", "x"]
TMP = tempfile.mkdtemp(prefix="autopsy_")


def slot(which, call):
    path = os.path.join(TMP, "slot.bin")
    cugo.check(L.cugo_debug_dump_call(which, call, path.encode()))
    raw = np.fromfile(path, dtype=np.uint8)
    os.remove(path)
    hdr, body = raw[:64].view(np.int64), raw[64:].view(np.float64)
    out, o = [], 0
    for k in range(5):
        out.append(body[o:o + hdr[k]]); o += int(hdr[k])
    return out


def plan(name):
    p = C.POINTER(C.c_int32)()
    n = L.cugo_debug_plan_array(1, name.encode(), C.byref(p))
    assert n >= 0, name
    return np.ctypeslib.as_array(p, shape=(n,)).copy() if n else np.zeros(0, np.int32)


def layout():
    """per front: offset and leading dimension of its storage (chol_symbolic.cpp, "5b storage"), W offset, level"""
    ncb, nb, alias = plan("ncb").astype(np.int64), plan("nb").astype(np.int64), plan("alias_of")
    ns = len(ncb)
    off, ld, woff, o, w = np.zeros(ns, np.int64), np.zeros(ns, np.int64), np.zeros(ns, np.int64), 0, 0
    for k in range(ns):
        c = alias[k]
        if c >= 0:
            ld[k] = ld[c]; off[k] = off[c] + 6 * ncb[c] * ld[c] + 6 * ncb[c]
        else:
            ld[k] = 6 * nb[k] + 1; off[k] = o; o += ld[k] * 6 * nb[k]
        woff[k] = w; w += ((6 * ncb[k] + 15) // 16 * 16) ** 2
    stp, tp, tf = plan("stage_task_ptr"), plan("task_ptr"), plan("task_fronts")
    stage = np.zeros(ns, int)
    for st in range(len(stp) - 1):
        for t in range(stp[st], stp[st + 1]):
            for fi in range(tp[t], tp[t + 1]):
                stage[tf[fi]] = st
    return ncb, nb, alias, off, ld, woff, stage


def optimise():
    g = cugo.graph_from_arrays(d)
    g.initialize(); g.optimize(10)
    return g, tuple(s["chi2"] for s in g.stats())


gref, ref = optimise()
cugo.check(L.cugo_debug_pin_reference())
print("reference", ref[-1], flush=True)
found, c = 0, 0
for c in range(1, N):
    if INJECT and c == 2:
        os.environ["CUGO_DEBUG_SKIP"] = INJECT
    g, chi = optimise()
    os.environ.pop("CUGO_DEBUG_SKIP", None)
    if chi != ref:
        found += 1
        it = next(i for i, (a, b) in enumerate(zip(chi, ref)) if a != b)
        print("run %d deviates from iteration %d on: chi2 %r instead of %r" % (c, it, chi[it], ref[it]), flush=True)
        ncb, nb, alias, off, ld, woff, stage = layout()
        A, B = slot(0, it), slot(1, it)
        P = slot(1, it - 1) if it > 0 else None
        for k, name in enumerate(NAMES):
            neq = np.flatnonzero(A[k] != B[k])
            line = "  %-22s %10d entries, %9d differ" % (name, len(A[k]), len(neq))
            if len(neq) and P is not None:
                line += "; %d of them hold the value of iteration %d (the factorisation before)" % (
                    int(np.count_nonzero(A[k][neq] == P[k][neq])), it - 1)
            print(line, flush=True)
            if not len(neq) or k > 1:
                continue
            rows = []
            for f_ in range(len(off)):
                if k == 0 and alias[f_] >= 0:
                    continue  # (stored inside its child's update block: counted with the child)
                lo = off[f_] if k == 0 else woff[f_]
                size = ld[f_] * 6 * nb[f_] if k == 0 else ((6 * ncb[f_] + 15) // 16 * 16) ** 2
                a, b = np.searchsorted(neq, lo), np.searchsorted(neq, lo + size)
                if b > a:
                    ldk = ld[f_] if k == 0 else (6 * ncb[f_] + 15) // 16 * 16
                    col, row = (neq[a:b] - lo) // ldk, (neq[a:b] - lo) % ldk
                    rows.append((int(stage[f_]), f_, int(b - a), int(col.min()), int(col.max()), int(row.min()), int(row.max()),
                                 int(6 * ncb[f_]), int(6 * nb[f_]), a))
            rows.sort()
            for r in rows[:10]:
                print("      level %2d front %4d: %8d entries differ, columns %d..%d rows %d..%d   (pivot columns %d, all %d)" % r[:9])
            f0, n0 = rows[0][1], rows[0][2]
            lo = off[f0] if k == 0 else woff[f0]
            ldk = ld[f0] if k == 0 else (6 * ncb[f0] + 15) // 16 * 16
            for i in neq[rows[0][9]:rows[0][9] + min(n0, 24)]:
                print("        front %d column %d row %d: %r instead of %r%s" % (f0, (i - lo) // ldk, (i - lo) % ldk, float(A[k][i]), float(B[k][i]),
                      ("   (iteration %d: %r)" % (it - 1, float(P[k][i]))) if P is not None else ""))
        del A, B, P
    g.close()
    if c % 50 == 49:
        print("  ... %d runs, %d deviating" % (c + 1, found), flush=True)
    if found >= 4:
        break
print("runs", c + 1, "deviating", found)
gref.close()
