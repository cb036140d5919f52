"""GPU box: the factorisation called back to back with CHANGING inputs.  tools/repro_stage.py calls every stage
with the same inputs over and over, so a value left over from the call before is the right value and a read
that comes too early cannot show; here the damping alternates between three values from call to call (every
intermediate array — fronts, W, L21, y — differs from its predecessor), K calls are queued without a host
wait in between, and each call's x_p goes to its own buffer and is compared bit for bit with the reference
of its damping value.  Several solvers (environment variants, read when a solver is created) take turns.
    python tools/repro_chain.py ROUNDS [--schur] [VAR=val,VAR=val ...]
--schur: the Schur complement is rebuilt (compute_schur for the call's damping value) before every call"""
import ctypes as C, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
cugo = importlib.import_module("cuda-bundle-adjustment_amd")
import numpy as np
import devmem, oracle
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
with_schur = "--schur" in sys.argv
args = [a for a in sys.argv[2:] if not a.startswith("--")]
configs = [dict(kv.split("=") for kv in a.split(",") if kv) for a in args] or [{}]
K = 12
LAMS = (0.37, 0.0411, 3.3)
L = cugo.lib()
d = cugo.synth(400, 8000, 33000, seed=11, n_loop_closures=200)
prob = oracle.Problem(d["pose"], d["pose_fixed"], d["lm"], d["lm_fixed"], d["e_pose"], d["e_lm"], d["e_stereo"],
                      d["e_meas"], d["e_omega"], d["e_cam"])
ctx = devmem.Ctx()
f = devmem.flatten(prob)
ev = devmem.upload_edges(ctx, f)
P, Lf, E = f["P"], f["L"], f["E"]
RK0 = cugo.Robust(0, 1.0, 0, 1.0)
d_poses, d_lms = ctx.to_dev(f["poses"]), ctx.to_dev(f["lms"])
b = dict(Hpp=ctx.empty(36 * P), bp=ctx.empty(6 * P), Hll=ctx.empty(9 * Lf), bl=ctx.empty(3 * Lf), Hpl=ctx.empty(18 * E),
         chi=ctx.empty(4))
cugo.check(L.cugo_construct_quadratic_form(ctx.h, C.byref(ev), d_poses, d_lms, RK0, b["Hpp"], b["bp"], b["Hll"],
                                           b["bl"], b["Hpl"], b["chi"]))
rowptr, colind, off_ptr, ei, ej = devmem.hsc_structure(f)
B = len(colind)
hs = cugo.HscStruct(B, ctx.to_dev(rowptr), ctx.to_dev(colind), ctx.to_dev(off_ptr), ctx.to_dev(ei), ctx.to_dev(ej))
inv, T = ctx.empty(9 * Lf), ctx.empty(18 * E)
# one (bsc, Hsc) per damping value when the Schur complement is not rebuilt inside the chain
sc = [(ctx.empty(6 * P), ctx.empty(36 * B)) for _ in LAMS]


def schur(q, bsc, Hsc):
    cugo.check(L.cugo_compute_schur(ctx.h, C.byref(ev), C.byref(hs), C.c_double(LAMS[q]), 0, b["Hpp"], b["bp"], b["Hll"],
                                    b["bl"], b["Hpl"], inv, T, bsc, Hsc))


for q in range(len(LAMS)):
    schur(q, *sc[q])
ctx.sync()
work = (ctx.empty(6 * P), ctx.empty(36 * B))  # (--schur) rebuilt before every call
solvers = []
for cfg in configs:
    for k, v in cfg.items():
        os.environ[k] = v
    s = C.c_void_p()
    cugo.check(L.cugo_chol_create(ctx.h, C.byref(s)))
    cugo.check(L.cugo_chol_analyze(s, P, rowptr.ctypes.data_as(C.POINTER(C.c_int32)), colind.ctypes.data_as(C.POINTER(C.c_int32))))
    xs = [ctx.empty(6 * P) for _ in range(K)]
    fail = ctx.empty(2, np.int32)
    ref = []
    for q in range(len(LAMS)):
        got = []
        for _ in range(3):
            cugo.check(L.cugo_chol_factor_solve(s, sc[q][1], C.c_double(LAMS[q]), sc[q][0], xs[0], fail))
            got.append(ctx.to_host(xs[0], 6 * P))
        assert all(np.array_equal(got[0], g) for g in got), "reference calls differ"
        ref.append(got[0])
    solvers.append((cfg, s, xs, fail, ref))
    for k in cfg:
        os.environ.pop(k, None)
bad = [[] for _ in configs]
for r in range(R):
    for ci, (cfg, s, xs, fail, ref) in enumerate(solvers):
        for k in range(K):
            q = (k + r) % len(LAMS)
            if with_schur:
                schur(q, *work)
                cugo.check(L.cugo_chol_factor_solve(s, work[1], C.c_double(LAMS[q]), work[0], xs[k], fail))
            else:
                cugo.check(L.cugo_chol_factor_solve(s, sc[q][1], C.c_double(LAMS[q]), sc[q][0], xs[k], fail))
        ctx.sync()
        for k in range(K):
            q = (k + r) % len(LAMS)
            x = ctx.to_host(xs[k], 6 * P)
            if not np.array_equal(x, ref[q]):
                nd = int(np.count_nonzero(x != ref[q]))
                rel = float(np.max(np.abs(x - ref[q])) / np.max(np.abs(ref[q])))
                bad[ci].append((r, k, nd, rel))
                if len(bad[ci]) <= 6:
                    print("  ", cfg, "round", r, "call", k, "damping", LAMS[q], "entries that differ", nd, "of", 6 * P,
                          "max abs diff / max abs", rel, flush=True)
for ci, cfg in enumerate(configs):
    print("chain%s" % (" + schur" if with_schur else ""), cfg, "calls", R * K, "deviating", len(bad[ci]), bad[ci][:4], flush=True)
