"""GPU box: which stage of an LM iteration is it that — about once in 9 000 iterations on the medium graph
(400 poses / 8 000 landmarks / 33 000 edges) — gives another result from the same inputs?  Every stage is
called N times through the C ABI on fixed device inputs and its outputs are compared bit for bit with those
of the first call.
    python tools/repro_stage.py [N_chol] [N_other]"""
import ctypes as C, hashlib, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
cugo = importlib.import_module("cuda-bundle-adjustment_amd")
import numpy as np
import devmem, oracle
Nc = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
No = int(sys.argv[2]) if len(sys.argv) > 2 else 8000
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


def digest(arrs):
    h = hashlib.blake2b(digest_size=16)
    for a in arrs:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.digest()


def stage(name, n, call, outs):
    ref, bad = None, []
    for i in range(n):
        call()
        dg = [digest([ctx.to_host(p, k)]) for p, k in outs]
        if ref is None:
            ref = dg
        elif dg != ref:
            bad.append((i, [j for j, (x, y) in enumerate(zip(dg, ref)) if x != y]))
    print("%-22s calls %6d  deviating %d %s" % (name, n, len(bad), bad[:6]), flush=True)


def build():
    cugo.check(L.cugo_construct_quadratic_form(ctx.h, C.byref(ev), d_poses, d_lms, RK0, b["Hpp"], b["bp"], b["Hll"],
                                               b["bl"], b["Hpl"], b["chi"]))


stage("build", No, build, [(b["Hpp"], 36 * P), (b["bp"], 6 * P), (b["Hll"], 9 * Lf), (b["bl"], 3 * Lf), (b["Hpl"], 18 * E),
                           (b["chi"], 1)])
rowptr, colind, off_ptr, ei, ej = devmem.hsc_structure(f)
B = len(colind)
hs = cugo.HscStruct(B, ctx.to_dev(rowptr), ctx.to_dev(colind), ctx.to_dev(off_ptr), ctx.to_dev(ei), ctx.to_dev(ej))
lam = 0.37
inv, T, bsc, Hsc = ctx.empty(9 * Lf), ctx.empty(18 * E), ctx.empty(6 * P), ctx.empty(36 * B)


def schur():
    cugo.check(L.cugo_compute_schur(ctx.h, C.byref(ev), C.byref(hs), C.c_double(lam), 0, b["Hpp"], b["bp"], b["Hll"],
                                    b["bl"], b["Hpl"], inv, T, bsc, Hsc))


stage("schur", No, schur, [(inv, 9 * Lf), (T, 18 * E), (bsc, 6 * P), (Hsc, 36 * B)])
s = C.c_void_p()
cugo.check(L.cugo_chol_create(ctx.h, C.byref(s)))
cugo.check(L.cugo_chol_analyze(s, P, rowptr.ctypes.data_as(C.POINTER(C.c_int32)), colind.ctypes.data_as(C.POINTER(C.c_int32))))
xp, fail = ctx.empty(6 * P), ctx.empty(2, np.int32)


def chol():
    cugo.check(L.cugo_chol_factor_solve(s, Hsc, C.c_double(lam), bsc, xp, fail))


stage("cholesky", Nc, chol, [(xp, 6 * P)])
xl, scale = ctx.empty(3 * Lf), ctx.empty(2)
po, lo = ctx.to_dev(f["poses"]), ctx.to_dev(f["lms"])


def backsubst():
    cugo.check(L.cugo_backsubst_update(ctx.h, C.byref(ev), C.c_double(lam), inv, b["bl"], b["bp"], b["Hpl"], xp, xl,
                                       d_poses, d_lms, po, lo, scale))


stage("backsubst_update", No, backsubst, [(xl, 3 * Lf), (po, int(np.size(f["poses"]))), (lo, int(np.size(f["lms"]))), (scale, 1)])
chi2 = ctx.empty(2)
stage("errors", No, lambda: cugo.check(L.cugo_compute_active_errors(ctx.h, C.byref(ev), po, lo, RK0, chi2)), [(chi2, 1)])
