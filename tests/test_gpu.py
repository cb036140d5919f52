"""GPU parity tests: every call goes through the C ABI of libcugo_hip.so and is checked
against the CPU oracle (oracle/ba_oracle.c) and the committed numpy goldens.

Tolerances: chi2 per iteration 1e-10 relative (north star), estimates 1e-9 absolute; the stress
fixture reject_8x60 uses max(1e-10, 4 x the oracle's own measured order sensitivity) per iteration
(conftest.golden_tolerances).  Kernel-level block outputs: 1e-11 relative to the block scale.
"""
import ctypes as C
import importlib

import numpy as np
import pytest

import synth
from conftest import GOLDEN_GRAPHS, PROBLEM_KEYS, golden_path, golden_tolerances

pytestmark = pytest.mark.gpu

cugo = importlib.import_module("cuda-bundle-adjustment_amd")


@pytest.fixture(scope="module")
def ctx():
    import devmem
    if cugo.device_count() == 0:
        pytest.fail("no HIP device: the GPU tests must run on the MI355X box")
    c = devmem.Ctx()
    yield c
    c.close()


def dptr(p):
    return C.c_void_p(p.value if isinstance(p, C.c_void_p) else p)


def mixed_problem(oracle, seed=4, **kw):
    args = dict(n_poses=14, n_landmarks=260, mean_obs=3.6, seed=seed, fixed_poses=(0, 5),
                fixed_landmarks=(3, 77, 200), loop_closure=True, per_edge_cam=True)
    args.update(kw)
    d = synth.make_problem(**args)
    return oracle.Problem(*synth.problem_fields(d))


RK0 = cugo.Robust(0, 1.0, 0, 1.0)


# ------------------------------------------------------------------ kernel level --------
def build_on_gpu(ctx, f, ev, rk=RK0):
    L = cugo.lib()
    P, Lf, E = f["P"], f["L"], f["E"]
    d_poses, d_lms = ctx.to_dev(f["poses"]), ctx.to_dev(f["lms"])
    d = dict(poses=d_poses, lms=d_lms, Hpp=ctx.empty(36 * P), bp=ctx.empty(6 * P), Hll=ctx.empty(9 * Lf),
             bl=ctx.empty(3 * Lf), Hpl=ctx.empty(18 * E), chi=ctx.empty(4))
    cugo.check(L.cugo_construct_quadratic_form(ctx.h, C.byref(ev), d_poses, d_lms, rk, d["Hpp"], d["bp"],
                                               d["Hll"], d["bl"], d["Hpl"], d["chi"]))
    return d


@pytest.mark.parametrize("rk", [(0, 1.0), (1, 2.5), (2, 6.0), (3, 1.5)])
def test_errors_and_quadratic_form_vs_oracle(ctx, oracle_lib, rk):
    import devmem
    prob = mixed_problem(oracle_lib)
    prob.rk_type, prob.rk_delta = rk
    f = devmem.flatten(prob)
    ev = devmem.upload_edges(ctx, f)
    rkc = cugo.Robust(rk[0], rk[1], rk[0], rk[1])
    d = build_on_gpu(ctx, f, ev, rkc)
    ref = prob.build_system()
    P, Lf, E = f["P"], f["L"], f["E"]
    chi = ctx.to_host(d["chi"], 1)[0]
    assert abs(chi - ref["chi"]) <= 1e-13 * ref["chi"]
    for name, shape in (("Hpp", (P, 36)), ("bp", (P, 6)), ("Hll", (Lf, 9)), ("bl", (Lf, 3))):
        got = ctx.to_host(d[name], shape)
        scale = np.abs(ref[name]).max()
        np.testing.assert_allclose(got, ref[name], rtol=0, atol=1e-12 * scale, err_msg=name)
    Hpl = ctx.to_host(d["Hpl"], (E, 18))
    np.testing.assert_allclose(Hpl, ref["Hpl"][f["src"]], rtol=0, atol=1e-12 * np.abs(ref["Hpl"]).max())
    # error-only pass gives the same chi2
    chi2 = ctx.empty(2)
    cugo.check(cugo.lib().cugo_compute_active_errors(ctx.h, C.byref(ev), d["poses"], d["lms"], rkc, chi2))
    assert abs(ctx.to_host(chi2, 1)[0] - ref["chi"]) <= 1e-13 * ref["chi"]
    # max diagonal
    md = ctx.empty(2)
    cugo.check(cugo.lib().cugo_max_diagonal(ctx.h, d["Hpp"], P, d["Hll"], Lf, md))
    want = max(0.0, ref["Hpp"].reshape(P, 6, 6)[:, range(6), range(6)].max(),
               ref["Hll"].reshape(Lf, 3, 3)[:, range(3), range(3)].max())
    assert ctx.to_host(md, 1)[0] == pytest.approx(want, rel=1e-13)


def test_schur_complement_and_backsubst_vs_oracle(ctx, oracle_lib):
    import devmem
    prob = mixed_problem(oracle_lib, seed=8)
    f = devmem.flatten(prob)
    ev = devmem.upload_edges(ctx, f)
    d = build_on_gpu(ctx, f, ev)
    P, Lf, E = f["P"], f["L"], f["E"]
    rowptr, colind, off_ptr, ei, ej = devmem.hsc_structure(f)
    B = len(colind)
    hs = cugo.HscStruct(B, ctx.to_dev(rowptr), ctx.to_dev(colind), ctx.to_dev(off_ptr), ctx.to_dev(ei),
                        ctx.to_dev(ej))
    lam = 3.7
    inv, T, bsc, Hsc = ctx.empty(9 * Lf), ctx.empty(18 * E), ctx.empty(6 * P), ctx.empty(36 * B)
    cugo.check(cugo.lib().cugo_compute_schur(ctx.h, C.byref(ev), C.byref(hs), C.c_double(lam), 1, d["Hpp"],
                                             d["bp"], d["Hll"], d["bl"], d["Hpl"], inv, T, bsc, Hsc))
    Href, bref = prob.schur_dense(lam)
    H = ctx.to_host(Hsc, (B, 36))
    dense = np.zeros_like(Href)
    for r in range(P):
        for k in range(rowptr[r], rowptr[r + 1]):
            c = colind[k]
            blk = H[k].reshape(6, 6).T
            dense[6 * r:6 * r + 6, 6 * c:6 * c + 6] = blk
            if c != r:
                dense[6 * c:6 * c + 6, 6 * r:6 * r + 6] = blk.T
    np.testing.assert_allclose(dense, Href, rtol=0, atol=1e-11 * np.abs(Href).max())
    np.testing.assert_allclose(ctx.to_host(bsc, 6 * P), bref, rtol=0, atol=1e-11 * np.abs(bref).max())

    # sparse LL^T through the ABI on this very system (undamped Hsc + lambda inside the solver)
    cugo.check(cugo.lib().cugo_compute_schur(ctx.h, C.byref(ev), C.byref(hs), C.c_double(lam), 0, d["Hpp"],
                                             d["bp"], d["Hll"], d["bl"], d["Hpl"], inv, T, bsc, Hsc))
    s = C.c_void_p()
    cugo.check(cugo.lib().cugo_chol_create(ctx.h, C.byref(s)))
    cugo.check(cugo.lib().cugo_chol_analyze(s, P, rowptr.ctypes.data_as(C.POINTER(C.c_int32)),
                                            colind.ctypes.data_as(C.POINTER(C.c_int32))))
    xp, fail = ctx.empty(6 * P), ctx.empty(2, np.int32)
    cugo.check(cugo.lib().cugo_chol_factor_solve(s, Hsc, C.c_double(lam), bsc, xp, fail))
    assert ctx.to_host(fail, 1, np.int32)[0] == 0
    ok, dxp, dxl = prob.solve_step(lam, dense=True)
    assert ok
    got_xp = ctx.to_host(xp, (P, 6))
    np.testing.assert_allclose(got_xp, dxp, rtol=1e-9, atol=1e-12 * np.abs(dxp).max())

    # back-substitution + update + scale
    xl, scale = ctx.empty(3 * Lf), ctx.empty(2)
    po, lo = ctx.to_dev(f["poses"]), ctx.to_dev(f["lms"])
    cugo.check(cugo.lib().cugo_backsubst_update(ctx.h, C.byref(ev), C.c_double(lam), inv, d["bl"], d["bp"],
                                                d["Hpl"], xp, xl, d["poses"], d["lms"], po, lo, scale))
    got_xl = ctx.to_host(xl, (Lf, 3))
    np.testing.assert_allclose(got_xl, dxl, rtol=1e-8, atol=1e-11 * np.abs(dxl).max())
    ref = prob.build_system()
    want_scale = (got_xp * (lam * got_xp + ref["bp"])).sum() + (got_xl * (lam * got_xl + ref["bl"])).sum()
    assert ctx.to_host(scale, 1)[0] == pytest.approx(want_scale, rel=1e-11)
    new_poses = ctx.to_host(po, (f["Pall"], 7))
    for i in range(prob.n_poses):
        idx = f["pidx"][i]
        if prob.pose_fixed[i]:
            assert np.array_equal(new_poses[idx], f["poses"][idx])
        else:
            np.testing.assert_allclose(new_poses[idx], oracle_lib.pose_update(prob.pose[i], got_xp[idx]),
                                       rtol=0, atol=1e-14)
    new_lms = ctx.to_host(lo, (f["Lall"], 3))
    np.testing.assert_allclose(new_lms[:Lf], f["lms"][:Lf] + got_xl, rtol=0, atol=1e-14)
    assert np.array_equal(new_lms[Lf:], f["lms"][Lf:])
    cugo.lib().cugo_chol_destroy(s)


@pytest.mark.parametrize("f32", [False, True])
def test_schur_landmark_major_plan_vs_gather_and_oracle(ctx, oracle_lib, f32):
    """cugo_hsc_plan_create + cugo_compute_schur: the landmark-major form of the Schur complement
    (products formed from LDS inside the edge pass, per-group partial slots, ordered reduction)
    against the destination-major gather kernels on the same padded layout and against the
    oracle's dense Schur complement; also the optional T output and the refusal of layouts in
    which a landmark straddles two 256-slot groups."""
    import devmem
    prob = mixed_problem(oracle_lib, seed=8, n_poses=40, n_landmarks=900, mean_obs=4.2)
    f0 = devmem.flatten(prob)
    L = cugo.lib()
    # un-padded layout: some landmark straddles a group boundary -> the plan is refused
    rowptr, colind, off_ptr, ei, ej = devmem.hsc_structure(f0)
    hs0 = cugo.HscStruct()
    plan0 = C.c_void_p()
    as_p = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    rc = L.cugo_hsc_plan_create(ctx.h, f0["E"], f0["P"], as_p(f0["pose"], C.c_int32), as_p(f0["lm"], C.c_int32),
                                as_p(f0["flags"], C.c_uint8), as_p(rowptr, C.c_int32), as_p(colind, C.c_int32),
                                C.byref(hs0), C.byref(plan0))
    assert rc == -3 and not plan0 and not hs0.d_grp_ptr   # CUGO_ERR_INVALID
    f = devmem.pad_to_groups(f0)
    assert f["E"] > f0["E"]
    ev = devmem.upload_edges(ctx, f)
    ev.block_f32 = int(f32)
    rowptr, colind, off_ptr, ei, ej = devmem.hsc_structure(f)
    B, P, Lf, E = len(colind), f["P"], f["L"], f["E"]
    blk = np.float32 if f32 else np.float64
    d = dict(poses=ctx.to_dev(f["poses"]), lms=ctx.to_dev(f["lms"]), Hpp=ctx.empty(36 * P), bp=ctx.empty(6 * P),
             Hll=ctx.empty(9 * Lf), bl=ctx.empty(3 * Lf), Hpl=ctx.empty(18 * E, blk), chi=ctx.empty(4))
    cugo.check(L.cugo_construct_quadratic_form(ctx.h, C.byref(ev), d["poses"], d["lms"], RK0, d["Hpp"], d["bp"],
                                               d["Hll"], d["bl"], d["Hpl"], d["chi"]))
    lam = 2.25
    out = {}
    for mode in ("gather", "plan"):
        hs = cugo.HscStruct(B, ctx.to_dev(rowptr), ctx.to_dev(colind), ctx.to_dev(off_ptr), ctx.to_dev(ei),
                            ctx.to_dev(ej))
        plan = C.c_void_p()
        if mode == "plan":
            cugo.check(L.cugo_hsc_plan_create(ctx.h, E, P, as_p(f["pose"], C.c_int32), as_p(f["lm"], C.c_int32),
                                              as_p(f["flags"], C.c_uint8), as_p(rowptr, C.c_int32),
                                              as_p(colind, C.c_int32), C.byref(hs), C.byref(plan)))
            assert hs.d_grp_ptr and hs.n_groups == (E + 255) // 256 and hs.n_slots >= hs.n_rhs > 0
        inv, T, bsc, Hsc = ctx.empty(9 * Lf), ctx.empty(18 * E, blk), ctx.empty(6 * P), ctx.empty(36 * B)
        cugo.check(L.cugo_compute_schur(ctx.h, C.byref(ev), C.byref(hs), C.c_double(lam), 1, d["Hpp"], d["bp"],
                                        d["Hll"], d["bl"], d["Hpl"], inv, T, bsc, Hsc))
        out[mode] = dict(H=ctx.to_host(Hsc, (B, 36)), b=ctx.to_host(bsc, 6 * P), T=ctx.to_host(T, (E, 18), blk),
                         inv=ctx.to_host(inv, (Lf, 9)))
        if mode == "plan":
            # without a T array (what the engine does): same Hsc / bsc
            Hsc2, bsc2 = ctx.empty(36 * B), ctx.empty(6 * P)
            cugo.check(L.cugo_compute_schur(ctx.h, C.byref(ev), C.byref(hs), C.c_double(lam), 1, d["Hpp"], d["bp"],
                                            d["Hll"], d["bl"], d["Hpl"], inv, None, bsc2, Hsc2))
            assert np.array_equal(ctx.to_host(Hsc2, (B, 36)), out["plan"]["H"])
            assert np.array_equal(ctx.to_host(bsc2, 6 * P), out["plan"]["b"])
            L.cugo_hsc_plan_destroy(plan)
    scale = np.abs(out["gather"]["H"]).max()
    # float block storage: the gather kernels read T rounded to float, the plan keeps it in fp64
    tol = 2e-6 if f32 else 1e-13
    np.testing.assert_allclose(out["plan"]["H"], out["gather"]["H"], rtol=0, atol=tol * scale)
    np.testing.assert_allclose(out["plan"]["b"], out["gather"]["b"], rtol=0, atol=tol * np.abs(out["gather"]["b"]).max())
    assert np.array_equal(out["plan"]["T"], out["gather"]["T"]) and np.array_equal(out["plan"]["inv"], out["gather"]["inv"])
    if not f32:
        Href, bref = prob.schur_dense(lam)
        dense = np.zeros_like(Href)
        for r in range(P):
            for k in range(rowptr[r], rowptr[r + 1]):
                c = colind[k]
                b6 = out["plan"]["H"][k].reshape(6, 6).T
                dense[6 * r:6 * r + 6, 6 * c:6 * c + 6] = b6
                if c != r:
                    dense[6 * c:6 * c + 6, 6 * r:6 * r + 6] = b6.T
        np.testing.assert_allclose(dense, Href, rtol=0, atol=1e-11 * np.abs(Href).max())
        np.testing.assert_allclose(out["plan"]["b"], bref, rtol=0, atol=1e-11 * np.abs(bref).max())


def test_float32_block_kernels_vs_oracle(ctx, oracle_lib):
    """cugo_edges.block_f32 = 1 (fp32-internal mode, BASELINE config 5): Hpl and T = Hpl invHll are
    float arrays, everything else stays fp64.  Stated tolerance at kernel level: the stored blocks
    are the fp64 values rounded to float (<= 2^-24 relative), Hsc and bsc agree with the fp64
    oracle to 2e-6 of their largest entry, the landmark step to 5e-5 of its largest entry."""
    import devmem
    prob = mixed_problem(oracle_lib, seed=8)
    f = devmem.flatten(prob)
    ev = devmem.upload_edges(ctx, f)
    ev.block_f32 = 1
    d = build_on_gpu(ctx, f, ev)
    P, Lf, E = f["P"], f["L"], f["E"]
    ref = prob.build_system()
    want = ref["Hpl"][f["src"]]
    Hpl = ctx.to_host(d["Hpl"], (E, 18), np.float32)
    assert Hpl.dtype == np.float32 and np.all(np.abs(Hpl - want) <= 6.1e-8 * np.abs(want) + 1e-12 * np.abs(want).max())
    for name, shape in (("Hpp", (P, 36)), ("bp", (P, 6)), ("Hll", (Lf, 9)), ("bl", (Lf, 3))):  # untouched: fp64
        got = ctx.to_host(d[name], shape)
        np.testing.assert_allclose(got, ref[name], rtol=0, atol=1e-12 * np.abs(ref[name]).max(), err_msg=name)
    rowptr, colind, off_ptr, ei, ej = devmem.hsc_structure(f)
    B = len(colind)
    hs = cugo.HscStruct(B, ctx.to_dev(rowptr), ctx.to_dev(colind), ctx.to_dev(off_ptr), ctx.to_dev(ei),
                        ctx.to_dev(ej))
    lam = 3.7
    inv, T, bsc, Hsc = ctx.empty(9 * Lf), ctx.empty(18 * E), ctx.empty(6 * P), ctx.empty(36 * B)
    cugo.check(cugo.lib().cugo_compute_schur(ctx.h, C.byref(ev), C.byref(hs), C.c_double(lam), 1, d["Hpp"],
                                             d["bp"], d["Hll"], d["bl"], d["Hpl"], inv, T, bsc, Hsc))
    Href, bref = prob.schur_dense(lam)
    H = ctx.to_host(Hsc, (B, 36))
    dense = np.zeros_like(Href)
    for r in range(P):
        for k in range(rowptr[r], rowptr[r + 1]):
            c = colind[k]
            blk = H[k].reshape(6, 6).T
            dense[6 * r:6 * r + 6, 6 * c:6 * c + 6] = blk
            if c != r:
                dense[6 * c:6 * c + 6, 6 * r:6 * r + 6] = blk.T
    err_H = np.abs(dense - Href).max() / np.abs(Href).max()
    err_b = np.abs(ctx.to_host(bsc, 6 * P) - bref).max() / np.abs(bref).max()
    assert 1e-12 < err_H < 2e-6 and err_b < 2e-6, (err_H, err_b)  # float storage is really in use
    ok, dxp, dxl = prob.solve_step(lam, dense=True)
    assert ok
    xp = ctx.to_dev(np.ascontiguousarray(dxp))
    xl, scale = ctx.empty(3 * Lf), ctx.empty(2)
    po, lo = ctx.to_dev(f["poses"]), ctx.to_dev(f["lms"])
    cugo.check(cugo.lib().cugo_backsubst_update(ctx.h, C.byref(ev), C.c_double(lam), inv, d["bl"], d["bp"],
                                                d["Hpl"], xp, xl, d["poses"], d["lms"], po, lo, scale))
    got_xl = ctx.to_host(xl, (Lf, 3))
    # bl - sum Hpl^T xp cancels: the landmark step keeps ~5 digits with float Hpl
    assert np.abs(got_xl - dxl).max() < 5e-5 * np.abs(dxl).max()


# CUGO_MIN_SUBTREE_TASKS=0 forces the subtree stage (k_subtree_factor) that small graphs skip
# CUGO_ALIAS_CHAINS=0 turns off the storage sharing of single-child chains (every front then gets
# its extend-add)
@pytest.mark.parametrize("env", [{}, {"CUGO_MIN_SUBTREE_TASKS": "0"}, {"CUGO_ALIAS_CHAINS": "0"},
                                 {"CUGO_ND_LEAF": "4", "CUGO_MAX_SUPER_COLS": "3", "CUGO_TARGET_TASKS": "4",
                                  "CUGO_MIN_SUBTREE_TASKS": "0"},
                                 {"CUGO_ND_LEAF": "1000", "CUGO_MAX_SUPER_COLS": "1", "CUGO_TARGET_TASKS": "100000"},
                                 {"CUGO_MAX_SUPER_COLS": "24", "CUGO_ZERO_FRAC": "0.9", "CUGO_MIN_SUBTREE_TASKS": "0"},
                                 # every level cut into 64x64 tiles (small problems otherwise take the 32x32 form)
                                 {"CUGO_TILE32_MAX_TILES": "0"},
                                 {"CUGO_TILE32_MAX_TILES": "0", "CUGO_ALIAS_CHAINS": "0", "CUGO_MAX_SUPER_COLS": "5"},
                                 # the look-ahead schedule (opt-in): a level's update tiles run with the next
                                 # level's potrf, only the lead block of each front is computed in between
                                 {"CUGO_LOOKAHEAD": "1"},
                                 {"CUGO_LOOKAHEAD": "1", "CUGO_TILE32_MAX_TILES": "0", "CUGO_ALIAS_CHAINS": "0"},
                                 {"CUGO_LOOKAHEAD": "1", "CUGO_MIN_SUBTREE_TASKS": "0", "CUGO_MAX_SUPER_COLS": "5"},
                                 # every level in the two-phase form (row tiles solved once, then syrk-only tiles)
                                 # that levels with more tiles than CUs take
                                 {"CUGO_TWO_PHASE_MIN_TILES": "1", "CUGO_TILE32_MAX_TILES": "0"},
                                 {"CUGO_TWO_PHASE_MIN_TILES": "1", "CUGO_TILE32_MAX_TILES": "0", "CUGO_ALIAS_CHAINS": "0",
                                  "CUGO_MAX_SUPER_COLS": "5"},
                                 # the 6-column LDS panels of rounds 1-2 (the default is the 16-column register panel)
                                 {"CUGO_PANEL16": "0"},
                                 {"CUGO_PANEL16": "0", "CUGO_MIN_SUBTREE_TASKS": "0", "CUGO_MAX_SUPER_COLS": "5"},
                                 # clear + scatter (two launches) instead of the assembly that writes every entry
                                 {"CUGO_ASM_FRONTS": "0"},
                                 {"CUGO_ASM_FRONTS": "0", "CUGO_ALIAS_CHAINS": "0", "CUGO_MAX_SUPER_COLS": "5"}])
def test_sparse_cholesky_vs_numpy(ctx, env, monkeypatch):
    from test_host import covis_pattern, patterns, random_spd_bsr
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    rng = np.random.default_rng(5)
    cases = dict(patterns())
    d = cugo.synth(160, 2500, 10500, seed=3, n_loop_closures=80)
    ep = d["e_pose"].astype(np.int64) - 1
    ep[ep < 0] = 10**6
    cases["synthetic"] = covis_pattern(159, ep, d["e_lm"])
    for name, pat in cases.items():
        if isinstance(pat, tuple):
            rowptr, colind = pat
        else:
            rowptr = np.array([0] + list(np.cumsum([len(r) for r in pat])), np.int32)
            colind = np.array([c for r in pat for c in r], np.int32)
        n = len(rowptr) - 1
        A, vals = random_spd_bsr(rowptr, colind, rng)
        s = C.c_void_p()
        cugo.check(cugo.lib().cugo_chol_create(ctx.h, C.byref(s)))
        cugo.check(cugo.lib().cugo_chol_analyze(s, n, rowptr.ctypes.data_as(C.POINTER(C.c_int32)),
                                                colind.ctypes.data_as(C.POINTER(C.c_int32))))
        b = rng.normal(size=6 * n)
        dH, db, dx, fail = ctx.to_dev(vals), ctx.to_dev(b), ctx.empty(6 * n), ctx.empty(2, np.int32)
        for lam in (0.0, 2.5):  # two solves with one analysis (one per LM trial)
            cugo.check(cugo.lib().cugo_chol_factor_solve(s, dH, C.c_double(lam), db, dx, fail))
            assert ctx.to_host(fail, 1, np.int32)[0] == 0, name
            x = ctx.to_host(dx, 6 * n)
            xref = np.linalg.solve(A + lam * np.eye(6 * n), b)
            np.testing.assert_allclose(x, xref, rtol=1e-9, atol=1e-12, err_msg=name)
        # an indefinite matrix raises the zero-pivot flag (ref: "factorize failed!" path)
        bad = vals.copy()
        bad[rowptr[n // 2]] = -np.eye(6).reshape(-1)
        dB = ctx.to_dev(bad)
        cugo.check(cugo.lib().cugo_chol_factor_solve(s, dB, C.c_double(0.0), db, dx, fail))
        assert ctx.to_host(fail, 1, np.int32)[0] == 1, name
        cugo.lib().cugo_chol_destroy(s)


# ------------------------------------------------------------------ graph level ---------
def run_graph(d, niter, rk=(0, 1.0), **kw):
    g = cugo.graph_from_arrays(d, rk=rk, **kw)
    g.initialize()
    g.optimize(niter)
    out = dict(stats=g.stats(), pose=g.poses(), lm=g.landmarks(), profile=g.time_profile(),
               sstats=g.structure_stats(), nedges=g.n_active_edges())
    g.close()
    return out


def assert_trajectories_match(got, ref, tol, check_trials=True):
    """tol: one relative chi2 tolerance, or one per iteration"""
    assert len(got) == len(ref)
    tols = list(tol) if isinstance(tol, (list, tuple)) else [tol] * len(ref)
    for a, b, tol in zip(got, ref, tols):
        assert abs(a["chi2"] - b["chi2"]) <= tol * max(abs(b["chi2"]), 1e-6), (a, b)
        if check_trials:
            assert a["trials"] == b["trials"]
            assert a["lam"] == pytest.approx(b["lam"], rel=max(100 * tol, 1e-8))


@pytest.mark.parametrize("name", GOLDEN_GRAPHS)
def test_lm_trajectory_vs_golden_and_oracle(oracle_lib, name):
    g = np.load(golden_path(name + ".npz"))
    d = {k: g[k] for k in PROBLEM_KEYS}
    rk = (int(g["rk_type"]), float(g["rk_delta"]))
    out = run_graph(d, 10, rk=rk)
    if name.startswith("zero_noise"):
        assert len(out["stats"]) <= 2 and all(s["chi2"] < 1e-12 for s in out["stats"])
        np.testing.assert_allclose(out["pose"], g["pose_out"], rtol=0, atol=1e-9)
        return
    tol, etol = golden_tolerances(name)   # 1e-10 / 1e-9; measured conditioning for the stress fixture
    tr = [dict(chi2=t[1], lam=t[2], trials=int(t[4])) for t in g["trace"]]
    assert_trajectories_match(out["stats"], tr, tol)
    np.testing.assert_allclose(out["pose"], g["pose_out"], rtol=0, atol=etol)
    np.testing.assert_allclose(out["lm"], g["lm_out"], rtol=0, atol=10 * etol)
    # and the C oracle on the same input
    P = oracle_lib.Problem(*[g[k] for k in PROBLEM_KEYS], rk_type=rk[0], rk_delta=rk[1])
    ref = P.optimize(10)
    assert_trajectories_match(out["stats"], ref, tol)


def synth_problem(oracle, P, L, E, seed, lc):
    d = cugo.synth(P, L, E, seed=seed, n_loop_closures=lc)
    prob = oracle.Problem(d["pose"], d["pose_fixed"], d["lm"], d["lm_fixed"], d["e_pose"], d["e_lm"],
                          d["e_stereo"], d["e_meas"], d["e_omega"], d["e_cam"])
    return d, prob


def rmse(a, b):
    return float(np.sqrt(np.mean((a - b) ** 2)))


@pytest.mark.parametrize("subtree_stage", [False, True, "lookahead", "two_phase", "hsc_rows", "hsc_strip", "potrf6"])
def test_medium_synthetic_vs_oracle(oracle_lib, subtree_stage, monkeypatch):
    if subtree_stage == "hsc_strip":  # the opt-in row-strip form of the off-diagonal gather (k_hsc_offdiag_strip)
        monkeypatch.setenv("CUGO_HSC_STRIP", "1")
    elif subtree_stage == "hsc_rows":  # the opt-in Schur complement by whole block rows (k_hsc_rows)
        monkeypatch.setenv("CUGO_HSC_ROWS", "1")
    elif subtree_stage == "potrf6":  # the 6-column LDS panels of rounds 1-2 instead of the register panels
        monkeypatch.setenv("CUGO_PANEL16", "0")
    elif subtree_stage == "lookahead":  # the opt-in Cholesky schedule, end to end
        monkeypatch.setenv("CUGO_LOOKAHEAD", "1")
    elif subtree_stage == "two_phase":  # every level in the form the widest levels take
        monkeypatch.setenv("CUGO_TWO_PHASE_MIN_TILES", "1")
        monkeypatch.setenv("CUGO_TILE32_MAX_TILES", "0")
    elif subtree_stage:
        monkeypatch.setenv("CUGO_MIN_SUBTREE_TASKS", "0")
    d, prob = synth_problem(oracle_lib, 400, 8000, 33000, seed=11, lc=200)
    out = run_graph(d, 10)
    ref = prob.optimize(10)
    assert_trajectories_match(out["stats"], ref, 1e-10)
    # README.md:175-178 style RMSE between CPU and GPU estimates
    assert rmse(out["pose"][:, :4], prob.pose[:, :4]) < 1e-11
    assert rmse(out["pose"][:, 4:], prob.pose[:, 4:]) < 1e-9
    assert rmse(out["lm"], prob.lm) < 1e-9
    assert out["nedges"] == 33000
    assert out["stats"][-1]["chi2"] < out["stats"][0]["chi2"]


def test_structure_build_beside_initialize_is_the_synchronous_one(oracle_lib, monkeypatch):
    """a changed co-visibility starts the Hsc pattern, the ordering and the symbolic factorisation on a
    helper thread inside initialize() (engine.cpp, Impl::pat_thread); CUGO_ASYNC_STRUCTURE=0 does all of it
    in the first optimize().  Same structure, bitwise the same run — also over a sequence of calls in which
    the topology changes, stays, and only a measurement changes (lists rebuilt, plan kept)"""
    d, _ = synth_problem(oracle_lib, 150, 2200, 9000, seed=17, lc=60)
    extra = cugo.synth(150, 2200, 9400, seed=18, n_loop_closures=60)
    have = set(zip(d["e_pose"].tolist(), d["e_lm"].tolist()))
    sel = [i for i in range(len(extra["e_pose"])) if (int(extra["e_pose"][i]), int(extra["e_lm"][i])) not in have][:300]
    runs = []
    for mode in ("1", "0"):
        monkeypatch.setenv("CUGO_ASYNC_STRUCTURE", mode)
        g = cugo.graph_from_arrays(d)
        out = []
        g.initialize(); g.optimize(3)
        out.append(([s["chi2"] for s in g.stats()], g.poses().copy(), g.structure_stats()))
        g.initialize(); g.optimize(2)                       # nothing changed: everything re-used
        out.append(([s["chi2"] for s in g.stats()], g.poses().copy(), g.structure_stats()))
        for dim in (2, 3):                                   # new edges: a new co-visibility
            k = [i for i in sel if int(extra["e_stereo"][i]) == (dim == 3)]
            g.add_edges(dim, extra["e_pose"][k], extra["e_lm"][k], extra["e_meas"][k][:, :dim], extra["e_omega"][k],
                        extra["e_cam"][k])
        g.initialize(); g.initialize(); g.optimize(3)        # (a second initialize() joins the first helper)
        out.append(([s["chi2"] for s in g.stats()], g.poses().copy(), g.structure_stats()))
        g.close()
        runs.append(out)
    for a, b in zip(*runs):
        assert a[0] == b[0] and np.array_equal(a[1], b[1]) and a[2] == b[2]
    assert runs[0][2][2]["hsc_blocks"] >= runs[0][0][2]["hsc_blocks"]


def test_initialize_again_with_more_landmarks_while_the_helper_may_still_run(oracle_lib):
    """initialize() starts a helper thread that reads the co-visibility lists, P and L; a second
    initialize() right behind it — on a graph that has GROWN (more landmarks, more edges: every one of those arrays is
    reallocated) — must wait for that helper before it touches them (join_pattern() is its first statement).  The
    result is that of a fresh optimiser on the grown graph, bit for bit; repeated, because the window is a race."""
    big = cugo.synth(120, 2600, 10800, seed=23, n_loop_closures=40)
    cam = np.asarray(big["e_cam"], np.float64).reshape(-1, 5)
    big["e_cam"] = cam if len(cam) > 1 else np.tile(cam, (len(big["e_pose"]), 1))
    keep = big["e_lm"] < 1500
    small = dict(big)
    for k in ("e_pose", "e_lm", "e_stereo", "e_meas", "e_omega", "e_cam"):
        small[k] = big[k][keep]
    small["lm"], small["lm_fixed"] = big["lm"][:1500], big["lm_fixed"][:1500]
    fresh = run_graph(big, 4)
    rest = ~keep
    for _ in range(5):
        g = cugo.graph_from_arrays(small)
        g.initialize()                                    # helper thread: pattern + symbolic for 1 500 landmarks
        g.add_landmarks(np.arange(1500, 2600, dtype=np.int32), big["lm"][1500:], big["lm_fixed"][1500:])
        for dim in (2, 3):
            k = np.flatnonzero(rest & (big["e_stereo"].astype(bool) == (dim == 3)))
            g.add_edges(dim, big["e_pose"][k], big["e_lm"][k], big["e_meas"][k][:, :dim], big["e_omega"][k], big["e_cam"][k])
        g.initialize()                                    # at once: L changed from 1 500 to 2 600
        g.optimize(4)
        st, pose = g.stats(), g.poses()
        g.close()
        assert [s["chi2"] for s in st] == [s["chi2"] for s in fresh["stats"]]
        assert np.array_equal(pose, fresh["pose"])


@pytest.mark.parametrize("shape", ["mixed_fixed", "medium", "dense_ring", "all_landmarks_fixed"])
def test_device_structure_build_equals_host_build(oracle_lib, shape, monkeypatch):
    """the Hsc pattern and the contribution lists built on the device (pairs per landmark, stable
    radix sort by pose pair, runs; csrc/host/structure_gpu.cpp — ref: findHschureMulBlockIndices +
    thrust::sort, .cu:1347-1378,1606-1634) are the host build's, entry for entry: the two runs are
    bitwise identical and report the same structure"""
    if shape == "mixed_fixed":
        d = synth.make_problem(n_poses=30, n_landmarks=400, mean_obs=5.0, seed=9, fixed_poses=(0, 7, 8),
                               fixed_landmarks=tuple(range(0, 400, 7)), loop_closure=True)
    elif shape == "medium":
        d = cugo.synth(400, 8000, 33000, seed=11, n_loop_closures=200)
    elif shape == "dense_ring":
        d = dense_ring_problem(300, 12, seed=3)  # 300 edges per landmark
    else:
        d = synth.make_problem(n_poses=9, n_landmarks=80, seed=2, fixed_landmarks=tuple(range(80)))
    dev = run_graph(d, 5)
    monkeypatch.setenv("CUGO_HOST_STRUCTURE", "1")
    host = run_graph(d, 5)
    assert dev["sstats"] == host["sstats"]
    assert [s["chi2"] for s in dev["stats"]] == [s["chi2"] for s in host["stats"]]
    assert np.array_equal(dev["pose"], host["pose"]) and np.array_equal(dev["lm"], host["lm"])


def test_landmark_major_schur_plan_end_to_end(oracle_lib, monkeypatch):
    """CUGO_SCHUR_PLAN=1: the engine takes the landmark-major form of the Schur complement
    (k_schur_fused + k_hsc_reduce, no T array) — same trajectory as the oracle, sharded too"""
    monkeypatch.setenv("CUGO_SCHUR_PLAN", "1")
    d, prob = synth_problem(oracle_lib, 400, 8000, 33000, seed=11, lc=200)
    out = run_graph(d, 10)
    ref = prob.optimize(10)
    assert out["sstats"]["schur_slots"] > 0
    assert_trajectories_match(out["stats"], ref, 1e-10)
    assert rmse(out["pose"], prob.pose) < 1e-9 and rmse(out["lm"], prob.lm) < 1e-9
    res = run_sharded_in_threads(d, 2, 6)
    for r in range(2):
        assert_trajectories_match(res[r]["stats"], ref[:6], 1e-10)


def test_outlier_rejection_matches_oracle(oracle_lib):
    """ref: EdgeSet::setOutlierThreshold + updateEdges (optimisable_graph.hpp:603-640): after
    optimize() the edges whose chi2 exceeds the set's threshold are inactivated and left out of
    the next initialize()/optimize()."""
    d, prob = synth_problem(oracle_lib, 40, 600, 2400, seed=21, lc=0)
    rng = np.random.default_rng(4)
    bad = rng.choice(len(d["e_pose"]), 30, replace=False)
    d["e_meas"] = d["e_meas"].copy()
    d["e_meas"][bad, :2] += rng.choice([-1.0, 1.0], (30, 2)) * 60.0  # gross outliers
    prob = oracle_lib.Problem(d["pose"], d["pose_fixed"], d["lm"], d["lm_fixed"], d["e_pose"], d["e_lm"],
                              d["e_stereo"], d["e_meas"], d["e_omega"], d["e_cam"])
    th = {2: 5.991, 3: 7.815}
    g = cugo.graph_from_arrays(d)
    for dim, t in th.items():
        g.set_outlier_threshold(dim, t)
    g.initialize()
    g.optimize(5)
    got = g.stats()
    ref = prob.optimize(5)
    assert_trajectories_match(got, ref, 1e-10)
    assert ref[-1]["rho"] > 0  # last trial accepted: the last error pass is at the final estimates
    # per-edge chi2 at the oracle's final estimates
    _, err, _ = prob.compute_errors(want_arrays=True)
    st = d["e_stereo"].astype(bool)
    chi = d["e_omega"] * np.sum(err ** 2, axis=1)
    thr = np.where(st, th[3], th[2])
    expect_out = chi > thr
    clear = np.abs(chi - thr) > 1e-6
    # active flags per set, in insertion order (= mono edges then stereo edges of d)
    act = np.ones(len(chi), bool)
    act[~st] = g.edge_active(2, int((~st).sum()))
    act[st] = g.edge_active(3, int(st.sum()))
    assert np.array_equal(~act[clear], expect_out[clear])
    assert expect_out[bad].mean() > 0.8 and expect_out.sum() < 0.2 * len(chi)
    assert g.n_outliers(2) + g.n_outliers(3) == int((~act).sum())
    assert g.n_active_edges() == len(chi) - int((~act).sum())
    # next optimisation runs without the flagged edges: same as the oracle on the reduced graph
    keep = act
    pose1, lm1 = g.poses(), g.landmarks()
    g.initialize()
    g.optimize(5)
    got2 = g.stats()
    prob2 = oracle_lib.Problem(pose1, d["pose_fixed"], lm1, d["lm_fixed"], d["e_pose"][keep], d["e_lm"][keep],
                               d["e_stereo"][keep], d["e_meas"][keep], d["e_omega"][keep], d["e_cam"][keep])
    ref2 = prob2.optimize(5)
    assert_trajectories_match(got2, ref2, 1e-9)
    assert got2[-1]["chi2"] < 0.5 * got[-1]["chi2"]
    g.close()


def dense_ring_problem(n_poses, n_landmarks, seed):
    """cameras on a ring looking at a small cloud at the origin: EVERY landmark is seen by EVERY
    pose (> 256 edges per landmark, a dense Schur complement)"""
    import synth
    rng = np.random.default_rng(seed)
    cam = np.array([718.856, 718.856, 607.1928, 185.2157, 386.1448])
    lm_true = rng.uniform(-1.0, 1.0, (n_landmarks, 3))
    poses, meas, ep, el = [], [], [], []
    for i in range(n_poses):
        a = 2 * np.pi * i / n_poses
        C = np.array([12.0 * np.cos(a), 0.4 * np.sin(3 * a), 12.0 * np.sin(a)])
        z = -C / np.linalg.norm(C)
        x = np.cross(np.array([0.0, 1.0, 0.0]), z)
        x /= np.linalg.norm(x)
        y = np.cross(z, x)
        R = np.stack([x, y, z])  # rows = camera axes in world coordinates
        t = -R @ C
        poses.append(np.concatenate([synth._R_to_quat(R), t]))
        Xc = lm_true @ R.T + t
        u = cam[0] * Xc[:, 0] / Xc[:, 2] + cam[2]
        v = cam[1] * Xc[:, 1] / Xc[:, 2] + cam[3]
        ur = u - cam[4] / Xc[:, 2]
        meas.append(np.stack([u, v, ur], 1) + rng.normal(0, 0.5, (n_landmarks, 3)))
        ep.append(np.full(n_landmarks, i)), el.append(np.arange(n_landmarks))
    pose = np.array(poses)
    E = n_poses * n_landmarks
    d = dict(pose=pose.copy(), lm=lm_true + rng.normal(0, 0.02, lm_true.shape),
             pose_fixed=np.zeros(n_poses, np.uint8), lm_fixed=np.zeros(n_landmarks, np.uint8),
             e_pose=np.concatenate(ep).astype(np.int32), e_lm=np.concatenate(el).astype(np.int32),
             e_stereo=(np.arange(E) % 3 == 0).astype(np.uint8), e_meas=np.concatenate(meas),
             e_omega=np.full(E, 1.0), e_cam=np.tile(cam, (E, 1)))
    d["pose_fixed"][0] = 1
    for i in range(1, n_poses):  # perturb the free poses a little
        dq = synth.quat_from_rotvec(rng.normal(0, 0.002, 3))
        d["pose"][i, :4] = synth.quat_mul(dq, pose[i, :4])
        d["pose"][i, 4:] = synth.quat_to_R(dq) @ pose[i, 4:] + rng.normal(0, 0.01, 3)
    d["e_meas"][d["e_stereo"] == 0, 2] = 0.0
    return d


def test_landmarks_with_more_than_256_edges_and_dense_schur(oracle_lib):
    """every landmark is observed by 300 poses: the per-landmark sums leave their 256-edge block
    (owner-lane slow paths of k_build_edges / k_backsubst_landmarks), Hsc is completely dense and
    the elimination tree is one long chain of fronts"""
    d = dense_ring_problem(300, 12, seed=3)
    prob = oracle_lib.Problem(d["pose"], d["pose_fixed"], d["lm"], d["lm_fixed"], d["e_pose"], d["e_lm"],
                              d["e_stereo"], d["e_meas"], d["e_omega"], d["e_cam"])
    out = run_graph(d, 6)
    ref = prob.optimize(6)
    assert_trajectories_match(out["stats"], ref, 1e-10)
    assert rmse(out["pose"], prob.pose) < 1e-9 and rmse(out["lm"], prob.lm) < 1e-9
    assert out["stats"][-1]["chi2"] <= out["stats"][0]["chi2"]


@pytest.mark.parametrize("rows", [False, True])
@pytest.mark.parametrize("which", ["all_landmarks_fixed", "all_poses_fixed"])
def test_degenerate_fixed_sets(oracle_lib, which, rows, monkeypatch):
    """pose-only BA (no free landmark: the Schur complement is just Hpp) and structure-only BA
    (no free pose: an empty pose system, landmarks solved by their 3x3 blocks); also with the opt-in
    block-row Schur kernel, which has to hand such structures back to the gather kernels"""
    if rows:
        monkeypatch.setenv("CUGO_HSC_ROWS", "1")
    d, _ = synth_problem(oracle_lib, 30, 400, 1600, seed=5, lc=0)
    if which == "all_landmarks_fixed":
        d["lm_fixed"] = np.ones(400, np.uint8)
    else:
        d["pose_fixed"] = np.ones(30, np.uint8)
    prob = oracle_lib.Problem(d["pose"], d["pose_fixed"], d["lm"], d["lm_fixed"], d["e_pose"], d["e_lm"],
                              d["e_stereo"], d["e_meas"], d["e_omega"], d["e_cam"])
    out = run_graph(d, 5)
    ref = prob.optimize(5)
    assert_trajectories_match(out["stats"], ref, 1e-10)
    assert rmse(out["pose"], prob.pose) < 1e-10 and rmse(out["lm"], prob.lm) < 1e-10


def test_global_information_and_camera_options(oracle_lib):
    d, prob = synth_problem(oracle_lib, 120, 1500, 6200, seed=5, lc=0)
    d["e_omega"][:] = 0.75
    prob.e_omega[:] = 0.75
    a = run_graph(d, 6)                                                   # per-edge arrays
    b = run_graph(d, 6, per_edge_information=False, per_edge_camera=False)  # one value per set
    ref = prob.optimize(6)
    assert_trajectories_match(a["stats"], ref, 1e-10)
    assert [s["chi2"] for s in a["stats"]] == [s["chi2"] for s in b["stats"]]  # same device data
    assert np.array_equal(a["pose"], b["pose"])


@pytest.mark.parametrize("f32", [False, True])
def test_build_pass_forming_T_on_the_side_is_bitwise_the_separate_kernel(oracle_lib, f32, monkeypatch):
    """from the second LM iteration on the build pass also leaves invHll and T = Hpl invHll for the
    first trial's damping (k_build_edges, fuse_lambda >= 0); CUGO_FUSE_T=0 keeps the separate edge
    kernel of the Schur complement.  Same arrays in, same arithmetic: bitwise the same run — also
    with float block storage (T is formed from the ROUNDED Hpl values there) and with rejected
    trials in between (reject_8x60 takes the separate kernel for its retries).  (CUGO_POSE_SCHUR=0: the pose pass of the
    fused iteration, which forms the diagonal blocks another way, has its own test below.)"""
    monkeypatch.setenv("CUGO_POSE_SCHUR", "0")
    d, _ = synth_problem(oracle_lib, 200, 3000, 12500, seed=23, lc=100)
    g8 = np.load(golden_path("reject_8x60.npz"))
    cases = [d, {k: g8[k] for k in PROBLEM_KEYS}]
    for dd in cases:
        runs = []
        for v in ("1", "0"):
            monkeypatch.setenv("CUGO_FUSE_T", v)
            g = cugo.graph_from_arrays(dd)
            g.set_float32(f32)
            g.initialize(); g.optimize(8)
            runs.append((g.stats(), g.poses(), g.landmarks()))
            g.close()
        assert [s["chi2"] for s in runs[0][0]] == [s["chi2"] for s in runs[1][0]]
        assert [s["trials"] for s in runs[0][0]] == [s["trials"] for s in runs[1][0]]
        assert np.array_equal(runs[0][1], runs[1][1]) and np.array_equal(runs[0][2], runs[1][2])


@pytest.mark.parametrize("speculate", ["1", "0"])
def test_fused_pose_pass_matches_the_separate_kernels(oracle_lib, speculate, monkeypatch):
    """from the second LM iteration on, the diagonal blocks of Hsc, bp and bsc come from ONE pass over the build
    pass's records (k_pose_schur: JP^T (w I - w^2 JL invHll JL^T) JP per edge) instead of k_build_poses +
    k_hsc_diag_mfma over the stored T / Hpl blocks (CUGO_POSE_SCHUR=0).  Same sums in another order: chi2 of every
    iteration within the parity tolerance of the fixture (1e-10 relative; the measured self-sensitivity for the stress
    fixture), the same trial counts and damping, estimates within 1e-7 — on a graph with loop closures,
    on the golden case whose trials get rejected (the retries need Hpp, which the fused build pass does not write:
    with the speculative build switched off the whole build pass is repeated, with it the records are still there or
    the rebuild happens anyway), and on golden cases with robust kernels, loop closures and stereo edges.  (Fixed
    vertices, per-edge information and several cameras: the tests of those features run the fused iteration by default
    and compare with the oracle.)"""
    monkeypatch.setenv("CUGO_SPECULATE", speculate)
    d, _ = synth_problem(oracle_lib, 200, 3000, 12500, seed=23, lc=100)
    cases = [(None, d, (0, 0.0))]
    for name in ("reject_8x60", "huber_8x80", "tukey_8x80", "loop_12x150", "small_10x200"):
        g8 = np.load(golden_path(name + ".npz"))
        cases.append((name, {k: g8[k] for k in PROBLEM_KEYS}, (int(g8["rk_type"]), float(g8["rk_delta"]))))
    for name, dd, rk in cases:
        # (the stress fixture amplifies round-off by itself: its bar is the measured one of conftest.py)
        tol, etol = golden_tolerances(name, 8) if name else ([1e-10] * 8, 1e-9)
        runs = []
        for v in ("1", "0"):
            monkeypatch.setenv("CUGO_POSE_SCHUR", v)
            runs.append(run_graph(dd, 8, rk=rk))
        assert_trajectories_match(runs[0]["stats"], runs[1]["stats"], tol[:len(runs[1]["stats"])])
        np.testing.assert_allclose(runs[0]["pose"], runs[1]["pose"], rtol=0, atol=100 * etol)
        np.testing.assert_allclose(runs[0]["lm"], runs[1]["lm"], rtol=0, atol=1000 * etol)


def test_trial_chi2_out_of_the_next_build_pass_gives_the_same_bits(oracle_lib, monkeypatch):
    """a first trial's chi2 comes out of the NEXT iteration's build pass — queued at the trial's estimates before the
    result is known, followed by the trial's reductions and by the Schur complement for the predicted damping — instead
    of an error pass of its own (CUGO_TRIAL_FROM_BUILD=0).  The same residuals summed in the same blocks: the same
    F-hat, so the same decisions and the same bits at the end — on a graph whose trials are all accepted, on the
    golden case whose trials get rejected (the speculative work is thrown away and H rebuilt) and with robust
    kernels (whose dampings miss the prediction now and then: the queued Schur complement is redone for the real one)."""
    d, _ = synth_problem(oracle_lib, 200, 3000, 12500, seed=31, lc=100)
    cases = [(d, (0, 0.0))]
    # (tukey_8x80 and tiny_3x8 end with accepted steps whose damping is NOT the predicted one right behind steps whose
    # damping was: the queued Schur complement is redone; reject_8x60 rejects a first trial behind such a step)
    for name in ("reject_8x60", "huber_8x80", "loop_12x150", "tukey_8x80", "tiny_3x8"):
        g8 = np.load(golden_path(name + ".npz"))
        cases.append(({k: g8[k] for k in PROBLEM_KEYS}, (int(g8["rk_type"]), float(g8["rk_delta"]))))
    for dd, rk in cases:
        runs = []
        for v in ("1", "0"):
            monkeypatch.setenv("CUGO_TRIAL_FROM_BUILD", v)
            runs.append(run_graph(dd, 10, rk=rk))
        assert [(s["chi2"], s["lam"], s["trials"]) for s in runs[0]["stats"]] == \
               [(s["chi2"], s["lam"], s["trials"]) for s in runs[1]["stats"]]
        assert np.array_equal(runs[0]["pose"], runs[1]["pose"]) and np.array_equal(runs[0]["lm"], runs[1]["lm"])


def test_bitwise_reproducible(oracle_lib):
    d, _ = synth_problem(oracle_lib, 200, 3000, 12500, seed=21, lc=100)
    a, b = run_graph(d, 8), run_graph(d, 8)
    assert [s["chi2"] for s in a["stats"]] == [s["chi2"] for s in b["stats"]]
    assert np.array_equal(a["pose"], b["pose"]) and np.array_equal(a["lm"], b["lm"])


def test_float32_block_storage(oracle_lib):
    """BASELINE config 5 (fp32 internal).  The reference's USE_FLOAT32 build does not compile in this
    fork, so there is no reference behaviour to match; this build's mode stores the Hpl
    and Hpl*Hll^-1 block streams as float and keeps everything else fp64.  Stated tolerance: chi2
    of every iteration within 1e-5 relative of the fp64 oracle, same number of LM trials, final
    estimates within 1e-4 (poses) / 1e-3 (landmarks) absolute of the fp64 run."""
    d, prob = synth_problem(oracle_lib, 400, 8000, 33000, seed=11, lc=200)
    g = cugo.graph_from_arrays(d)
    g.set_float32(True)
    g.initialize(); g.optimize(10)
    st32, pose32, lm32 = g.stats(), g.poses(), g.landmarks()
    g.set_float32(False)  # the same graph object goes back to fp64 at the next initialize()
    ids_p, ids_l = np.arange(len(d["pose"]), dtype=np.int32), np.arange(len(d["lm"]), dtype=np.int32)
    g.set_poses(ids_p, d["pose"]); g.set_landmarks(ids_l, d["lm"])
    g.initialize(); g.optimize(10)
    st64 = g.stats()
    g.close()
    ref = prob.optimize(10)
    assert_trajectories_match(st64, ref, 1e-10)
    rel = max(abs(a["chi2"] - b["chi2"]) / b["chi2"] for a, b in zip(st32, ref))
    assert [a["trials"] for a in st32] == [b["trials"] for b in ref]
    assert 1e-13 < rel < 1e-5, rel  # differs from fp64 (float storage in use), within the stated tolerance
    assert np.abs(pose32 - prob.pose).max() < 1e-4 and np.abs(lm32 - prob.lm).max() < 1e-3
    print("fp32 block storage: max rel chi2 diff %.3e, pose %.3e, landmark %.3e" %
          (rel, np.abs(pose32 - prob.pose).max(), np.abs(lm32 - prob.lm).max()))


@pytest.mark.parametrize("stereo", [0.0, 0.7])
def test_edge_insertion_order_does_not_matter(oracle_lib, stereo):
    """initialize() sorts the edges landmark-major (threaded above 100k edges; a shortcut when the
    caller already added them landmark by landmark): any insertion order gives the same device
    layout, hence bitwise the same run.  Mono-only input in landmark order takes the shortcut,
    the shuffled copy and the mono|stereo split the general path."""
    d = cugo.synth(300, 30000, 126000, seed=5, n_loop_closures=300, stereo_fraction=stereo)
    assert bool(np.all(np.diff(d["e_lm"]) >= 0))  # the generator emits landmark-major edges
    a = run_graph(d, 4)
    perm = np.random.default_rng(1).permutation(len(d["e_pose"]))
    sh = dict(d)
    for k in ("e_pose", "e_lm", "e_stereo", "e_meas", "e_omega"):
        sh[k] = np.ascontiguousarray(np.asarray(d[k])[perm])
    if np.asarray(d["e_cam"]).reshape(-1, 5).shape[0] > 1:
        sh["e_cam"] = np.ascontiguousarray(np.asarray(d["e_cam"]).reshape(-1, 5)[perm])
    b = run_graph(sh, 4)
    assert [x["chi2"] for x in a["stats"]] == [x["chi2"] for x in b["stats"]]
    assert np.array_equal(a["pose"], b["pose"]) and np.array_equal(a["lm"], b["lm"])
    prob = oracle_lib.Problem(*synth.problem_fields(d))
    assert_trajectories_match(a["stats"], prob.optimize(4), 1e-10)


def test_ids_fixed_vertices_and_reinitialize(oracle_lib):
    """non-contiguous ids, fixed poses and landmarks, the sample's warm-up protocol
    (initialize; optimize(1); initialize; optimize(n) — ref samples/sample_ba_from_file/main.cpp:168-188)"""
    dd = synth.make_problem(n_poses=12, n_landmarks=220, seed=31, fixed_poses=(0, 7), fixed_landmarks=(5, 9, 100),
                            loop_closure=True)
    pose_ids = np.arange(12) * 3 + 100
    lm_ids = np.arange(220)[::-1] * 2 + 7  # descending ids: index order differs from array order
    g = cugo.graph_from_arrays(dd, pose_ids=pose_ids, lm_ids=lm_ids)
    g.initialize(); g.optimize(1)
    first = g.stats()
    g.initialize(); g.optimize(5)
    st = g.stats()
    assert len(first) == 1 and len(st) == 5  # initialize() clears the statistics
    prob = oracle_lib.Problem(*synth.problem_fields(dd))
    # the oracle wants vertices in ascending-id order: landmark ids are descending -> reverse
    rev = oracle_lib.Problem(dd["pose"], dd["pose_fixed"], dd["lm"][::-1], dd["lm_fixed"][::-1], dd["e_pose"],
                             219 - dd["e_lm"], dd["e_stereo"], dd["e_meas"], dd["e_omega"], dd["e_cam"])
    r1 = rev.optimize(1)
    r5 = rev.optimize(5)
    assert_trajectories_match(first, r1, 1e-10)
    assert_trajectories_match(st, r5, 1e-10)
    np.testing.assert_allclose(g.poses(pose_ids), rev.pose, rtol=0, atol=1e-9)
    np.testing.assert_allclose(g.landmarks(lm_ids), rev.lm[::-1], rtol=0, atol=1e-8)
    assert np.array_equal(g.poses(pose_ids[[0, 7]]), dd["pose"][[0, 7]])      # fixed stay put
    assert np.array_equal(g.landmarks(lm_ids[[5, 9, 100]]), dd["lm"][[5, 9, 100]])
    g.close()
    del prob


@pytest.fixture(scope="module")
def kitti00(oracle_lib):
    """BASELINE configs 2 / 3 / 5 share this graph (1322 / 133 383 / 561 116, seed 0) and ONE
    10-iteration run of the fp64 oracle on it."""
    d, prob = synth_problem(oracle_lib, 1322, 133383, 561116, seed=0, lc=4000)
    ref = prob.optimize(10)
    return dict(d=d, ref=ref, pose=prob.pose.copy(), lm=prob.lm.copy())


def test_reinitialize_on_an_unchanged_graph_refreshes_estimates_only(oracle_lib):
    """initialize() on a graph in which only vertex estimates changed keeps the flattened graph on
    the device (change tracking of the vertex / edge sets) — bitwise the result of a full
    flattening; any other change (set-wide information, new edges, a newly fixed vertex through a
    re-added vertex set) takes the full path again and is seen."""
    d, prob = synth_problem(oracle_lib, 120, 1500, 6200, seed=5, lc=0)
    d["e_omega"][:] = 0.75
    ids_p, ids_l = np.arange(120, dtype=np.int32), np.arange(1500, dtype=np.int32)
    g = cugo.graph_from_arrays(d, per_edge_information=False)
    g.initialize(); g.optimize(4)
    first, pose1 = g.stats(), g.poses()
    assert g.flatten_reuses() == 0
    g.set_poses(ids_p, d["pose"]); g.set_landmarks(ids_l, d["lm"])
    g.initialize(); g.optimize(4)                      # estimates only
    assert g.flatten_reuses() == 1
    assert [s["chi2"] for s in g.stats()] == [s["chi2"] for s in first] and np.array_equal(g.poses(), pose1)
    # two refreshes in a row: the estimates travel through pinned staging by a copy nobody waits for, and the second
    # gather may overwrite the staging before the first copy has run — the later copy is the one that counts
    moved = d["pose"].copy(); moved[:, 4:] += 0.01
    g.set_poses(ids_p, moved); g.set_landmarks(ids_l, d["lm"] + 0.01)
    g.initialize()
    g.set_poses(ids_p, d["pose"]); g.set_landmarks(ids_l, d["lm"])
    g.initialize(); g.optimize(4)
    assert g.flatten_reuses() == 3
    assert [s["chi2"] for s in g.stats()] == [s["chi2"] for s in first] and np.array_equal(g.poses(), pose1)
    g.initialize(); g.optimize(2)                      # continues from the optimised estimates
    assert g.flatten_reuses() == 4
    prob.e_omega[:] = 0.75
    ref = prob.optimize(4)
    assert_trajectories_match(first, ref, 1e-10)
    cont = prob.optimize(2)
    assert_trajectories_match(g.stats(), cont, 1e-10)
    # a set-wide information value is a change: full flattening, new weights in effect
    for dim in (2, 3):
        g.set_information(dim, 0.25)
    g.set_poses(ids_p, d["pose"]); g.set_landmarks(ids_l, d["lm"])
    g.initialize(); g.optimize(3)
    assert g.flatten_reuses() == 4
    p2 = oracle_lib.Problem(*[d[k] for k in PROBLEM_KEYS])
    p2.e_omega[:] = 0.25
    assert_trajectories_match(g.stats(), p2.optimize(3), 1e-10)
    # new edges are a change too
    extra = cugo.synth(120, 1500, 6200, seed=6)
    sel = np.flatnonzero(extra["e_stereo"] == 0)[:50]
    have = set(zip(d["e_pose"].tolist(), d["e_lm"].tolist()))
    sel = np.array([i for i in sel if (int(extra["e_pose"][i]), int(extra["e_lm"][i])) not in have], np.int64)
    g.add_edges(2, extra["e_pose"][sel], extra["e_lm"][sel], extra["e_meas"][sel], extra["e_omega"][sel],
                extra["e_cam"][sel])
    g.set_poses(ids_p, d["pose"]); g.set_landmarks(ids_l, d["lm"])
    g.initialize()
    assert g.flatten_reuses() == 4 and g.n_active_edges() == 6200 + len(sel)
    g.close()


def test_kitti00_shape_full_size(kitti00):
    """BASELINE config 2 shape (1322 / 133 383 / 561 116): parity with the oracle plus
    size-independent properties (monotone chi2, bitwise reproducibility)."""
    d = kitti00["d"]
    out = run_graph(d, 10)
    chi = [s["chi2"] for s in out["stats"]]
    assert len(chi) == 10 and all(b < a for a, b in zip(chi, chi[1:]))
    assert out["nedges"] == 561116
    assert_trajectories_match(out["stats"], kitti00["ref"], 1e-10)
    assert rmse(out["pose"][:, :4], kitti00["pose"][:, :4]) < 1e-11
    assert rmse(out["pose"][:, 4:], kitti00["pose"][:, 4:]) < 1e-9
    assert rmse(out["lm"], kitti00["lm"]) < 1e-8
    again = run_graph(d, 10)
    assert chi == [s["chi2"] for s in again["stats"]]


def test_kitti07_shape_full_size(oracle_lib):
    """BASELINE config 1's shape (ba_kitti_07: 248 / 26 127 / 95 037, seed 7, the README's other
    dataset): 10 LM iterations of the HIP path against the CPU oracle (the "g2o CPU reference path" of
    that config) at 1e-10, monotone chi2, estimates compared as the reference's README does."""
    d = cugo.synth(248, 26127, 95037, seed=7, n_loop_closures=500)
    out = run_graph(d, 10)
    prob = oracle_lib.Problem(*[d[k] for k in PROBLEM_KEYS])
    ref = prob.optimize(10)
    chi = [s["chi2"] for s in out["stats"]]
    assert len(chi) == 10 and all(b < a for a, b in zip(chi, chi[1:]))
    assert out["nedges"] == 95037
    assert_trajectories_match(out["stats"], ref, 1e-10)
    assert rmse(out["pose"][:, :4], prob.pose[:, :4]) < 1e-11
    assert rmse(out["pose"][:, 4:], prob.pose[:, 4:]) < 1e-9
    assert rmse(out["lm"], prob.lm) < 1e-8


def test_kitti00_float32_full_size(kitti00):
    """BASELINE config 5 at full size: the fp32-internal mode (float storage of the Hpl / Hpl*invHll
    block streams, everything else fp64) against the fp64 oracle, tolerance of DESIGN.md section 4a:
    chi2 of every iteration within 1e-5 relative, same LM trial counts, poses within 1e-4 and
    landmarks within 1e-3 absolute."""
    d = kitti00["d"]
    g = cugo.graph_from_arrays(d)
    g.set_float32(True)
    g.initialize(); g.optimize(10)
    st, pose, lm = g.stats(), g.poses(), g.landmarks()
    g.close()
    ref = kitti00["ref"]
    assert len(st) == len(ref) == 10
    assert [a["trials"] for a in st] == [b["trials"] for b in ref]
    rel = max(abs(a["chi2"] - b["chi2"]) / b["chi2"] for a, b in zip(st, ref))
    assert 1e-13 < rel < 1e-5, rel   # float storage really in use, and inside the stated tolerance
    assert np.abs(pose - kitti00["pose"]).max() < 1e-4 and np.abs(lm - kitti00["lm"]).max() < 1e-3
    print("config 5 full size: max rel chi2 diff %.3e, pose %.3e, landmark %.3e"
          % (rel, np.abs(pose - kitti00["pose"]).max(), np.abs(lm - kitti00["lm"]).max()))


def test_sharded_device_structure_build_equals_host_build(monkeypatch):
    """a shard builds the GLOBAL Hsc pattern on the device from the global co-visibility lists and
    its contribution lists from its local slots (structure_gpu.cpp, sharded form): same structure
    statistics and bitwise the same trajectory as with the host build, for 2 and 3 shards, and a
    graph whose last shard owns fixed landmarks only"""
    cases = [(cugo.synth(160, 2500, 10500, seed=3, n_loop_closures=80), 2),
             (synth.make_problem(n_poses=30, n_landmarks=400, mean_obs=5.0, seed=9, fixed_poses=(0, 7, 8),
                                 fixed_landmarks=tuple(range(300, 400)), loop_closure=True), 3)]
    for d, world in cases:
        dev = run_sharded_in_threads(d, world, 4, want_sstats=True)
        monkeypatch.setenv("CUGO_HOST_STRUCTURE", "1")
        host = run_sharded_in_threads(d, world, 4, want_sstats=True)
        monkeypatch.delenv("CUGO_HOST_STRUCTURE")
        for r in range(world):
            assert dev[r]["sstats"] == host[r]["sstats"], (r, dev[r]["sstats"], host[r]["sstats"])
            assert [s["chi2"] for s in dev[r]["stats"]] == [s["chi2"] for s in host[r]["stats"]]
            assert np.array_equal(dev[r]["pose"], host[r]["pose"]) and np.array_equal(dev[r]["lm"], host[r]["lm"])


def run_sharded_in_threads(d, world, niter, want_sstats=False):
    """landmark-sharded run with `world` shards emulated in ONE process on one GPU: one optimiser
    per shard in its own thread, the all-reduce goes through host memory (the callback form of the
    exchange; the RCCL form needs one GPU per rank).  Returns the per-rank results."""
    import threading
    import devmem
    bufs, barrier, results = {}, threading.Barrier(world), [None] * world
    lock = threading.Lock()
    ctxs = [devmem.Ctx() for _ in range(world)]

    def make_exchange(rank):
        def fn(ptr, n, op):
            host = np.zeros(n)
            cugo.check(cugo.lib().cugo_memcpy_d2h(ctxs[rank].h, host.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), 8 * n))
            with lock:
                bufs[rank] = host
            barrier.wait()
            if op >= 2:  # broadcast from rank op - 2 (rank-owned elimination subtrees)
                tot = bufs[op - 2].copy()
            elif op == -1:  # reduce-scatter of `world` equal segments: this rank gets the sum of ITS segment only
                seg = n // world
                tot = np.full(n, np.nan)   # (nothing outside the own segment may be read afterwards)
                acc = bufs[0][rank * seg:(rank + 1) * seg].copy()
                for r in range(1, world):  # fixed rank order
                    acc = acc + bufs[r][rank * seg:(rank + 1) * seg]
                tot[rank * seg:(rank + 1) * seg] = acc
            else:
                tot = bufs[0].copy()
                for r in range(1, world):  # fixed rank order
                    tot = tot + bufs[r] if op == 0 else np.maximum(tot, bufs[r])
            barrier.wait()
            cugo.check(cugo.lib().cugo_memcpy_h2d(ctxs[rank].h, C.c_void_p(ptr), tot.ctypes.data_as(C.c_void_p), 8 * n))
        return fn

    def worker(rank):
        g = cugo.graph_from_arrays(d)
        g.set_shard(rank, world, make_exchange(rank))
        g.initialize()
        g.optimize(niter)
        results[rank] = dict(stats=g.stats(), pose=g.poses(), lm=g.landmarks(), xstats=g.exchange_stats(),
                             nedges=g.n_active_edges(), sstats=g.structure_stats() if want_sstats else None)
        g.close()

    ts = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for c in ctxs:
        c.close()
    return results


@pytest.mark.parametrize("world", [2, 3, 4])
def test_rank_owned_elimination_subtrees_match_unsharded(oracle_lib, world, monkeypatch):
    """a sharded run factors the sparse LL^T by rank-owned elimination subtrees (chol_symbolic.cpp:
    CholPlan::owner): every rank factors its own subtrees and the replicated top of the tree, the update
    blocks that enter the top and the other ranks' solution ranges arrive by broadcast.  Same trajectory
    as the unsharded run and as the replicated form (CUGO_OWN_SUBTREES=0); the work really is split."""
    d, prob = synth_problem(oracle_lib, 700, 12000, 50000, seed=21, lc=150)
    single = run_graph(d, 6)
    # (a graph this small stays replicated unless the form is forced: CUGO_OWN_MIN_GFLOP, chol_solver.cpp)
    auto = run_sharded_in_threads(d, world, 2, want_sstats=True)
    assert auto[0]["sstats"]["chol_bcasts"] == 0
    monkeypatch.setenv("CUGO_OWN_SUBTREES", "1")
    res = run_sharded_in_threads(d, world, 6, want_sstats=True)
    ref = prob.optimize(6)
    total = None
    for r in range(world):
        assert_trajectories_match(res[r]["stats"], single["stats"], 1e-11)
        assert_trajectories_match(res[r]["stats"], ref, 1e-10)
        assert rmse(res[r]["pose"], single["pose"]) < 1e-11 and rmse(res[r]["lm"], single["lm"]) < 1e-10
        ss = res[r]["sstats"]
        assert ss["chol_bcasts"] >= 2 and ss["chol_bcast_bytes"] > 0
        assert ss["chol_top_flops"] == res[0]["sstats"]["chol_top_flops"]
        total = (total or 0.0) + ss["chol_rank_flops"]
    top = res[0]["sstats"]["chol_top_flops"]
    own = [res[r]["sstats"]["chol_rank_flops"] for r in range(world)]
    assert all(o > 0 for o in own) and max(own) < 0.75 * (total + top)   # nobody factors (nearly) everything
    assert [s["chi2"] for s in res[0]["stats"]] == [s["chi2"] for s in res[1]["stats"]]  # ranks agree bitwise
    monkeypatch.setenv("CUGO_OWN_SUBTREES", "0")
    rep = run_sharded_in_threads(d, world, 6, want_sstats=True)
    assert rep[0]["sstats"]["chol_bcasts"] == 0
    assert_trajectories_match(rep[0]["stats"], res[0]["stats"], 1e-11)


def test_synth10k_eight_shards_with_owned_subtrees():
    """BASELINE config 4 (10 000 poses / 1 M landmarks / 5 M edges) in the form it runs on 8 GPUs — eight
    landmark shards, the LL^T split into rank-owned elimination subtrees below a replicated top — emulated
    with eight optimisers in threads on one GPU: the first LM iterations match the unsharded run at 1e-10,
    and per rank the Cholesky work and the bytes exchanged are what the plan says."""
    d = cugo.synth(10000, 1000000, 5000000, seed=10000, n_loop_closures=0, stereo_fraction=0.0)
    single = run_graph(d, 2)
    res = run_sharded_in_threads(d, 8, 2, want_sstats=True)
    top = res[0]["sstats"]["chol_top_flops"]
    own = [res[r]["sstats"]["chol_rank_flops"] for r in range(8)]
    whole = top + sum(own)
    for r in range(8):
        assert_trajectories_match(res[r]["stats"], single["stats"], 1e-10)
        assert res[r]["sstats"]["chol_top_flops"] == top
        # a rank's factorisation: its own subtrees + the replicated top — well under the whole
        assert own[r] + top < 0.45 * whole
        assert res[r]["sstats"]["chol_bcast_bytes"] < 0.5 * 8 * 36 * res[r]["sstats"]["hsc_blocks"]
        # the per-trial exchange of [Hsc | bsc] is keyed on front ownership: a rank RECEIVES its own segment
        # (reduce-scatter) and the top's part (all-reduce), not the whole system.  The callback of this emulation
        # hands back NaNs outside the own segment, so a unit read from elsewhere would wreck the trajectory above.
        ss = res[r]["sstats"]
        full = 8 * 36 * ss["hsc_blocks"]
        assert ss["xchg_sys_full_bytes"] >= full
        assert ss["xchg_sys_bytes"] <= (1.0 / 8 + top / whole + 0.10) * full, (ss["xchg_sys_bytes"] / full, top / whole)
    assert max(own) < 1.6 * (sum(own) / 8)   # balanced within 60 %
    print("config 4, 8 shards: replicated top %.1f %% of the factorisation, own shares %s %%; broadcast %.1f MB per trial; "
          "Schur system received per rank and trial %.1f MB (all-reduce payload %.1f MB)"
          % (100 * top / whole, [round(100 * o / whole, 1) for o in own], res[0]["sstats"]["chol_bcast_bytes"] / 1e6,
             res[0]["sstats"]["xchg_sys_bytes"] / 1e6, res[0]["sstats"]["xchg_sys_full_bytes"] / 1e6))


def test_kitti00_two_shards_match_unsharded(kitti00):
    """BASELINE config 3 (kitti_00 landmark-sharded) at full size with 2 shards on one GPU: both
    ranks reproduce the unsharded chi2 trajectory (the partial Schur systems are summed in a
    different order, so not bitwise) and the oracle's, and end at the same estimates."""
    d = kitti00["d"]
    res = run_sharded_in_threads(d, 2, 10)
    single = run_graph(d, 10)
    for r in range(2):
        assert res[r] is not None and res[r]["nedges"] == 561116
        assert_trajectories_match(res[r]["stats"], single["stats"], 1e-11)
        assert_trajectories_match(res[r]["stats"], kitti00["ref"], 1e-10)
        assert rmse(res[r]["pose"], single["pose"]) < 1e-11 and rmse(res[r]["lm"], single["lm"]) < 1e-10
        # exchanges: 3 at iteration 0 + 2 per LM trial; payload = [Hsc | bsc] + scalars
        trials = sum(max(s["trials"], 0) + 1 for s in res[r]["stats"])
        assert res[r]["xstats"]["calls"] >= 3 + 2 * 10 and res[r]["xstats"]["calls"] <= 3 + 2 * trials + 2
    assert [s["chi2"] for s in res[0]["stats"]] == [s["chi2"] for s in res[1]["stats"]]  # ranks agree bitwise


@pytest.mark.parametrize("form", ["default", "owned"])
def test_kitti00_eight_shards_match_unsharded(kitti00, form, monkeypatch):
    """BASELINE config 3 at its real world size: kitti_00 landmark-sharded over 8 ranks (eight optimisers in threads
    on one GPU, exchange through the callback).  "default": the policy of a real run — 1.6 GFLOP per factorisation is
    below CUGO_OWN_MIN_GFLOP, so the LL^T stays replicated and the system is all-reduced; "owned": rank-owned
    elimination subtrees forced, with the ownership-keyed reduce-scatter.  Both reproduce the unsharded trajectory
    and the oracle's, all ranks bitwise equal."""
    if form == "owned":
        monkeypatch.setenv("CUGO_OWN_SUBTREES", "1")
    d = kitti00["d"]
    res = run_sharded_in_threads(d, 8, 10, want_sstats=True)
    single = run_graph(d, 10)
    for r in range(8):
        assert res[r] is not None and res[r]["nedges"] == 561116
        assert_trajectories_match(res[r]["stats"], single["stats"], 1e-11)
        assert_trajectories_match(res[r]["stats"], kitti00["ref"], 1e-10)
        assert rmse(res[r]["pose"], single["pose"]) < 1e-11 and rmse(res[r]["lm"], single["lm"]) < 1e-10
        assert [s["chi2"] for s in res[r]["stats"]] == [s["chi2"] for s in res[0]["stats"]]  # ranks agree bitwise
        ss = res[r]["sstats"]
        if form == "owned":
            assert ss["chol_bcasts"] >= 2 and ss["xchg_sys_bytes"] < 0.75 * ss["xchg_sys_full_bytes"]
        else:
            assert ss["chol_bcasts"] == 0 and ss["xchg_sys_bytes"] == ss["xchg_sys_full_bytes"]


def test_synth10k_full_size(oracle_lib):
    """BASELINE config 4 at full size on one GPU (10 000 poses / 1 000 000 landmarks / 5 000 000 mono
    edges): the first 2 LM iterations against the oracle at 1e-10, then size-independent properties
    of the 10-iteration run — strictly decreasing chi2, the edge count, bitwise reproducibility."""
    d = cugo.synth(10000, 1000000, 5000000, seed=10000, n_loop_closures=0, stereo_fraction=0.0)
    assert not d["e_stereo"].any()
    prob = oracle_lib.Problem(d["pose"], d["pose_fixed"], d["lm"], d["lm_fixed"], d["e_pose"], d["e_lm"],
                              d["e_stereo"], d["e_meas"], d["e_omega"], d["e_cam"])
    g = cugo.graph_from_arrays(d)
    g.initialize(); g.optimize(10)
    st, pose = g.stats(), g.poses()
    assert g.n_active_edges() == 5000000
    chi = [s["chi2"] for s in st]
    assert len(chi) == 10 and all(b < a for a, b in zip(chi, chi[1:]))
    ref = prob.optimize(2)
    assert_trajectories_match(st[:2], ref, 1e-10)
    ids_p, ids_l = np.arange(10000, dtype=np.int32), np.arange(1000000, dtype=np.int32)
    g.set_poses(ids_p, d["pose"]); g.set_landmarks(ids_l, d["lm"])
    g.initialize(); g.optimize(10)
    chi2 = [s["chi2"] for s in g.stats()]
    assert chi == chi2, [(i, a, b) for i, (a, b) in enumerate(zip(chi, chi2)) if a != b]
    assert np.array_equal(pose, g.poses())
    # no wait for an LM trial ever returned before the trial's numbers were in the pinned block
    assert g.structure_stats()["trial_sync_retries"] == 0
    g.close()


def test_native_comm_single_rank(oracle_lib):
    """cugo_comm_* (RCCL inside the library): a 1-rank communicator issues its (identity)
    ncclAllReduce calls on the solver's stream; the run is bitwise the single-GPU run.  RCCL does not
    allow two ranks on one device, so the multi-rank form is exercised by bench.py --gpus N only."""
    d, _ = synth_problem(oracle_lib, 150, 2200, 9000, seed=9, lc=60)
    single = run_graph(d, 6)
    comm = cugo.Comm(cugo.comm_unique_id(), 0, 1)
    g = cugo.graph_from_arrays(d)
    g.set_comm(comm)
    g.initialize(); g.optimize(6)
    st, pose, xs = g.stats(), g.poses(), g.exchange_stats()
    g.close()
    comm.close()
    assert [s["chi2"] for s in st] == [s["chi2"] for s in single["stats"]]
    assert np.array_equal(pose, single["pose"])
    B = single["sstats"]["hsc_blocks"]
    assert xs["calls"] >= 3 + 2 * 6 and xs["bytes"] >= 6 * 8 * (36 * B + 6 * 149)


def test_native_comm_single_rank_runs_the_owned_form(oracle_lib, monkeypatch):
    """The multi-GPU path of the factorisation under a ONE-rank native communicator (the rehearsal a single GPU
    allows; RCCL refuses two ranks on one device): with CUGO_OWN_SUBTREES=1 every elimination subtree below the top
    belongs to rank 0, and every collective of the owned form is really issued on the solver's stream inside
    libcugo_hip.so — ncclReduceScatter of the packed Schur system + ncclAllReduce of the top's part, the update
    blocks into the top by ncclBroadcast inside ncclGroupStart / End after their level, the solution ranges by
    grouped ncclBroadcast, the zero-pivot flag riding in the all-reduce of (F-hat, scale).  With one rank every one
    of them is the identity, so the run must be BITWISE the plain single-GPU run."""
    d, _ = synth_problem(oracle_lib, 400, 8000, 33000, seed=11, lc=200)
    single = run_graph(d, 6)
    monkeypatch.setenv("CUGO_OWN_SUBTREES", "1")
    comm = cugo.Comm(cugo.comm_unique_id(), 0, 1)
    g = cugo.graph_from_arrays(d)
    g.set_comm(comm)
    g.initialize(); g.optimize(6)
    st, pose, lm, xs, ss = g.stats(), g.poses(), g.landmarks(), g.exchange_stats(), g.structure_stats()
    # a second call on the same optimiser re-uses structure, plan and exchange buffers
    g.set_poses(np.arange(400, dtype=np.int32), d["pose"]); g.set_landmarks(np.arange(8000, dtype=np.int32), d["lm"])
    g.initialize(); g.optimize(6)
    st2 = g.stats()
    g.close()
    comm.close()
    assert [s["chi2"] for s in st] == [s["chi2"] for s in single["stats"]]
    assert [s["chi2"] for s in st2] == [s["chi2"] for s in single["stats"]]
    assert np.array_equal(pose, single["pose"]) and np.array_equal(lm, single["lm"])
    # the form really ran: a replicated top above owned subtrees, update blocks and solution ranges broadcast,
    # the system reduce-scattered (one segment = everything below the top) + the top's part all-reduced
    assert ss["chol_top_flops"] > 0 and ss["chol_rank_flops"] > 0 and ss["chol_bcasts"] >= 2
    assert ss["xchg_sys_bytes"] >= ss["xchg_sys_full_bytes"] > 0   # (one rank: everything, plus the segment's padding)
    trials = sum(max(s["trials"], 0) + 1 for s in st)
    assert xs["calls"] >= 3 + 3 * 6 and xs["calls"] <= 3 + 3 * trials + 3


def test_kernel_group_times_add_up_to_the_device_time_of_optimize(oracle_lib):
    """cugo_graph_set_kernel_timing(2) — one HIP event per group boundary (what bench.py's `kernel_groups` are made
    of): a group's time runs to the next group's event, so the groups add up to the device time between the first and
    the last event of optimize() and cannot exceed its wall time (round 3 summed event PAIRS per kernel into group
    totals that exceeded the step).  Mode 1 (a pair round every launch) reports kernels, with the same launch counts
    in every run."""
    import time
    d, _ = synth_problem(oracle_lib, 400, 8000, 33000, seed=11, lc=200)
    g = cugo.graph_from_arrays(d)
    g.initialize(); g.optimize(1)
    g.set_poses(np.arange(400, dtype=np.int32), d["pose"]); g.set_landmarks(np.arange(8000, dtype=np.int32), d["lm"])
    g.initialize()
    g.set_kernel_timing(2)
    t0 = time.perf_counter()
    g.optimize(6)
    wall_ms = (time.perf_counter() - t0) * 1e3
    groups = g.kernel_times()
    chi_timed = [s["chi2"] for s in g.stats()]
    assert set(groups) >= {"build", "schur", "cholesky", "backsubst_update", "errors"} and not any(k.startswith("k_") for k in groups)
    total = sum(v["ms"] for v in groups.values())
    assert 0.3 * wall_ms < total <= wall_ms * 1.001, (total, wall_ms, groups)
    assert groups["cholesky"]["ms"] > groups["errors"]["ms"]
    g.set_kernel_timing(0)
    g.set_poses(np.arange(400, dtype=np.int32), d["pose"]); g.set_landmarks(np.arange(8000, dtype=np.int32), d["lm"])
    g.initialize(); g.optimize(6)
    assert [s["chi2"] for s in g.stats()] == chi_timed      # timing changes nothing
    g.close()


def test_sharded_two_ranks_on_one_gpu_matches_single(oracle_lib):
    """landmark-sharded path with world=2 and world=3 emulated in ONE process: the graphs take turns
    and exchange through host memory. Exercises partial Hsc/bsc/chi/scale sums + replicated LL^T."""
    d, prob = synth_problem(oracle_lib, 150, 2200, 9000, seed=9, lc=60)
    single = run_graph(d, 6)
    for world in (2, 3):
        results = run_sharded_in_threads(d, world, 6)
        for r in range(world):
            assert results[r] is not None
            assert_trajectories_match(results[r]["stats"], single["stats"], 1e-11)
            np.testing.assert_allclose(results[r]["pose"], single["pose"], rtol=0, atol=1e-10)
            np.testing.assert_allclose(results[r]["lm"], single["lm"], rtol=0, atol=1e-9)


def test_cpp_sample_reads_reference_json_format(oracle_lib, tmp_path):
    """the C++ sample (mirrored public API, reference sample protocol) on a file in the
    reference's ba_kitti_*.json layout gives the chi2 sequence of the flat-array path and of
    the oracle (warm-up optimize(1), then initialize()+optimize(10) on the updated estimates)"""
    import os
    import re
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "samples", "build", "sample_ba_from_file")
    if not os.path.exists(exe):
        pytest.fail("samples/build/sample_ba_from_file missing: run __graft_entry__.build()")
    d, prob = synth_problem(oracle_lib, 60, 900, 3700, seed=17, lc=0)
    pose_ids = np.arange(60) * 2 + 5
    lm_ids = np.arange(900) + 1000
    path = str(tmp_path / "graph.json")
    cugo.save_ba_json(path, d, pose_ids, lm_ids)
    d2, pid2, lid2 = cugo.load_ba_json(path)
    assert np.array_equal(pid2, pose_ids) and np.array_equal(lid2, lm_ids)
    out = subprocess.run([exe, path, "10"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    chi = [float(m.group(1)) for m in re.finditer(r"iter:\s*\d+, chi2: ([0-9.eE+-]+)", out.stdout)]
    assert len(chi) == 10
    prob.optimize(1)
    ref = prob.optimize(10)
    for a, b in zip(chi, ref):
        assert abs(a - b["chi2"]) <= 0.06  # the sample prints with %.1f
    g = cugo.graph_from_arrays(d2, pose_ids=pid2, lm_ids=lid2)
    g.initialize(); g.optimize(1); g.initialize(); g.optimize(10)
    assert_trajectories_match(g.stats(), ref, 1e-10)
    g.close()
    # GraphOptimisationOptions::useFloat32 through the C++ API: within the stated fp32 tolerance
    out32 = subprocess.run([exe, path, "10", "float32"], capture_output=True, text=True, timeout=120)
    assert out32.returncode == 0, out32.stderr
    chi32 = [float(m.group(1)) for m in re.finditer(r"iter:\s*\d+, chi2: ([0-9.eE+-]+)", out32.stdout)]
    assert len(chi32) == 10
    for a, b in zip(chi32, ref):
        assert abs(a - b["chi2"]) <= 0.06 + 1e-5 * b["chi2"]


def test_repeated_new_optimisers_no_memory_drift(oracle_lib):
    """a new optimiser per call (the ORB-SLAM2 pattern) over changing topologies: device blocks
    and streams are recycled by the process-wide cache, results are bitwise reproducible and the
    free device memory does not drift (tools/soak.py, short run)"""
    import importlib.util
    from conftest import ROOT
    import os
    spec = importlib.util.spec_from_file_location("soak", os.path.join(ROOT, "tools", "soak.py"))
    soak = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(soak)
    import sys
    argv = sys.argv
    try:
        sys.argv = ["soak.py", "32"]
        soak.main()
    finally:
        sys.argv = argv


@pytest.mark.parametrize("mode", ["1", "2"])
def test_no_result_depends_on_unwritten_device_memory(mode):
    """CUGO_POISON_ALLOC (hip_util.h): every device buffer of the library sits between guard zones, and a fresh
    floating-point buffer starts as NaNs (mode 1) or as the finite value 32.5 (mode 2: what max / min /
    comparisons do not swallow).  A part of this suite — golden trajectories, degenerate fixed sets (no free
    landmark / no free pose), float storage, the medium graphs with every Cholesky form, outlier rejection, the
    soak cycles — runs in a child process in that mode: a result that read memory nobody wrote is wrong there,
    and a store past a buffer's end aborts the child when the buffer is released.  (Found this way: the diagonal
    Schur kernels multiplied a zero T block with bl[0], unwritten when no landmark is free.)"""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, CUGO_POISON_ALLOC=mode)
    sel = ("golden or degenerate or float32_block_storage or medium_synthetic or outlier or repeated_new or "
           "more_than_256 or schur_complement_and_backsubst or bitwise_reproducible")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu.py"), "-m", "gpu", "-q", "-x",
                        "-p", "no:cacheprovider", "-k", sel], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "guard zone" not in r.stderr


def test_factorisation_gives_the_same_bits_whichever_waves_run_late():
    """CUGO_DEBUG_DELAY in the HOOKS build (libcugo_hip_hooks.so, `make HOOKS=1`; the product library carries no delay
    pattern): chosen waves / workgroups of k_up_potrf — the task waves, the panel waves, every other wave of phase A or
    B, the potrf workgroups, the extend-add workgroups of the same launch, the SECOND wave of a two-wave panel alone
    (7: the pattern that found the round-3 race) — sleep ~25 k cycles at the start of their phase, and (10..14) chosen
    waves of the trsm / syrk / fused-tile / backward kernels sleep behind EVERY barrier of those kernels.  Every
    hand-over goes through a barrier, so the optimisation must end on the same bits whatever runs late.
    (tools/delay_check.py in a child process that loads the hooks library; the negative control — pattern 8, the
    kernel as it was before the fix — is `python tools/delay_check.py --medium CUGO_DEBUG_DELAY 0 8`, not asserted
    here: the outcome of a data race is not a test oracle.)"""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    hooks = cugo.HOOKS_LIB_PATH
    if not os.path.exists(hooks):
        pytest.fail("libcugo_hip_hooks.so missing: run __graft_entry__.build()")
    env = dict(os.environ, CUGO_LIB=hooks)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "delay_check.py"), "--medium"], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "CUGO_DEBUG_DELAY check ok" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
    assert r.stdout.count(" same ") == 14 and "DIFFERENT" not in r.stdout, r.stdout
