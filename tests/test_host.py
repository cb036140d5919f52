"""CPU tests of the product's host logic (no GPU compute): library/ABI surface, synthetic
generator, ordering + symbolic multifrontal plan (replayed with numpy), shard ranges."""
import ctypes as C
import importlib
import os
import re

import numpy as np
import pytest

from conftest import ROOT

cugo = importlib.import_module("cuda-bundle-adjustment_amd")


@pytest.fixture(scope="module")
def lib():
    cugo.build()
    return cugo.lib()


def test_library_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "cugo_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = sorted(set(re.findall(r"\b(cugo_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) > 40
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_fails_loudly_without_gpu(lib):
    if cugo.device_count() > 0:
        pytest.skip("a GPU is present")
    g = C.c_void_p()
    rc = lib.cugo_graph_create(1, 1, C.byref(g))
    assert rc == -1  # CUGO_ERR_NO_DEVICE
    assert b"no HIP device" in lib.cugo_last_error()
    ctx = C.c_void_p()
    assert lib.cugo_ctx_create(-1, C.byref(ctx)) == -1


def test_synthetic_generator_shapes_and_determinism(lib, oracle_lib):
    P, L, E = 200, 900, 3600
    a = cugo.synth(P, L, E, seed=7, n_loop_closures=40)
    b = cugo.synth(P, L, E, seed=7, n_loop_closures=40)
    c = cugo.synth(P, L, E, seed=8, n_loop_closures=40)
    for k in a:
        assert np.array_equal(a[k], b[k])
    assert not np.array_equal(a["e_meas"], c["e_meas"])
    assert len(a["e_pose"]) == E and a["e_pose"].max() < P and a["e_lm"].max() == L - 1
    pairs = a["e_pose"].astype(np.int64) * L + a["e_lm"]
    assert len(np.unique(pairs)) == E                      # no duplicate (pose, landmark) edges
    assert np.bincount(a["e_lm"], minlength=L).min() >= 2  # every landmark seen twice or more
    # quaternions normalised, w >= 0
    assert np.allclose(np.linalg.norm(a["pose"][:, :4], axis=1), 1.0) and (a["pose"][:, 3] >= 0).all()
    # the CPU oracle reduces chi2 on it (the graph is a sane BA problem)
    prob = oracle_lib.Problem(a["pose"], a["pose_fixed"], a["lm"], a["lm_fixed"], a["e_pose"], a["e_lm"],
                              a["e_stereo"], a["e_meas"], a["e_omega"], a["e_cam"])
    chi0 = prob.compute_errors()
    r = prob.optimize(6)
    assert r[-1]["chi2"] < 0.05 * chi0
    assert r[-1]["chi2"] < 4.0 * E  # ~ noise level (1 px, 2-3 dof per edge, information <= 1)


def covis_pattern(n_poses_free, e_pose_idx, e_lm_idx):
    """upper block CSR (diag first, ascending) of the Schur complement pattern"""
    rows = [set([p]) for p in range(n_poses_free)]
    order = np.argsort(e_lm_idx, kind="stable")
    lm_sorted = e_lm_idx[order]
    starts = np.flatnonzero(np.r_[True, lm_sorted[1:] != lm_sorted[:-1], True])
    for a, b in zip(starts[:-1], starts[1:]):
        ps = sorted(set(int(p) for p in e_pose_idx[order[a:b]] if p < n_poses_free))
        for i, p in enumerate(ps):
            rows[p].update(ps[i:])
    rowptr, colind = [0], []
    for p in range(n_poses_free):
        colind += sorted(rows[p])
        rowptr.append(len(colind))
    return np.array(rowptr, np.int32), np.array(colind, np.int32)


def random_spd_bsr(rowptr, colind, rng):
    n = len(rowptr) - 1
    A = np.zeros((6 * n, 6 * n))
    vals = np.zeros((len(colind), 36))
    deg = np.zeros(n)
    for r in range(n):
        for k in range(rowptr[r], rowptr[r + 1]):
            c = colind[k]
            if c != r:
                B = rng.normal(size=(6, 6))
                A[6 * r:6 * r + 6, 6 * c:6 * c + 6] = B
                A[6 * c:6 * c + 6, 6 * r:6 * r + 6] = B.T
                vals[k] = B.T.reshape(-1)  # column-major
                deg[r] += 1
                deg[c] += 1
    for r in range(n):
        M = rng.normal(size=(6, 6))
        D = M @ M.T + (8.0 * deg[r] + 6.0) * np.eye(6)
        A[6 * r:6 * r + 6, 6 * r:6 * r + 6] = D
        vals[rowptr[r]] = D.T.reshape(-1)
    return A, vals


def plan_arrays(lib, s):
    out = {}
    for name in ["perm", "super_ptr", "rows_ptr", "rows", "sparent", "child_ptr", "child", "rel_ptr",
                 "rel", "ncb", "nb", "col0", "col_front", "stage_task_ptr", "task_ptr", "task_fronts",
                 "blk_front", "blk_row", "blk_col", "blk_trans"]:
        p = C.POINTER(C.c_int32)()
        n = lib.cugo_chol_plan_array(s, name.encode(), C.byref(p))
        assert n >= 0, name
        out[name] = np.ctypeslib.as_array(p, shape=(n,)).copy() if n else np.zeros(0, np.int32)
    return out


def replay_multifrontal(pl, vals, lam, b):
    """numpy replay of the device algorithm using ONLY the plan's maps (assembly, extend-add
    through rel, stage/task order, rhs-row trick, backward pass)."""
    ns = len(pl["ncb"])
    n = len(pl["perm"])
    F = []
    for f in range(ns):
        nb = pl["nb"][f]
        F.append(np.zeros((6 * nb + 1, 6 * nb)))
    for k in range(len(pl["blk_front"])):
        f, rb, cb, tr = pl["blk_front"][k], pl["blk_row"][k], pl["blk_col"][k], pl["blk_trans"][k]
        B = vals[k].reshape(6, 6).T  # stored column-major
        if rb == cb:
            Bl = np.tril(B) + lam * np.eye(6)
            F[f][6 * rb:6 * rb + 6, 6 * cb:6 * cb + 6] = Bl
        else:
            F[f][6 * rb:6 * rb + 6, 6 * cb:6 * cb + 6] = B.T if tr else B
    for j in range(n):
        f = pl["col_front"][j]
        lc = 6 * (j - pl["col0"][f])
        F[f][-1, lc:lc + 6] = b[6 * pl["perm"][j]:6 * pl["perm"][j] + 6]
    done = np.zeros(ns, bool)
    order = []
    for st in range(len(pl["stage_task_ptr"]) - 1):
        for t in range(pl["stage_task_ptr"][st], pl["stage_task_ptr"][st + 1]):
            for f in pl["task_fronts"][pl["task_ptr"][t]:pl["task_ptr"][t + 1]]:
                order.append(f)
                # children must be complete (earlier stage, or earlier in the same task)
                for c in pl["child"][pl["child_ptr"][f]:pl["child_ptr"][f + 1]]:
                    assert done[c]
                    ncb, nb = pl["ncb"][c], pl["nb"][c]
                    rel = pl["rel"][pl["rel_ptr"][c]:pl["rel_ptr"][c + 1]]
                    assert len(rel) == nb - ncb
                    idx = np.concatenate([6 * np.repeat(rel, 6) + np.tile(np.arange(6), len(rel)),
                                          [F[f].shape[0] - 1]]).astype(int)
                    U = F[c][6 * ncb:, 6 * ncb:]
                    Ul = np.tril(U[:-1]) if U.shape[1] else U[:-1]
                    F[f][np.ix_(idx[:-1], idx[:-1])] += Ul
                    F[f][-1, idx[:-1]] += U[-1]
                nc = 6 * pl["ncb"][f]
                A11 = np.tril(F[f][:nc, :nc]); A11 = A11 + np.tril(A11, -1).T
                L11 = np.linalg.cholesky(A11)
                L21 = np.linalg.solve(L11, F[f][nc:, :nc].T).T
                F[f][:nc, :nc] = L11
                F[f][nc:, :nc] = L21
                S = L21 @ L21[:-1].T
                F[f][nc:, nc:] -= S
                done[f] = True
    assert done.all() and len(order) == ns
    xnew = np.zeros(6 * n)
    for f in reversed(order):
        nc = 6 * pl["ncb"][f]
        rows = pl["rows"][pl["rows_ptr"][f]:pl["rows_ptr"][f + 1]]
        ridx = (6 * np.repeat(rows, 6) + np.tile(np.arange(6), len(rows))).astype(int)
        y = F[f][-1, :nc].copy()
        L21 = F[f][nc:-1, :nc]
        v = y - L21.T @ xnew[ridx]
        xj = np.linalg.solve(F[f][:nc, :nc].T, v)
        c0 = 6 * pl["col0"][f]
        xnew[c0:c0 + nc] = xj
    x = np.zeros(6 * n)
    for j in range(n):
        x[6 * pl["perm"][j]:6 * pl["perm"][j] + 6] = xnew[6 * j:6 * j + 6]
    return x


def patterns():
    rng = np.random.default_rng(0)
    out = {}
    # banded chain
    n = 70
    rows = [[c for c in range(r, min(n, r + 5))] for r in range(n)]
    out["band"] = rows
    # chain + loop closures
    rows2 = [list(r) for r in rows]
    for _ in range(12):
        a, b = sorted(rng.integers(0, n, 2))
        if a != b and b not in rows2[a]:
            rows2[a].append(int(b))
    out["band_loops"] = [sorted(r) for r in rows2]
    # random sparse
    n3 = 40
    rows3 = [[r] for r in range(n3)]
    for _ in range(100):
        a, b = sorted(rng.integers(0, n3, 2))
        if a != b and b not in rows3[a]:
            rows3[a].append(int(b))
    out["random"] = [sorted(r) for r in rows3]
    out["dense"] = [list(range(r, 9)) for r in range(9)]
    out["diag_only"] = [[r] for r in range(7)]
    out["single"] = [[0]]
    # two disconnected chains
    rows4 = [[r] + ([r + 1] if (r + 1) % 15 else []) for r in range(30)]
    out["two_components"] = [sorted(set(c for c in r if c < 30)) for r in rows4]
    return out


@pytest.mark.parametrize("name", list(patterns().keys()) + ["synthetic"])
def test_one_pass_assembly_map_equals_clear_plus_scatter(lib, name):
    """k_assemble_fronts (chol_kernels.hip) writes every lower-triangle entry of every stored front once, from
    CholPlan::asm_map; replayed here in numpy it must leave the fronts exactly as clearing them and scattering
    the Hsc blocks + right-hand side through blk_front / blk_row / blk_col / blk_trans does — fronts stored
    inside their child's update block included."""
    rng = np.random.default_rng(5)
    if name == "synthetic":
        d = cugo.synth(120, 1500, 6200, seed=3, n_loop_closures=60)
        ep = d["e_pose"].astype(np.int64) - 1
        ep[ep < 0] = 10**6
        rowptr, colind = covis_pattern(119, ep, d["e_lm"])
    else:
        rows = patterns()[name]
        rowptr = np.array([0] + list(np.cumsum([len(r) for r in rows])), np.int32)
        colind = np.array([c for r in rows for c in r], np.int32)
    n = len(rowptr) - 1
    _, vals = random_spd_bsr(rowptr, colind, rng)
    s = C.c_void_p()
    assert lib.cugo_chol_create(None, C.byref(s)) == 0
    assert lib.cugo_chol_analyze(s, n, rowptr.ctypes.data_as(C.POINTER(C.c_int32)),
                                 colind.ctypes.data_as(C.POINTER(C.c_int32))) == 0, lib.cugo_last_error()
    pl = plan_arrays(lib, s)
    for nm in ["alias_of", "asm_map", "wl", "asm_info"]:
        p = C.POINTER(C.c_int32)()
        k = lib.cugo_chol_plan_array(s, nm.encode(), C.byref(p))
        assert k >= 0, nm
        pl[nm] = np.ctypeslib.as_array(p, shape=(k,)).copy()
    ns = len(pl["ncb"])
    ld, off, total = np.zeros(ns, np.int64), np.zeros(ns, np.int64), 0
    for f in range(ns):  # storage as chol_symbolic.cpp lays it out (5b)
        c = pl["alias_of"][f]
        if c >= 0:
            ld[f], off[f] = ld[c], off[c] + 6 * pl["ncb"][c] * (ld[c] + 1)
        else:
            ld[f], off[f] = 6 * pl["nb"][f] + 1, total
            total += ld[f] * 6 * pl["nb"][f]
    if name == "synthetic":
        assert (pl["alias_of"] >= 0).any()
    lam, b, H = 0.5, rng.normal(size=6 * n), vals.reshape(-1)
    old, lower = np.zeros(total), np.zeros(total, bool)
    for f in range(ns):
        if pl["alias_of"][f] < 0:
            for c in range(6 * pl["nb"][f]):
                lower[off[f] + c * ld[f] + c:off[f] + (c + 1) * ld[f]] = True
    r6, c6 = np.arange(36) % 6, np.arange(36) // 6
    for k in range(len(pl["blk_front"])):
        f, rb, cb, tr = pl["blk_front"][k], pl["blk_row"][k], pl["blk_col"][k], pl["blk_trans"][k]
        v = H[36 * k:36 * k + 36].copy()
        if rb == cb:
            v[r6 == c6] += lam
            keep = r6 >= c6
            old[off[f] + (6 * cb + c6[keep]) * ld[f] + 6 * rb + r6[keep]] = v[keep]
        elif not tr:
            old[off[f] + (6 * cb + c6) * ld[f] + 6 * rb + r6] = v
        else:
            old[off[f] + (6 * cb + r6) * ld[f] + 6 * rb + c6] = v
    for jb in range(n):
        f = pl["col_front"][jb]
        old[off[f] + (6 * (jb - pl["col0"][f]) + np.arange(6)) * ld[f] + 6 * pl["nb"][f]] = b[6 * pl["perm"][jb] + np.arange(6)]
    new = np.full(total, np.nan)
    asm0, nasm, aoff, mp = pl["asm_info"][0], pl["asm_info"][1], pl["asm_info"][2:], pl["asm_map"]
    assert ((aoff >= 0) == (pl["alias_of"] < 0)).all()
    for it in range(nasm):
        f, cb0, cb1 = pl["wl"][3 * (asm0 + it):3 * (asm0 + it) + 3]
        nb, m = pl["nb"][f], mp[aoff[f]:]
        for cb in range(cb0, cb1):
            for rb in range(cb, nb + 1):
                for i in range(6 if rb < nb else 1):
                    base = off[f] + 6 * cb * ld[f] + 6 * rb + i
                    for c in range(6):
                        if rb == nb:
                            jb = m[cb]
                            v = 0.0 if jb < 0 else b[6 * pl["perm"][jb] + c]
                        elif rb > cb or i >= c:
                            src = m[nb + cb * nb - cb * (cb - 1) // 2 + rb - cb]
                            v = 0.0 if src < 0 else (H[36 * (src >> 1) + 6 * i + c] if src & 1 else H[36 * (src >> 1) + i + 6 * c])
                            if rb == cb and i == c and src >= 0:
                                v += lam
                        else:
                            continue
                        assert np.isnan(new[base + c * ld[f]])  # written once
                        new[base + c * ld[f]] = v
    assert (~np.isnan(new) == lower).all()      # every lower-triangle entry of the stored fronts, nothing else
    assert (new[lower] == old[lower]).all() and not old[~lower].any()
    lib.cugo_chol_destroy(s)


@pytest.mark.parametrize("name", list(patterns().keys()) + ["synthetic"])
@pytest.mark.parametrize("env", [{}, {"CUGO_ND_LEAF": "4", "CUGO_MAX_SUPER_COLS": "3", "CUGO_TARGET_TASKS": "4"},
                                 {"CUGO_ND_LEAF": "1000", "CUGO_MAX_SUPER_COLS": "1", "CUGO_TARGET_TASKS": "100000"}])
def test_symbolic_plan_replay_solves_the_system(lib, name, env, monkeypatch):
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    rng = np.random.default_rng(1)
    if name == "synthetic":
        d = cugo.synth(120, 1500, 6200, seed=3, n_loop_closures=60)
        # pose 0 is fixed -> indices: free poses are ids 1..P-1 -> index id-1
        ep = d["e_pose"].astype(np.int64) - 1
        ep[ep < 0] = 10**6
        rowptr, colind = covis_pattern(119, ep, d["e_lm"])
    else:
        rows = patterns()[name]
        rowptr = np.array([0] + list(np.cumsum([len(r) for r in rows])), np.int32)
        colind = np.array([c for r in rows for c in r], np.int32)
    n = len(rowptr) - 1
    A, vals = random_spd_bsr(rowptr, colind, rng)
    s = C.c_void_p()
    assert lib.cugo_chol_create(None, C.byref(s)) == 0
    rc = lib.cugo_chol_analyze(s, n, rowptr.ctypes.data_as(C.POINTER(C.c_int32)),
                               colind.ctypes.data_as(C.POINTER(C.c_int32)))
    assert rc == 0, lib.cugo_last_error()
    pl = plan_arrays(lib, s)
    # permutation is a permutation, supernodes tile the columns, children precede parents
    assert sorted(pl["perm"]) == list(range(n))
    assert pl["super_ptr"][0] == 0 and pl["super_ptr"][-1] == n and (np.diff(pl["super_ptr"]) > 0).all()
    for f, p in enumerate(pl["sparent"]):
        assert p == -1 or p > f
    lam = 0.37
    b = rng.normal(size=6 * n)
    x = replay_multifrontal(pl, vals, lam, b)
    xref = np.linalg.solve(A + lam * np.eye(6 * n), b)
    np.testing.assert_allclose(x, xref, rtol=1e-9, atol=1e-11)
    nnzL, flops, nsup, nst, fb = C.c_double(), C.c_double(), C.c_int(), C.c_int(), C.c_double()
    lib.cugo_chol_stats(s, C.byref(nnzL), C.byref(flops), C.byref(nsup), C.byref(nst), C.byref(fb))
    assert nsup.value == len(pl["ncb"]) and nst.value == len(pl["stage_task_ptr"]) - 1
    assert nnzL.value >= 21 * n and flops.value > 0
    lib.cugo_chol_destroy(s)


def test_shard_ranges_partition_all_landmarks(lib):
    rng = np.random.default_rng(2)
    cnt = rng.integers(0, 9, 1000).astype(np.int32)
    for world in (1, 2, 3, 8):
        prev = 0
        loads = []
        for r in range(world):
            a, b = cugo.shard_range(cnt, r, world)
            assert a == prev and b >= a
            prev = b
            loads.append(int(cnt[a:b].sum()))
        assert prev == len(cnt)
        assert max(loads) - min(loads) <= 2 * cnt.max() + 1


def _expected_structure(d):
    """Hsc block count and block-product counts of a flat-array problem, straight from the definition
    (ref: src/sparse_block_matrix.cpp:63-156 and findHschureMulBlockIndicesKernel .cu:1347-1378)"""
    pf, lf = d["pose_fixed"].astype(bool), d["lm_fixed"].astype(bool)
    ff = ~pf[d["e_pose"]] & ~lf[d["e_lm"]]
    pairs = set()
    products = offdiag = 0
    P = len(pf)
    for p in np.flatnonzero(~pf):
        pairs.add((int(p), int(p)))
    order = np.argsort(d["e_lm"][ff], kind="stable")
    ep, el = d["e_pose"][ff][order], d["e_lm"][ff][order]
    starts = np.flatnonzero(np.r_[True, el[1:] != el[:-1], True]) if len(el) else np.array([0])
    for a, b in zip(starts[:-1], starts[1:]):
        ps = sorted(int(p) for p in ep[a:b])
        k = len(ps)
        products += k * (k + 1) // 2
        offdiag += k * (k - 1) // 2
        for i in range(k):
            for j in range(i, k):
                pairs.add((ps[i], ps[j]))
    return len(pairs), products, offdiag


@pytest.mark.parametrize("shape", ["small_mixed", "fixed_heavy", "medium", "all_landmarks_fixed", "threads"])
def test_plan_only_graph_runs_the_whole_host_side(lib, shape):
    """a plan-only graph (no GPU) drives flattening, index assignment, landmark-major layout, Hsc
    pattern, product lists, ordering and symbolic factorisation; its statistics match the
    definitions.  This is also the entry `tools/run_san.sh` runs under ASan + UBSan."""
    import synth
    if shape == "small_mixed":
        d = synth.make_problem(n_poses=14, n_landmarks=260, mean_obs=3.6, seed=4, fixed_poses=(0, 5),
                               fixed_landmarks=(3, 77, 200), loop_closure=True)
    elif shape == "fixed_heavy":
        d = synth.make_problem(n_poses=30, n_landmarks=400, mean_obs=5.0, seed=9, fixed_poses=tuple(range(0, 30, 2)),
                               fixed_landmarks=tuple(range(0, 400, 3)), loop_closure=True, stereo_frac=1.0)
    elif shape == "medium":
        d = cugo.synth(248, 26127, 95037, seed=7, n_loop_closures=500)
    elif shape == "threads":
        # large enough for every threaded host pass: chunked edge sort, per-row structure build
        # (>= 200k co-visibility entries), nested dissection halves on two threads (> 2000 nodes);
        # `tools/run_san.sh thread` runs this under ThreadSanitizer
        d = cugo.synth(2600, 60000, 262000, seed=12, n_loop_closures=1500)
    else:
        d = synth.make_problem(n_poses=9, n_landmarks=80, seed=2, fixed_landmarks=tuple(range(80)))
    B, products, offdiag = _expected_structure(d)
    g = cugo.graph_from_arrays(d, plan_only=True)
    g.initialize()
    s = g.structure_stats()
    assert int(s["hsc_blocks"]) == B and int(s["products"]) == products and int(s["offdiag_products"]) == offdiag
    active = int(np.count_nonzero(~(d["pose_fixed"].astype(bool)[d["e_pose"]] & d["lm_fixed"].astype(bool)[d["e_lm"]])))
    assert g.n_active_edges() == active
    assert s["supernodes"] >= 1 and s["nnzL"] >= 21 * int(np.count_nonzero(d["pose_fixed"] == 0))
    g.initialize()  # re-initialise: the structure is re-used (same topology) and stays consistent
    assert g.structure_stats() == s
    with pytest.raises(cugo.CugoError, match="plan-only"):
        g.optimize(1)
    g.close()


def test_plan_only_sharded_structure_is_global(lib):
    """every shard sees the GLOBAL Hsc pattern (the all-reduce payload has the same layout on every
    rank) and the shards' product lists partition the unsharded one"""
    d = cugo.synth(60, 900, 3700, seed=1)
    B, products, offdiag = _expected_structure(d)
    tot = 0
    for world in (2, 3):
        tot = 0
        for r in range(world):
            g = cugo.graph_from_arrays(d, plan_only=True)
            g.set_shard(r, world, lambda ptr, n, op: None)
            g.initialize()
            s = g.structure_stats()
            assert int(s["hsc_blocks"]) == B
            tot += int(s["offdiag_products"])
            g.close()
        assert tot == offdiag


def test_duplicate_pose_landmark_edges_are_rejected(lib):
    """two active edges between the same free pose and free landmark have no defined behaviour in the
    reference (one Hpl block per pair, ref .cu:1347-1378) and would break the Hsc lists here: the
    flattening refuses them with a clear error; a duplicate that touches a fixed vertex is fine"""
    import synth
    d = synth.make_problem(n_poses=8, n_landmarks=60, seed=3, fixed_poses=(0,))
    ff = np.flatnonzero((d["pose_fixed"][d["e_pose"]] == 0) & (d["lm_fixed"][d["e_lm"]] == 0))
    k = int(ff[len(ff) // 2])

    def with_copy_of(e):
        out = dict(d)
        for key in ("e_pose", "e_lm", "e_stereo", "e_omega", "e_cam"):
            out[key] = np.concatenate([d[key], d[key][e:e + 1]])
        out["e_meas"] = np.concatenate([d["e_meas"], d["e_meas"][e:e + 1] + 0.25], axis=0)
        return out
    g = cugo.graph_from_arrays(with_copy_of(k), plan_only=True)
    with pytest.raises(cugo.CugoError, match="duplicate"):
        g.initialize()
    g.close()
    fixed_e = int(np.flatnonzero(d["pose_fixed"][d["e_pose"]] != 0)[0])
    g = cugo.graph_from_arrays(with_copy_of(fixed_e), plan_only=True)
    g.initialize()
    assert g.n_active_edges() == len(d["e_pose"]) + 1
    g.close()


def test_changing_the_shard_forces_a_full_initialize(lib):
    """initialize(); set_shard(); initialize(): the second call must not take the estimates-only path
    (slot layout, landmark range and Hsc lists belong to the old rank / world)"""
    d = cugo.synth(60, 900, 3700, seed=1)
    _, _, offdiag = _expected_structure(d)
    g = cugo.graph_from_arrays(d, plan_only=True)
    g.initialize()
    assert int(g.structure_stats()["offdiag_products"]) == offdiag
    g.initialize()
    reuses = g.flatten_reuses()
    assert reuses >= 1
    g.set_shard(0, 2, lambda ptr, n, op: None)
    g.initialize()
    assert g.flatten_reuses() == reuses                      # a full flattening ran
    part0 = int(g.structure_stats()["offdiag_products"])
    g.close()
    f = cugo.graph_from_arrays(d, plan_only=True)
    f.set_shard(0, 2, lambda ptr, n, op: None)
    f.initialize()
    assert part0 == int(f.structure_stats()["offdiag_products"]) and 0 < part0 < offdiag
    f.close()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_plan_only_rank_owned_subtrees_partition_the_factorisation(lib, world, monkeypatch):
    """rank-owned elimination subtrees of a sharded run (chol_symbolic.cpp, CholPlan::owner), host side only:
    every rank sees the same replicated top, the ranks' own shares add up with it to the whole factorisation
    (the unsharded plan's), and no rank keeps (nearly) all of it"""
    monkeypatch.setenv("CUGO_OWN_SUBTREES", "1")  # (forced: a graph this small stays replicated by default)
    d = cugo.synth(900, 16000, 66000, seed=4, n_loop_closures=200)
    g = cugo.graph_from_arrays(d, plan_only=True)
    g.initialize()
    whole = g.structure_stats()
    g.close()
    assert whole["chol_top_flops"] == 0 and whole["chol_rank_flops"] > 0
    own, top, recv = [], None, None
    for r in range(world):
        g = cugo.graph_from_arrays(d, plan_only=True)
        g.set_shard(r, world, lambda ptr, n, op: None)
        g.initialize()
        s = g.structure_stats()
        g.close()
        assert s["supernodes"] == whole["supernodes"] and s["nnzL"] == whole["nnzL"]
        top = s["chol_top_flops"] if top is None else top
        assert s["chol_top_flops"] == top and s["chol_bcasts"] >= 1
        own.append(s["chol_rank_flops"])
        # the ownership-keyed exchange of the Schur system: every rank receives one (padded) segment + the top's
        # part — the same number of bytes on every rank, at least 1 / world of the system and well under all of it
        full = 8.0 * (36 * s["hsc_blocks"] + 6 * 899)
        assert s["xchg_sys_full_bytes"] == full
        recv = s["xchg_sys_bytes"] if recv is None else recv
        assert s["xchg_sys_bytes"] == recv and full / world <= recv < 0.9 * full
    assert abs(sum(own) + top - whole["chol_rank_flops"]) <= 1e-9 * whole["chol_rank_flops"]
    assert all(o > 0 for o in own) and max(own) + top < 0.9 * whole["chol_rank_flops"]


def test_landmark_major_sort_paths_give_the_same_structure(lib):
    """Engine::initialize orders the edges landmark-major by one of three paths — the container order is already
    sorted (one edge set, landmark by landmark), it is a few sorted runs (one per edge set: merged), or anything
    (every thread scans all edges) — and all of them must lay out the same graph: same Hsc pattern, products,
    symbolic factor (plan-only graphs: host side only)."""
    d = cugo.synth(300, 5000, 21000, seed=12, n_loop_closures=60, stereo_fraction=0.5)
    order = np.lexsort((d["e_pose"], d["e_lm"]))           # landmark by landmark
    rng = np.random.default_rng(3)
    shuffled = rng.permutation(len(d["e_lm"]))

    def stats(perm, all_mono=False):
        e = dict(d)
        for k in ("e_pose", "e_lm", "e_stereo", "e_meas", "e_omega"):
            e[k] = d[k][perm]
        if all_mono:
            e["e_stereo"] = np.zeros_like(e["e_stereo"])
        g = cugo.graph_from_arrays(e, plan_only=True)
        g.initialize()
        s = g.structure_stats()
        g.close()
        return s
    runs = stats(order)                   # two edge sets (mono, stereo), each sorted: the merge of runs
    anyorder = stats(shuffled)            # many descents: the general path
    assert runs == anyorder
    one_sorted = stats(order, all_mono=True)      # one set, sorted: the identity path
    one_shuffled = stats(shuffled, all_mono=True)
    assert one_sorted == one_shuffled
    assert one_sorted["hsc_blocks"] == runs["hsc_blocks"] and one_sorted["products"] == runs["products"]
