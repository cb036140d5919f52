"""CPU tests: pin oracle/ba_oracle.c against the independent numpy goldens and self-checks."""
import numpy as np
import pytest

import synth
from conftest import GOLDEN_GRAPHS, PROBLEM_KEYS, golden_path, golden_tolerances


def load_problem(oracle, name):
    g = np.load(golden_path(name + ".npz"))
    P = oracle.Problem(*[g[k] for k in PROBLEM_KEYS], rk_type=int(g["rk_type"]),
                       rk_delta=float(g["rk_delta"]))
    return g, P


def test_edge_kats(oracle_lib):
    k = np.load(golden_path("kat_edges.npz"))
    for i in range(len(k["omega"])):
        dim = 3 if k["stereo"][i] else 2
        r = oracle_lib.edge_eval(k["pose"][i], k["Xw"][i], k["meas"][i], dim, k["omega"][i], k["cam"][i])
        np.testing.assert_allclose(r["e"], k["e"][i][:dim], rtol=1e-11, atol=1e-10)
        np.testing.assert_allclose(r["Xc"], k["Xc"][i], rtol=1e-13, atol=1e-13)
        np.testing.assert_allclose(r["JP"], k["JP"][i][:dim], rtol=1e-11, atol=1e-10)
        np.testing.assert_allclose(r["JL"], k["JL"][i][:dim], rtol=1e-11, atol=1e-10)
        assert abs(r["chi"] - k["omega"][i] * (r["e"] @ r["e"])) < 1e-9
        assert r["w"] == k["omega"][i]


def test_expmap_kats(oracle_lib):
    k = np.load(golden_path("kat_expmap.npz"))
    for i in range(len(k["pose"])):
        out = oracle_lib.pose_update(k["pose"][i], k["dx"][i])
        np.testing.assert_allclose(out, k["out"][i], rtol=0, atol=2e-14)
        assert out[3] >= 0


def test_sym3_inv(oracle_lib):
    rng = np.random.default_rng(0)
    for _ in range(20):
        M = rng.normal(size=(3, 3))
        A = M @ M.T + 0.1 * np.eye(3)
        np.testing.assert_allclose(oracle_lib.sym3_inv(A), np.linalg.inv(A), rtol=1e-10, atol=1e-12)


def test_robust_kernels(oracle_lib):
    L = oracle_lib.lib()
    for x in [0.0, 0.5, 3.0, 10.0, 100.0]:
        assert L.ba_rk_rho(0, 2.0, x) == x and L.ba_rk_drho(0, 2.0, x) == 1.0
        d2 = 4.0
        assert abs(L.ba_rk_rho(1, 2.0, x) - d2 * np.log(1 + x / d2)) < 1e-12
        assert abs(L.ba_rk_drho(1, 2.0, x) - 1 / (1 + x / d2)) < 1e-15
        t = (d2 / 3) * (1 - (1 - x / d2) ** 3) if x <= d2 else d2 / 3
        assert abs(L.ba_rk_rho(2, 2.0, x) - t) < 1e-12
        assert abs(L.ba_rk_drho(2, 2.0, x) - ((1 - x / d2) ** 2 if x <= d2 else 0.0)) < 1e-15


@pytest.mark.parametrize("name", GOLDEN_GRAPHS)
@pytest.mark.parametrize("dense", [True, False])
def test_lm_trajectory_vs_numpy_golden(oracle_lib, name, dense):
    g, P = load_problem(oracle_lib, name)
    assert abs(P.compute_errors() - float(g["chi0"])) <= 1e-12 * max(1.0, float(g["chi0"]))
    r = P.optimize(10, dense=dense)
    tr = g["trace"]
    tols, etol = golden_tolerances(name)
    if name.startswith("zero_noise"):
        # chi2 ~ 1e-25: the sign of rho is round-off, only "stays at the optimum" is defined
        assert len(r) <= 2 and all(a["chi2"] < 1e-12 for a in r)
        np.testing.assert_allclose(P.pose, g["pose_out"], rtol=0, atol=1e-9)
        return
    assert len(r) == len(tr)
    for a, t, tol in zip(r, tr, tols):
        assert a["trials"] == int(t[4])
        assert abs(a["chi2"] - t[1]) <= tol * max(abs(t[1]), 1e-6)
        assert abs(a["lam"] - t[2]) <= max(tol * 10, 1e-9) * abs(t[2])
    np.testing.assert_allclose(P.pose, g["pose_out"], rtol=0, atol=etol)
    np.testing.assert_allclose(P.lm, g["lm_out"], rtol=0, atol=etol * 10)


@pytest.mark.parametrize("name", ["tiny_3x8", "small_10x200", "cauchy_8x80"])
def test_first_step_vs_full_dense_system(oracle_lib, name):
    """dx from the oracle's Schur route equals the golden's solve of the same damped system."""
    g, P = load_problem(oracle_lib, name)
    s = P.build_system()
    b = np.concatenate([s["bp"].ravel(), s["bl"].ravel()])
    np.testing.assert_allclose(b, g["b0"], rtol=1e-12, atol=1e-9)
    for dense in (True, False):
        ok, dxp, dxl = P.solve_step(float(g["lam0"]), dense=dense)
        assert ok
        dx = np.concatenate([dxp.ravel(), dxl.ravel()])
        np.testing.assert_allclose(dx, g["dx0"], rtol=1e-8, atol=1e-11)


def test_sparse_block_cholesky_vs_numpy(oracle_lib):
    rng = np.random.default_rng(4)
    nb = 40
    # random banded + a few far blocks SPD matrix in upper block CSR
    A = np.zeros((6 * nb, 6 * nb))
    pat = set((i, i) for i in range(nb))
    for i in range(nb):
        for j in range(i + 1, min(nb, i + 4)):
            pat.add((i, j))
    for _ in range(10):
        i, j = sorted(rng.integers(0, nb, 2))
        if i != j:
            pat.add((int(i), int(j)))
    for (i, j) in pat:
        B = rng.normal(size=(6, 6))
        if i == j:
            B = B @ B.T + 60 * np.eye(6)
            A[6 * i:6 * i + 6, 6 * i:6 * i + 6] = B
        else:
            A[6 * i:6 * i + 6, 6 * j:6 * j + 6] = B
            A[6 * j:6 * j + 6, 6 * i:6 * i + 6] = B.T
    rowptr, colind, vals = [0], [], []
    for i in range(nb):
        for j in range(i, nb):
            if (i, j) in pat:
                colind.append(j)
                vals.append(A[6 * i:6 * i + 6, 6 * j:6 * j + 6].T.reshape(-1))  # col-major
        rowptr.append(len(colind))
    b = rng.normal(size=6 * nb)
    ok, x = oracle_lib.bsr_chol_solve(rowptr, colind, np.array(vals), b)
    assert ok
    np.testing.assert_allclose(x, np.linalg.solve(A, b), rtol=1e-10, atol=1e-12)
    # indefinite -> reported as failure (zero-pivot rule, src/cholesky.hpp:85)
    vals2 = np.array(vals).copy()
    vals2[0] = -np.eye(6).reshape(-1)
    ok, _ = oracle_lib.bsr_chol_solve(rowptr, colind, vals2, b)
    assert not ok


def test_fixed_vertices_get_no_update(oracle_lib):
    d = synth.make_problem(n_poses=5, n_landmarks=30, seed=2, fixed_poses=(0, 3), fixed_landmarks=(1, 7))
    P = oracle_lib.Problem(*synth.problem_fields(d))
    pose0, lm0 = P.pose.copy(), P.lm.copy()
    r = P.optimize(5)
    assert r[-1]["chi2"] < r[0]["chi2"] or len(r) == 1
    assert np.array_equal(P.pose[[0, 3]], pose0[[0, 3]])
    assert np.array_equal(P.lm[[1, 7]], lm0[[1, 7]])
    assert not np.array_equal(P.pose[1], pose0[1])
