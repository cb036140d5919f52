"""ctypes binding of oracle/libba_oracle.so (CPU restatement of the reference path).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product never touches it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB = None

RK_NONE, RK_CAUCHY, RK_TUKEY = 0, 1, 2

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_bp = C.POINTER(C.c_ubyte)


class _Problem(C.Structure):
    _fields_ = [
        ("n_poses", C.c_int), ("n_landmarks", C.c_int), ("n_edges", C.c_int),
        ("pose", _dp), ("pose_fixed", _bp), ("lm", _dp), ("lm_fixed", _bp),
        ("e_pose", _ip), ("e_lm", _ip), ("e_stereo", _bp), ("e_meas", _dp),
        ("e_omega", _dp), ("e_cam", _dp), ("rk_type", C.c_int), ("rk_delta", C.c_double),
    ]


class IterInfo(C.Structure):
    _fields_ = [("iteration", C.c_int), ("chi2", C.c_double), ("lambda_", C.c_double),
                ("rho", C.c_double), ("trials", C.c_int)]


def build(force=False):
    so = os.path.join(ORACLE_DIR, "libba_oracle.so")
    src = os.path.join(ORACLE_DIR, "ba_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.ba_compute_errors.restype = C.c_double
        _LIB.ba_build_system.restype = C.c_double
        _LIB.ba_rk_rho.restype = C.c_double
        _LIB.ba_rk_drho.restype = C.c_double
        _LIB.ba_rk_rho.argtypes = [C.c_int, C.c_double, C.c_double]
        _LIB.ba_rk_drho.argtypes = [C.c_int, C.c_double, C.c_double]
    return _LIB


_FAST = {}


def fast_lib(openmp=True):
    """Timing-only variants of the same source for bench.py's cpu_baseline leg: -O3 -march=native,
    and with openmp also -fopenmp -DBA_OMP (OpenMP over edges / landmarks; the summation order is then
    not fixed, so it is never the parity checker).  Compiled on the machine that runs it
    (-march=native) into a temporary directory; omp_set_num_threads selects the thread count."""
    if openmp not in _FAST:
        import tempfile
        out = os.path.join(tempfile.mkdtemp(prefix="ba_oracle_fast_"), "libba_oracle_fast.so")
        flags = ["-fopenmp", "-DBA_OMP"] if openmp else []
        subprocess.check_call(["gcc", "-O3", "-march=native"] + flags + ["-fPIC", "-std=c99", "-shared", "-o", out,
                               os.path.join(ORACLE_DIR, "ba_oracle.c"), "-lm"])
        L = C.CDLL(out)
        L.ba_compute_errors.restype = C.c_double
        L.ba_build_system.restype = C.c_double
        _FAST[openmp] = L
    return _FAST[openmp]


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


class Problem:
    """Flat-array BA problem (same field meaning as ba_problem in oracle/ba_oracle.h).

    Arrays are copied and owned; `pose`/`lm` are updated in place by optimize()."""

    def __init__(self, pose, pose_fixed, lm, lm_fixed, e_pose, e_lm, e_stereo, e_meas, e_omega,
                 e_cam, rk_type=RK_NONE, rk_delta=1.0):
        f8 = lambda a: np.ascontiguousarray(np.array(a, dtype=np.float64))
        self.pose = f8(pose).reshape(-1, 7)
        self.lm = f8(lm).reshape(-1, 3)
        self.pose_fixed = np.ascontiguousarray(np.array(pose_fixed, dtype=np.uint8))
        self.lm_fixed = np.ascontiguousarray(np.array(lm_fixed, dtype=np.uint8))
        self.e_pose = np.ascontiguousarray(np.array(e_pose, dtype=np.int32))
        self.e_lm = np.ascontiguousarray(np.array(e_lm, dtype=np.int32))
        self.e_stereo = np.ascontiguousarray(np.array(e_stereo, dtype=np.uint8))
        self.e_meas = f8(e_meas).reshape(-1, 3)
        self.e_omega = f8(e_omega).reshape(-1)
        cam = f8(e_cam)
        if cam.size == 5:
            cam = np.tile(cam.reshape(1, 5), (len(self.e_pose), 1))
        self.e_cam = np.ascontiguousarray(cam.reshape(-1, 5))
        self.rk_type, self.rk_delta = int(rk_type), float(rk_delta)
        assert len(self.e_lm) == len(self.e_pose) == len(self.e_meas) == len(self.e_omega)

    def copy(self):
        return Problem(self.pose, self.pose_fixed, self.lm, self.lm_fixed, self.e_pose, self.e_lm,
                       self.e_stereo, self.e_meas, self.e_omega, self.e_cam, self.rk_type,
                       self.rk_delta)

    @property
    def n_poses(self):
        return len(self.pose)

    @property
    def n_landmarks(self):
        return len(self.lm)

    @property
    def n_edges(self):
        return len(self.e_pose)

    def _c(self):
        s = _Problem()
        s.n_poses, s.n_landmarks, s.n_edges = self.n_poses, self.n_landmarks, self.n_edges
        s.pose, s.pose_fixed = _p(self.pose, _dp), _p(self.pose_fixed, _bp)
        s.lm, s.lm_fixed = _p(self.lm, _dp), _p(self.lm_fixed, _bp)
        s.e_pose, s.e_lm = _p(self.e_pose, _ip), _p(self.e_lm, _ip)
        s.e_stereo, s.e_meas = _p(self.e_stereo, _bp), _p(self.e_meas, _dp)
        s.e_omega, s.e_cam = _p(self.e_omega, _dp), _p(self.e_cam, _dp)
        s.rk_type, s.rk_delta = self.rk_type, self.rk_delta
        return s

    # ---- oracle entry points -------------------------------------------------------
    def indices(self):
        pi = np.zeros(self.n_poses, np.int32)
        li = np.zeros(self.n_landmarks, np.int32)
        a, b = C.c_int(), C.c_int()
        s = self._c()
        lib().ba_assign_indices(C.byref(s), _p(pi, _ip), _p(li, _ip), C.byref(a), C.byref(b))
        return pi, li, a.value, b.value

    def compute_errors(self, want_arrays=False):
        s = self._c()
        if want_arrays:
            err = np.zeros((self.n_edges, 3))
            xc = np.zeros((self.n_edges, 3))
            chi = lib().ba_compute_errors(C.byref(s), _p(err, _dp), _p(xc, _dp))
            return chi, err, xc
        return lib().ba_compute_errors(C.byref(s), None, None)

    def build_system(self):
        _, _, npf, nlf = self.indices()
        Hpp = np.zeros((npf, 36)); bp = np.zeros((npf, 6))
        Hll = np.zeros((nlf, 9)); bl = np.zeros((nlf, 3))
        Hpl = np.zeros((self.n_edges, 18))
        s = self._c()
        chi = lib().ba_build_system(C.byref(s), _p(Hpp, _dp), _p(bp, _dp), _p(Hll, _dp),
                                    _p(bl, _dp), _p(Hpl, _dp))
        return dict(chi=chi, Hpp=Hpp, bp=bp, Hll=Hll, bl=bl, Hpl=Hpl)

    def schur_dense(self, lam):
        _, _, npf, _ = self.indices()
        H = np.zeros((6 * npf, 6 * npf)); b = np.zeros(6 * npf)
        s = self._c()
        lib().ba_schur_dense(C.byref(s), C.c_double(lam), _p(H, _dp), _p(b, _dp))
        return H, b

    def solve_step(self, lam, dense=False):
        _, _, npf, nlf = self.indices()
        dxp = np.zeros((npf, 6)); dxl = np.zeros((nlf, 3))
        s = self._c()
        ok = lib().ba_solve_step(C.byref(s), C.c_double(lam), int(dense), _p(dxp, _dp),
                                 _p(dxl, _dp))
        return bool(ok), dxp, dxl

    def optimize(self, niter, dense=False, use_lib=None):
        info = (IterInfo * max(niter, 1))()
        s = self._c()
        n = (use_lib or lib()).ba_optimize(C.byref(s), int(niter), int(dense), info)
        return [dict(iteration=info[i].iteration, chi2=info[i].chi2, lam=info[i].lambda_,
                     rho=info[i].rho, trials=info[i].trials) for i in range(n)]


def self_sensitivity(prob, niter, seeds=(1, 2), with_estimates=False):
    """Conditioning probe.  The SAME oracle solves the SAME problem (a) with the edges listed in
    another order (every sum over edges then runs in another, equally valid order) and (b) with its
    other factorisation (dense LL^T instead of block-sparse with minimum-degree ordering).  Returns,
    per LM iteration, the largest relative chi2 difference between those runs and the reference run
    (running maximum over the iterations: a difference made in iteration i is carried into the later
    ones), or None when a probe takes another accept / reject path.  A GPU-vs-oracle difference of the
    same size is round-off amplified by the problem, not an error of either implementation."""
    base = prob.copy()
    ref = base.optimize(niter)
    worst = [0.0] * len(ref)
    runs = []
    for sd in seeds:
        perm = np.random.default_rng(sd).permutation(prob.n_edges)
        q = Problem(prob.pose, prob.pose_fixed, prob.lm, prob.lm_fixed, prob.e_pose[perm], prob.e_lm[perm],
                    prob.e_stereo[perm], prob.e_meas[perm], prob.e_omega[perm], prob.e_cam[perm],
                    prob.rk_type, prob.rk_delta)
        runs.append((q.optimize(niter), q))
    if prob.n_poses <= 400:
        q = prob.copy()
        runs.append((q.optimize(niter, dense=True), q))
    est = 0.0
    for r, q in runs:
        if len(r) != len(ref) or any(a["trials"] != b["trials"] for a, b in zip(r, ref)):
            return None
        for i, (a, b) in enumerate(zip(r, ref)):
            worst[i] = max(worst[i], abs(a["chi2"] - b["chi2"]) / max(abs(b["chi2"]), 1e-6))
        est = max(est, float(np.abs(q.pose - base.pose).max()), float(np.abs(q.lm - base.lm).max()))
    for i in range(1, len(worst)):
        worst[i] = max(worst[i], worst[i - 1])
    if with_estimates:
        return worst, est
    return worst


def edge_eval(pose7, Xw, meas, dim, omega, cam, rk_type=RK_NONE, rk_delta=1.0):
    pose7 = np.ascontiguousarray(pose7, np.float64); Xw = np.ascontiguousarray(Xw, np.float64)
    meas = np.ascontiguousarray(meas, np.float64); cam = np.ascontiguousarray(cam, np.float64)
    e = np.zeros(3); Xc = np.zeros(3); chi = C.c_double(); w = C.c_double()
    JP = np.zeros(dim * 6); JL = np.zeros(dim * 3)
    lib().ba_edge_eval(_p(pose7, _dp), _p(Xw, _dp), _p(meas, _dp), int(dim), C.c_double(omega),
                       _p(cam, _dp), int(rk_type), C.c_double(rk_delta), _p(e, _dp), _p(Xc, _dp),
                       C.byref(chi), _p(JP, _dp), _p(JL, _dp), C.byref(w))
    return dict(e=e[:dim].copy(), Xc=Xc, chi=chi.value, w=w.value,
                JP=JP.reshape(6, dim).T.copy(), JL=JL.reshape(3, dim).T.copy())


def pose_update(pose7, dx6):
    p = np.ascontiguousarray(np.array(pose7, np.float64))
    d = np.ascontiguousarray(np.array(dx6, np.float64))
    lib().ba_pose_update(_p(p, _dp), _p(d, _dp))
    return p


def sym3_inv(A):
    A = np.ascontiguousarray(np.array(A, np.float64).T.reshape(-1))
    B = np.zeros(9)
    lib().ba_sym3_inv(_p(A, _dp), _p(B, _dp))
    return B.reshape(3, 3).T.copy()


def bsr_chol_solve(rowptr, colind, vals, b):
    rowptr = np.ascontiguousarray(rowptr, np.int32); colind = np.ascontiguousarray(colind, np.int32)
    vals = np.ascontiguousarray(vals, np.float64); b = np.ascontiguousarray(b, np.float64)
    x = np.zeros_like(b)
    ok = lib().ba_bsr_chol_solve(len(rowptr) - 1, _p(rowptr, _ip), _p(colind, _ip), _p(vals, _dp),
                                 _p(b, _dp), _p(x, _dp))
    return bool(ok), x
