"""Independent numpy/scipy restatement of the BA path -> golden fixtures (tests/golden/*.npz).

Run from the repo root:   python tests/golden/make_golden.py

This script does NOT run or import any reference code (the reference is CUDA C++ and cannot
be built here; it holds no fixtures of its own).  It restates SURVEY.md Appendix A in a way
that is deliberately different from oracle/ba_oracle.c so the two pin each other:
  * rotation by an explicit 3x3 matrix (oracle: quaternion cross-product form),
  * Jacobians assembled as one big dense J and H = J^T W J (oracle: per-edge block sums),
  * damped system solved BOTH as the full (6P+3L) dense system and via a dense Schur
    complement with numpy.linalg.solve (oracle: block-sparse LL^T),
  * SE3 exponential via scipy.linalg.expm of the 4x4 twist (oracle: Rodrigues closed form).
Formulas: /root/reference/src/cuda/cuda_block_solver.cu:379-424 (projection), 491-578
(Jacobians), 1152-1220 (normal equations), 781-823 (update), 972-1027 (robust kernels);
LM control /root/reference/src/cuda_graph_optimisation.cpp:48-154.
"""
import os
import sys

import numpy as np
from scipy.linalg import expm

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def quat_to_R(q):
    return synth.quat_to_R(q)


def R_to_quat(R):
    return synth._R_to_quat(R)


def rho(kind, delta, x):
    d2 = delta * delta
    if kind == 2:  # Tukey
        return (d2 / 3) * (1 - (1 - x / d2) ** 3) if x <= d2 else d2 / 3
    if kind == 1:  # Cauchy
        return d2 * np.log1p(x / d2)
    if kind == 3:  # Huber (g2o RobustKernelHuber)
        return x if x <= d2 else 2 * delta * np.sqrt(x) - d2
    return x


def drho(kind, delta, x):
    d2 = delta * delta
    if kind == 2:
        return (1 - x / d2) ** 2 if x <= d2 else 0.0
    if kind == 1:
        return 1.0 / (1.0 + x / d2)
    if kind == 3:
        return 1.0 if x <= d2 else delta / np.sqrt(x)
    return 1.0


def edge(pose, Xw, meas, stereo, cam):
    """residual e = proj - meas and the reference's Jacobians J = d(meas - proj)/dx."""
    fx, fy, cx, cy, bf = cam
    R = quat_to_R(pose[:4])
    Xc = R @ Xw + pose[4:]
    X, Y, Z = Xc
    proj = np.array([fx * X / Z + cx, fy * Y / Z + cy, fx * X / Z + cx - bf / Z])
    dim = 3 if stereo else 2
    e = proj[:dim] - meas[:dim]
    # d proj / d Xc
    dp = np.array([[fx / Z, 0, -fx * X / Z**2],
                   [0, fy / Z, -fy * Y / Z**2],
                   [fx / Z, 0, -fx * X / Z**2 + bf / Z**2]])[:dim]
    # left perturbation Xc' = exp([w]x) Xc + v  ->  dXc/d[w,v] = [-[Xc]x, I]
    skew = np.array([[0, -Z, Y], [Z, 0, -X], [-Y, X, 0]])
    JP = -dp @ np.hstack([-skew, np.eye(3)])   # minus: J is d(meas-proj)
    JL = -dp @ R
    return e, Xc, JP, JL


class Graph:
    def __init__(self, d, rk=(0, 1.0)):
        self.pose = np.array(d["pose"], float).copy()
        self.lm = np.array(d["lm"], float).copy()
        self.pf = np.array(d["pose_fixed"]).astype(bool)
        self.lf = np.array(d["lm_fixed"]).astype(bool)
        self.ep, self.el = np.array(d["e_pose"]), np.array(d["e_lm"])
        self.st = np.array(d["e_stereo"]).astype(bool)
        self.meas, self.om, self.cam = np.array(d["e_meas"]), np.array(d["e_omega"]), np.array(d["e_cam"])
        self.rk = rk
        # free-first indices
        self.pidx = np.zeros(len(self.pose), int)
        self.pidx[~self.pf] = np.arange((~self.pf).sum())
        self.pidx[self.pf] = (~self.pf).sum() + np.arange(self.pf.sum())
        self.lidx = np.zeros(len(self.lm), int)
        self.lidx[~self.lf] = np.arange((~self.lf).sum())
        self.lidx[self.lf] = (~self.lf).sum() + np.arange(self.lf.sum())
        self.np_, self.nl = int((~self.pf).sum()), int((~self.lf).sum())
        self.active = ~(self.pf[self.ep] & self.lf[self.el])

    def chi2(self):
        tot = [0.0, 0.0]
        for k in np.nonzero(self.active)[0]:
            e, *_ = edge(self.pose[self.ep[k]], self.lm[self.el[k]], self.meas[k], self.st[k], self.cam[k])
            tot[int(self.st[k])] += rho(self.rk[0], self.rk[1], self.om[k] * (e @ e))
        return tot[0] + tot[1]

    def normal_equations(self):
        n = 6 * self.np_ + 3 * self.nl
        H = np.zeros((n, n)); b = np.zeros(n)
        for k in np.nonzero(self.active)[0]:
            ip, il = self.ep[k], self.el[k]
            e, Xc, JP, JL = edge(self.pose[ip], self.lm[il], self.meas[k], self.st[k], self.cam[k])
            w = self.om[k] * drho(self.rk[0], self.rk[1], self.om[k] * (e @ e))
            J = np.zeros((len(e), n))
            if not self.pf[ip]:
                J[:, 6 * self.pidx[ip]:6 * self.pidx[ip] + 6] = JP
            if not self.lf[il]:
                c = 6 * self.np_ + 3 * self.lidx[il]
                J[:, c:c + 3] = JL
            H += w * J.T @ J
            b += w * J.T @ e
        return H, b

    def solve(self, H, b, lam, via_schur=True):
        n = len(b); npd = 6 * self.np_
        Hd = H + lam * np.eye(n)
        if not via_schur:
            return np.linalg.solve(Hd, b)
        Hpp, Hpl, Hll = Hd[:npd, :npd], Hd[:npd, npd:], Hd[npd:, npd:]
        iHll = np.zeros_like(Hll)
        for l in range(self.nl):
            s = slice(3 * l, 3 * l + 3)
            iHll[s, s] = np.linalg.inv(Hll[s, s])
        T = Hpl @ iHll
        Hsc = Hpp - T @ Hpl.T
        bsc = b[:npd] - T @ b[npd:]
        np.linalg.cholesky(Hsc)  # raises if not SPD (reference: zero pivot -> reject)
        dxp = np.linalg.solve(Hsc, bsc)
        dxl = iHll @ (b[npd:] - Hpl.T @ dxp)
        return np.concatenate([dxp, dxl])

    def apply(self, dx):
        npd = 6 * self.np_
        for i in np.nonzero(~self.pf)[0]:
            d = dx[6 * self.pidx[i]:6 * self.pidx[i] + 6]
            tw = np.zeros((4, 4))
            tw[:3, :3] = np.array([[0, -d[2], d[1]], [d[2], 0, -d[0]], [-d[1], d[0], 0]])
            tw[:3, 3] = d[3:]
            Td = expm(tw)
            T = np.eye(4); T[:3, :3] = quat_to_R(self.pose[i, :4]); T[:3, 3] = self.pose[i, 4:]
            Tn = Td @ T
            q = R_to_quat(Tn[:3, :3])
            self.pose[i, :4] = q
            self.pose[i, 4:] = Tn[:3, 3]
        for i in np.nonzero(~self.lf)[0]:
            self.lm[i] += dx[npd + 3 * self.lidx[i]:npd + 3 * self.lidx[i] + 3]

    def optimize(self, niter):
        maxq, tau = 10, 1e-5
        nu, lam = 2.0, 0.0
        trace = []
        for it in range(niter):
            F = self.chi2()
            H, b = self.normal_equations()
            if it == 0:
                lam = tau * max(0.0, np.diag(H).max())
            q, rho_ = 0, -1.0
            while q < maxq and rho_ < 0:
                bak = (self.pose.copy(), self.lm.copy())
                ok = True
                try:
                    dx = self.solve(H, b, lam)
                    self.apply(dx)
                except np.linalg.LinAlgError:
                    ok = False
                    dx = np.zeros_like(b)
                Fhat = self.chi2()
                scale = float(dx @ (lam * dx + b)) + 1e-3
                rho_ = (F - Fhat) / scale if ok else -1.0
                if rho_ > 0:
                    lam *= min(max(1 - (2 * rho_ - 1) ** 3, 1 / 3), 2 / 3)
                    nu = 2.0
                    F = Fhat
                    break
                lam *= nu
                nu *= 2
                self.pose, self.lm = bak
                if not np.isfinite(lam) or (ok and Fhat - F < 1e-4):
                    break
                q += 1
            trace.append((it, F, lam, rho_, q))
            if q == maxq or rho_ < 1e-6 or not np.isfinite(lam):
                break
        return trace


def kat_edges(seed=3, n=12):
    """per-edge known answers incl. fixed flags; finite-difference verified here."""
    rng = np.random.default_rng(seed)
    out = []
    for k in range(n):
        q = synth.quat_from_rotvec(rng.normal(0, 0.4, 3))
        pose = np.concatenate([q / np.linalg.norm(q), rng.normal(0, 1.0, 3)])
        Xc0 = np.array([rng.uniform(-5, 5), rng.uniform(-3, 3), rng.uniform(4, 40)])
        Xw = quat_to_R(pose[:4]).T @ (Xc0 - pose[4:])
        stereo = k % 2 == 1
        cam = synth.KITTI_CAM * (1 + 0.01 * k)
        e0, _, JP, JL = edge(pose, Xw, np.zeros(3), stereo, cam)
        meas = np.zeros(3); meas[:len(e0)] = e0 + rng.normal(0, 2.0, len(e0))
        e, Xc, JP, JL = edge(pose, Xw, meas, stereo, cam)
        # finite differences of e wrt landmark and left pose perturbation
        h = 1e-6
        JLn = np.zeros_like(JL); JPn = np.zeros_like(JP)
        for j in range(3):
            d = np.zeros(3); d[j] = h
            ep, *_ = edge(pose, Xw + d, meas, stereo, cam)
            em, *_ = edge(pose, Xw - d, meas, stereo, cam)
            JLn[:, j] = -(ep - em) / (2 * h)
        for j in range(6):
            d = np.zeros(6); d[j] = h
            g = Graph(dict(pose=[pose], lm=[Xw], pose_fixed=[0], lm_fixed=[0], e_pose=[0], e_lm=[0],
                           e_stereo=[stereo], e_meas=[meas], e_omega=[1.0], e_cam=[cam]))
            g.apply(np.concatenate([d, np.zeros(3)])); ep, *_ = edge(g.pose[0], Xw, meas, stereo, cam)
            g = Graph(dict(pose=[pose], lm=[Xw], pose_fixed=[0], lm_fixed=[0], e_pose=[0], e_lm=[0],
                           e_stereo=[stereo], e_meas=[meas], e_omega=[1.0], e_cam=[cam]))
            g.apply(np.concatenate([-d, np.zeros(3)])); em, *_ = edge(g.pose[0], Xw, meas, stereo, cam)
            JPn[:, j] = -(ep - em) / (2 * h)
        assert np.allclose(JL, JLn, rtol=1e-6, atol=1e-5), (JL, JLn)
        assert np.allclose(JP, JPn, rtol=1e-6, atol=1e-4), (JP, JPn)
        om = float(rng.uniform(0.2, 1.0))
        JPp = np.zeros((3, 6)); JPp[:len(e)] = JP
        JLp = np.zeros((3, 3)); JLp[:len(e)] = JL
        ee = np.zeros(3); ee[:len(e)] = e
        out.append(dict(pose=pose, Xw=Xw, meas=meas, stereo=int(stereo), cam=cam, omega=om, e=ee,
                        Xc=Xc, JP=JPp, JL=JLp))
    return {k: np.array([o[k] for o in out]) for k in out[0]}


def main():
    np.set_printoptions(precision=12)
    only = set(sys.argv[1:])  # optional: regenerate just the named graph fixtures
    # ---- per-edge KATs ------------------------------------------------------------
    if not only:
        np.savez(os.path.join(OUT, "kat_edges.npz"), **kat_edges())
    # ---- exp-map KATs (incl. theta < 1e-5 branch and w<0 flip) ---------------------
    rng = np.random.default_rng(11)
    poses, dxs, outs = [], [], []
    for k in range(10):
        q = synth.quat_from_rotvec(rng.normal(0, 1.2, 3))
        if k == 7:
            q = synth.quat_from_rotvec(np.array([0.0, 3.1, 0.0]))   # near pi: w ~ 0
        pose = np.concatenate([q, rng.normal(0, 2, 3)])
        dx = rng.normal(0, 0.05, 6)
        if k in (3, 4):
            dx[:3] *= 1e-7           # small-angle branch
        if k == 7:
            dx[:3] = [0.0, 0.2, 0.0]  # pushes w negative -> sign flip
        g = Graph(dict(pose=[pose], lm=[np.zeros(3)], pose_fixed=[0], lm_fixed=[1], e_pose=[0],
                       e_lm=[0], e_stereo=[0], e_meas=[np.zeros(3)], e_omega=[1.0], e_cam=[synth.KITTI_CAM]))
        g.apply(np.concatenate([dx]))
        poses.append(pose); dxs.append(dx); outs.append(g.pose[0].copy())
    if not only:
        np.savez(os.path.join(OUT, "kat_expmap.npz"), pose=np.array(poses), dx=np.array(dxs), out=np.array(outs))

    # ---- small graphs: 10-iteration trajectories -----------------------------------
    cases = {
        "tiny_3x8": dict(n_poses=3, n_landmarks=8, mean_obs=2.5, seed=5, stereo_frac=0.5, fixed_landmarks=(2,)),
        "small_10x200": dict(n_poses=10, n_landmarks=200, mean_obs=3.5, seed=7, loop_closure=True),
        "loop_12x150": dict(n_poses=12, n_landmarks=150, seed=1, loop_closure=True),
        "reject_8x60": dict(n_poses=8, n_landmarks=60, mean_obs=3.0, seed=16, pose_noise=(0.05, 0.5), lm_noise=3.0),
        "zero_noise_6x40": dict(n_poses=6, n_landmarks=40, seed=13, pix_noise=0.0, pose_noise=(0, 0), lm_noise=0.0),
        "cauchy_8x80": dict(n_poses=8, n_landmarks=80, seed=17, mean_obs=3.0),
        "tukey_8x80": dict(n_poses=8, n_landmarks=80, seed=19, mean_obs=3.0),
        "huber_8x80": dict(n_poses=8, n_landmarks=80, seed=23, mean_obs=3.0, pix_noise=2.0),
    }
    for name, kw in cases.items():
        if only and name not in only:
            continue
        d = synth.make_problem(**kw)
        rk = ((1, 3.0) if name.startswith("cauchy") else (2, 8.0) if name.startswith("tukey")
              else (3, 1.5) if name.startswith("huber") else (0, 1.0))
        g = Graph(d, rk)
        H, b = g.normal_equations()
        lam0 = 1e-5 * max(0.0, np.diag(H).max())
        dx_full = g.solve(H, b, lam0, via_schur=False)
        dx_schur = g.solve(H, b, lam0, via_schur=True)
        rel = np.linalg.norm(dx_full - dx_schur) / max(np.linalg.norm(dx_full), 1e-300)
        assert rel < 1e-6, rel   # Schur route == full dense solve
        chi0 = g.chi2()
        trace = g.optimize(10)
        print(name, "E=%d" % len(d["e_pose"]), "chi0=%.6f" % chi0, "schur-vs-full rel=%.2e" % rel)
        for t in trace:
            print("   it %d chi2 %.9f lam %.6g rho %.4f q %d" % t)
        np.savez(os.path.join(OUT, name + ".npz"),
                 **{k: d[k] for k in ["pose", "pose_fixed", "lm", "lm_fixed", "e_pose", "e_lm", "e_stereo",
                                      "e_meas", "e_omega", "e_cam"]},
                 rk_type=rk[0], rk_delta=rk[1], chi0=chi0, H0=H if H.shape[0] <= 64 else np.zeros(0),
                 b0=b, lam0=lam0, dx0=dx_schur,
                 trace=np.array(trace, float), pose_out=g.pose, lm_out=g.lm)


if __name__ == "__main__":
    main()
