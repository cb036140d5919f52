"""Randomised parity sweep (not part of the test-suite): small graphs with random sizes, fixed
vertices, stereo fractions, robust kernels, information / camera modes and Cholesky plan knobs,
GPU (through the C ABI) against the CPU oracle.

    python tools/fuzz.py [cases] [seed]           (needs an MI355X)
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
cugo = importlib.import_module("cuda-bundle-adjustment_amd")
import oracle  # noqa: E402
import synth   # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    worst = 0.0
    for c in range(cases):
        P = int(rng.integers(3, 48))
        L = int(rng.integers(8, 500))
        nfix_p = int(rng.integers(1, max(2, P // 4)))
        fixed_p = tuple(sorted(rng.choice(P, nfix_p, replace=False).tolist()))
        nfix_l = int(rng.integers(0, max(1, L // 10)))
        fixed_l = tuple(sorted(rng.choice(L, nfix_l, replace=False).tolist()))
        d = synth.make_problem(n_poses=P, n_landmarks=L, mean_obs=float(rng.uniform(2.2, 6.0)),
                               stereo_frac=float(rng.choice([0.0, 0.5, 1.0])), seed=int(rng.integers(1 << 30)),
                               fixed_poses=fixed_p, fixed_landmarks=fixed_l,
                               loop_closure=bool(rng.integers(0, 2)), pose_noise=(0.005, 0.03))
        rk = [(0, 1.0), (1, 2.5), (2, 6.0), (3, 1.5)][int(rng.integers(0, 4))]
        for k, v in (("CUGO_ND_LEAF", str(int(rng.choice([2, 4, 8, 96])))),
                     ("CUGO_MAX_FRONT_COLS", str(int(rng.choice([2, 5, 16])))),
                     ("CUGO_ALIAS_CHAINS", str(int(rng.integers(0, 2)))),
                     ("CUGO_FLOAT32", "0")):
            os.environ[k] = v
        if rng.integers(0, 4) == 0:
            os.environ["CUGO_MIN_SUBTREE_TASKS"] = "0"
        else:
            os.environ.pop("CUGO_MIN_SUBTREE_TASKS", None)
        prob = oracle.Problem(*synth.problem_fields(d))
        prob.rk_type, prob.rk_delta = rk
        ref = prob.optimize(6)
        g = cugo.graph_from_arrays(d, rk=rk)
        g.initialize()
        g.optimize(6)
        st = g.stats()
        pose, lm = g.poses(), g.landmarks()
        g.close()
        assert len(st) == len(ref), (c, len(st), len(ref))
        for a, b in zip(st, ref):
            rel = abs(a["chi2"] - b["chi2"]) / max(abs(b["chi2"]), 1e-6)
            worst = max(worst, rel)
            if rel > 3e-11:
                print("case %d: P %d L %d rk %s iteration %d rel %.2e chi2 %.6g trials %d lambda %.3g"
                      % (c, P, L, rk, a["iteration"], rel, b["chi2"], b["trials"], b["lam"]))
            assert rel < 1e-7, (c, P, L, rk, a, b)
            assert a["trials"] == b["trials"], (c, a, b)
        assert np.abs(pose - prob.pose).max() < 1e-6 and np.abs(lm - prob.lm).max() < 1e-5, c
    print("fuzz ok: %d cases, worst relative chi2 difference %.2e" % (cases, worst))


if __name__ == "__main__":
    main()
