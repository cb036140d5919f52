import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_path(name):
    return os.path.join(HERE, "golden", name)


PROBLEM_KEYS = ["pose", "pose_fixed", "lm", "lm_fixed", "e_pose", "e_lm", "e_stereo", "e_meas",
                "e_omega", "e_cam"]
GOLDEN_GRAPHS = ["tiny_3x8", "small_10x200", "loop_12x150", "reject_8x60", "zero_noise_6x40",
                 "cauchy_8x80", "tukey_8x80", "huber_8x80"]
# chi2 relative tolerance: 1e-10 (the north star) on every fixture.  The stress fixture reject_8x60
# starts 4 orders of magnitude from the optimum and takes rejected trials; it amplifies round-off by
# itself: the oracle run against ITSELF with the edges summed in another order differs by up to 3e-9
# from iteration 6 on (tests/oracle.self_sensitivity).  For such a fixture the bar of iteration i is
# max(1e-10, 4 x that measured self-sensitivity) — derived from the problem, not picked.
STRESS_FIXTURES = ("reject_8x60",)
_TOL_CACHE = {}


def golden_tolerances(name, niter=10):
    """(per-iteration relative chi2 tolerance list, absolute tolerance of the final estimates)"""
    if name not in STRESS_FIXTURES:
        return [1e-10] * niter, 1e-9
    if name not in _TOL_CACHE:
        import numpy as np
        import oracle
        g = np.load(golden_path(name + ".npz"))
        P = oracle.Problem(*[g[k] for k in PROBLEM_KEYS], rk_type=int(g["rk_type"]), rk_delta=float(g["rk_delta"]))
        sens, est = oracle.self_sensitivity(P, niter, seeds=(1, 2, 3), with_estimates=True)
        _TOL_CACHE[name] = ([max(1e-10, 4.0 * s) for s in sens], max(1e-9, 4.0 * est))
    return _TOL_CACHE[name]


@pytest.fixture(scope="session")
def oracle_lib():
    import oracle
    oracle.lib()
    return oracle


# ---- optional: one repeat for a GPU test that fails (CUGO_TEST_REPEAT=1) -----------------------------------
# Written while a rare run-to-run deviation was being hunted (a race between the two waves of
# a wide panel in k_up_potrf, about one run in a thousand, since fixed): with CUGO_TEST_REPEAT=1 a GPU test that
# fails is run ONCE more, and every such repeat is reported at the end of the run (and appended to
# gpurun_out/repeated_tests.txt) with the first failure's message — a recorder for rare events, off by default:
# the suite is strict.
_REPEATED = []


@pytest.hookimpl(tryfirst=True)
def pytest_runtest_protocol(item, nextitem):
    if item.get_closest_marker("gpu") is None or os.environ.get("CUGO_TEST_REPEAT") != "1":
        return None
    from _pytest.runner import runtestprotocol
    item.ihook.pytest_runtest_logstart(nodeid=item.nodeid, location=item.location)
    reports = runtestprotocol(item, nextitem=nextitem, log=False)
    failed = [r for r in reports if r.when == "call" and r.failed]
    if failed:
        first = str(failed[0].longrepr).strip().splitlines()
        _REPEATED.append((item.nodeid, next((ln.strip() for ln in first if ln.startswith("E ")), first[-1] if first else "")))
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "repeated_tests.txt"), "a") as fh:
                fh.write(item.nodeid + "\n" + "\n".join(first[-25:]) + "\n\n")
        except OSError:
            pass
        if hasattr(item, "_initrequest"):
            item._initrequest()  # fresh function-scoped fixtures for the second run
        reports = runtestprotocol(item, nextitem=nextitem, log=False)
    for r in reports:
        item.ihook.pytest_runtest_logreport(report=r)
    item.ihook.pytest_runtest_logfinish(nodeid=item.nodeid, location=item.location)
    return True


def pytest_terminal_summary(terminalreporter):
    if _REPEATED:
        terminalreporter.section("GPU tests that failed once and were run a second time (DESIGN.md section 2)")
        for nodeid, msg in _REPEATED:
            terminalreporter.write_line("%s\n    first run: %s" % (nodeid, msg))
