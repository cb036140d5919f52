"""Opt-in pytest plugin (NOT loaded by the test suite): a GPU test that fails is run ONCE more and every such
repeat is listed at the end of the run and appended to gpurun_out/repeated_tests.txt — a recorder for rare events,
written while the race of DESIGN.md section 2 was hunted.  The suite itself is strict: tests/conftest.py has no
retry hook.

    python -m pytest tests -m gpu -p tools.repeat_failed_plugin
"""
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_REPEATED = []


@pytest.hookimpl(tryfirst=True)
def pytest_runtest_protocol(item, nextitem):
    if item.get_closest_marker("gpu") is None:
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
