import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    _spread_the_cpu_suite(config)


def _spread_the_cpu_suite(config):
    """`-m "not gpu"` interprets the unchanged GPU kernels on the host (tests/hipsim: fibers, lockstep SIMT) and runs the multi-process gloo
    tests: 17 minutes in one process.  Such a run is spread over a few pytest-xdist workers (each test is self-contained; the data-parallel
    tests pick their rendezvous ports from their own pid) unless the caller chose a distribution (-n / --dist / -p no:xdist) or sets
    LXO_TESTS_SERIAL=1.  A `-m gpu` run is NEVER spread: the persistent decoder chains want every CU of the one GPU."""
    if hasattr(config, "workerinput") or os.environ.get("LXO_TESTS_SERIAL") == "1":
        return
    opt = config.option
    if "not gpu" not in (getattr(opt, "markexpr", "") or "") or not config.pluginmanager.hasplugin("xdist"):
        return
    if getattr(opt, "numprocesses", None) is not None or getattr(opt, "dist", "no") != "no" or getattr(opt, "tx", None) or getattr(opt, "collectonly", False):
        return
    try:
        ncpu = len(os.sched_getaffinity(0))
    except Exception:
        ncpu = os.cpu_count() or 1
    n = max(1, min(4, ncpu // 2))
    if n < 2:
        return
    try:                                                          # one build of the interpreter library before the workers start (theirs then find it up to date)
        from simlib import build_sim
        build_sim()
    except Exception:
        return
    opt.numprocesses, opt.dist, opt.tx = n, "load", ["popen"] * n
