import os
import sys

import pytest

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # before anything initialises the HIP runtime (mi355/__init__.py, mi355/dp.py)
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "medical-image-segmentation-and-classification_amd")
for p in (REPO, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


_CONFIG = None


def _usable_cores():
    """the affinity mask capped by the cgroup CPU quota (a 1-GPU box shows every core of its host but owns a 16-core share)"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def pytest_sessionstart(session):
    global _CONFIG
    _CONFIG = session.config
    # the oracle is CPU torch: its default thread count is the host's core count, which oversubscribes a quota'd box many times over
    import torch
    torch.set_num_threads(min(_usable_cores(), 16))


@pytest.hookimpl(trylast=True)
def pytest_runtest_logreport(report):
    """Flush the progress output after every test: with stdout on a pipe or a file pytest's dots sit in a block buffer until the
    run ends, and a GPU box that sees no output for several minutes takes the run for hung."""
    if report.when == "teardown" and _CONFIG is not None:
        tr = _CONFIG.pluginmanager.get_plugin("terminalreporter")
        if tr is not None:
            tr.flush()
        sys.stdout.flush()
        sys.stderr.flush()
