import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def _cpu_quota():
    """CPUs the container may use per scheduling period (cgroup v2 cpu.max), capped by the visible CPUs"""
    import os
    n = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    return n


def pytest_sessionstart(session):
    # The GPU boxes give the container 16 CPUs' worth of time per 100 ms under 256 visible CPUs; torch's default of 128 intra-op threads
    # burns that in a fraction of the period and the whole process is descheduled for the rest (profiles/r05/README.md section 0): the
    # oracle runs FASTER on as many threads as the quota has CPUs.
    import torch
    torch.set_num_threads(max(1, min(_cpu_quota(), torch.get_num_threads())))
